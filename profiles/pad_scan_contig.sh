#!/bin/bash
# physically contiguous spectrum arrays: row paddings of the x side (MI_FFT_XPAD, float4 per row) and of the paired z side (MI_FFT_ZPAD)
for xp in 0 8 72 136 264 392 520 1032; do
 for zp in 8 264 520; do
  echo -n "xpad $xp zpad $zp: "
  MI_CONTIG_MIN_MB=1024 MI_FFT_XPAD=$xp MI_FFT_ZPAD=$zp python3 bench.py --steps 6 --warmup 2 --no-ncc --no-cpu-baseline --no-stages 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); p = d['roofline']['pass_ms']
        print('%.2f ms/iteration  y %.2f/%.2f z %.2f x %.2f/%.2f' % (d['ms_per_step'], p['y_forward'], p['y_inverse'], p['z_conv'], p['x_fused_ratio'], p['x_fused_update']))
"
 done
done
