#!/bin/bash
# with physically contiguous spectrum arrays the distance between S and T is the same in physical memory as in virtual: scan it
for g in 4224 128 8320 16512 65664 1048704 2101376 3145856 5242880 16781440 33554560 50331776 67113088 134217856 1579904; do
  echo -n "gap $g: "
  MI_CONTIG_MIN_MB=${CONTIG:-1024} MI_FFT_STGAP=$g python3 bench.py --steps 10 --warmup 3 --no-ncc --no-cpu-baseline --no-stages 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); p = d['roofline']['pass_ms']
        print('%.2f ms/iteration  y %.2f/%.2f z %.2f x %.2f/%.2f' % (d['ms_per_step'], p['y_forward'], p['y_inverse'], p['z_conv'], p['x_fused_ratio'], p['x_fused_update']))
"
done
