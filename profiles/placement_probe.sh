#!/bin/bash
# does keeping the fastest of several allocations of the spectrum arrays remove the two pass speeds?  (bench --no-ncc --no-cpu-baseline)
B="bench.py --steps 20 --warmup 5 --no-ncc --no-cpu-baseline --no-stages"
for t in ${TRIES:-1 3 1 3 1 3 1 3}; do
  MI_FFT_PLACEMENT_LOG=1 MI_FFT_PLACEMENT_TRIES=$t python3 $B 2> gpurun_out/place.err | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; p = r['pass_ms']
        print('tries $t: %.2f ms/iteration  y %.2f/%.2f z %.2f x %.2f/%.2f  %s' % (d['ms_per_step'], p['y_forward'], p['y_inverse'], p['z_conv'], p['x_fused_ratio'], p['x_fused_update'], r.get('pass_mode', {}).get('mode')))
"
  grep "placement trial\|spectrum arrays" gpurun_out/place.err | tr '\n' ' '; echo
done
