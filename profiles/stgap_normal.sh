#!/bin/bash
# distance between the two spectrum arrays under the driver's ordinary allocation (MI_FFT_STGAP), several processes per value
cd "$GRAFT_REPO_ROOT" || exit 1
for round in 1 2; do
for g in default 128 4224 65664 1048704 5242880 2097152; do
  echo -n "gap $g: "
  if [ "$g" = default ]; then unset MI_FFT_STGAP; else export MI_FFT_STGAP=$g; fi
  python3 bench.py --steps 10 --warmup 3 --no-ncc --no-cpu-baseline --no-stages 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); p = d['roofline']['pass_ms']
        print('%.2f ms/iteration  y %.2f/%.2f z %.2f x %.2f/%.2f' % (d['ms_per_step'], p['y_forward'], p['y_inverse'], p['z_conv'], p['x_fused_ratio'], p['x_fused_update']))
"
done
done
