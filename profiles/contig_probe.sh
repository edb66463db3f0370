#!/bin/bash
# pass speeds of the bench after the device's memory has been churned by the test suite: plain hipMalloc vs contiguous blocks
python -m pytest tests/test_gpu_fft_native.py tests/test_gpu_slab.py -x -q -m gpu > /dev/null 2>&1
for m in 0 1024 0 1024 0 1024; do
  echo -n "MI_CONTIG_MIN_MB=$m: "
  MI_CONTIG_MIN_MB=$m python3 bench.py --steps 20 --warmup 5 --no-ncc --no-cpu-baseline --no-stages 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); p = d['roofline']['pass_ms']
        print('%.2f ms/iteration  y %.2f/%.2f z %.2f x %.2f/%.2f' % (d['ms_per_step'], p['y_forward'], p['y_inverse'], p['z_conv'], p['x_fused_ratio'], p['x_fused_update']))
"
done
