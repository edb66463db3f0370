#!/bin/bash
# round 5: parity of the MIP passes (float, 16-bit, 8-bit), wall time of the batch, MIP launch times.   bash profiles/r5_ncc_second.sh <tag>
tag=${1:-r5b}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_ncc.py tests/test_gpu_terastitcher_golden.py "tests/test_gpu_full_size.py::test_ncc_full_size_pair_vs_oracle" -x -q -m gpu > gpurun_out/${tag}_ncc_tests.log 2>&1 || { tail -40 gpurun_out/${tag}_ncc_tests.log; exit 1; }
tail -3 gpurun_out/${tag}_ncc_tests.log
timeout -k 5 300 python3 profiles/ncc_batch_probe.py 10 > gpurun_out/${tag}_ncc_wall.txt 2>&1 && cat gpurun_out/${tag}_ncc_wall.txt
PROBE_U16=1 timeout -k 5 300 python3 profiles/ncc_batch_probe.py 10 > gpurun_out/${tag}_ncc_wall_u16.txt 2>&1 && cat gpurun_out/${tag}_ncc_wall_u16.txt
timeout -k 5 600 python3 bench_ncc.py > gpurun_out/${tag}_bench_ncc.json 2> gpurun_out/${tag}_bench_ncc.err; tail -c 3000 gpurun_out/${tag}_bench_ncc.json
