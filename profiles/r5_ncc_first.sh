#!/bin/bash
# round 5, first GPU call: parity of the new MIP pass, A/B against the old one, wall time of the batch.   bash profiles/r5_ncc_first.sh <tag>
tag=${1:-r5a}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_ncc.py tests/test_gpu_terastitcher_golden.py -x -q -m gpu > gpurun_out/${tag}_ncc_tests.log 2>&1 || { tail -40 gpurun_out/${tag}_ncc_tests.log; exit 1; }
tail -3 gpurun_out/${tag}_ncc_tests.log
timeout -k 10 600 python profiles/r5_mips_ab.py > gpurun_out/${tag}_mips_ab.txt 2>&1; cat gpurun_out/${tag}_mips_ab.txt
timeout -k 5 300 python3 profiles/ncc_batch_probe.py 10 > gpurun_out/${tag}_ncc_wall.txt 2>&1 && cat gpurun_out/${tag}_ncc_wall.txt
MI_IPP_PROBES=1 MI_NCC_MIPS_OLD=1 timeout -k 5 300 python3 profiles/ncc_batch_probe.py 10 > gpurun_out/${tag}_ncc_wall_old.txt 2>&1 && cat gpurun_out/${tag}_ncc_wall_old.txt
