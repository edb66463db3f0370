#!/bin/bash
# NCC chain probe of round 4: parity tests, wall time of the batch, kernel timeline with all MIP passes first (the chain alone),
# SQ counters of every NCC kernel.   bash profiles/r4_ncc_probe.sh <tag>
tag=${1:-r4}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_ncc.py -x -q -m gpu > gpurun_out/${tag}_ncc_tests.log 2>&1 || { tail -30 gpurun_out/${tag}_ncc_tests.log; exit 1; }
tail -3 gpurun_out/${tag}_ncc_tests.log
timeout -k 5 300 python3 profiles/ncc_batch_probe.py 10 > gpurun_out/${tag}_ncc_wall.txt 2>&1 && cat gpurun_out/${tag}_ncc_wall.txt
MI_NCC_SERIAL_MIPS=1 timeout -k 5 300 python3 profiles/ncc_batch_probe.py 10 > gpurun_out/${tag}_ncc_wall_serial.txt 2>&1 && cat gpurun_out/${tag}_ncc_wall_serial.txt
export MI_NCC_SERIAL_MIPS=1
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ncc_trace -o ncc -- python3 profiles/ncc_batch_probe.py 1 > gpurun_out/ncc_trace.log 2>&1 &&
python3 profiles/ncc_timeline.py gpurun_out/ncc_trace > gpurun_out/${tag}_ncc_timeline_serial.txt && cat gpurun_out/${tag}_ncc_timeline_serial.txt
rm -rf gpurun_out/ncc_trace
unset MI_NCC_SERIAL_MIPS
bash profiles/sq_ncc.sh gpurun_out/${tag}_ncc_sq_counters.txt && grep -E "k_lag|CONFLICT|IDX_ACTIVE" gpurun_out/${tag}_ncc_sq_counters.txt | head -40
