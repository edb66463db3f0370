#!/bin/bash
# kernel trace of two batch calls on the C5 grid (per-kernel start / end / queue) -> gpurun_out/ncc_trace/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ncc_trace -o ncc -- python3 profiles/ncc_batch_probe.py 1 > gpurun_out/ncc_trace.log 2>&1
python3 profiles/ncc_timeline.py gpurun_out/ncc_trace > gpurun_out/ncc_timeline.txt
