#!/bin/bash
# kernel trace of the contention probe -> gpurun_out/r03_overlap_timeline.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ovl_trace -o ovl -- python3 profiles/overlap_probe.py trace > gpurun_out/ovl_trace.log 2>&1
python3 profiles/overlap_timeline.py gpurun_out/ovl_trace > gpurun_out/r03_overlap_timeline.txt
