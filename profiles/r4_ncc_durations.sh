#!/bin/bash
# per-launch durations of the NCC kernels (grid by grid), MIP passes first
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export MI_IPP_PROBES=1 MI_NCC_SERIAL_MIPS=1   # (the switch exists in the probes build only)
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ncc_trace -o ncc -- python3 profiles/ncc_batch_probe.py 2 > gpurun_out/ncc_trace.log 2>&1 &&
python3 profiles/kernel_durations.py gpurun_out/ncc_trace "k_(lag|band|mip|tile|plane)" > gpurun_out/${1:-r4}_ncc_durations.txt && cat gpurun_out/${1:-r4}_ncc_durations.txt
rm -rf gpurun_out/ncc_trace
