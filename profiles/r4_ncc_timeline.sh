#!/bin/bash
# kernel timeline of one batch call (default stream layout and MI_NCC_SERIAL_MIPS=1 in the probes build)
tag=${1:-r4}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for mode in default serial; do
  if [ $mode = serial ]; then export MI_IPP_PROBES=1 MI_NCC_SERIAL_MIPS=1; fi
  timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ncc_trace -o ncc -- python3 profiles/ncc_batch_probe.py 1 > gpurun_out/ncc_trace.log 2>&1 &&
  python3 profiles/ncc_timeline.py gpurun_out/ncc_trace > gpurun_out/${tag}_ncc_timeline_$mode.txt && echo "---- $mode" && cat gpurun_out/${tag}_ncc_timeline_$mode.txt
  rm -rf gpurun_out/ncc_trace
done
