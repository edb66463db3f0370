#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
PROBE_WORKERS=${1:-3} timeout -k 5 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/busy -o dw -- python3 profiles/decwrap_scale_probe.py 2048 2048 2048 > gpurun_out/busy.log 2>&1
python3 profiles/busy_timeline.py gpurun_out/busy 500 > gpurun_out/${2:-r05}_decwrap_busy.txt
tail -n 3 gpurun_out/busy.log
rm -rf gpurun_out/busy
