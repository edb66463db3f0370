#!/bin/bash
# kernel statistics of a decwrap run (4.3-GB volume, 27 blocks) -> gpurun_out/<tag>_decwrap_kernel_stats.csv
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 5 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_decwrap -o $tag -- python3 profiles/decwrap_scale_probe.py ${2:-512} 2048 2048 > gpurun_out/prof_decwrap.log 2>&1
db=$(ls gpurun_out/prof_decwrap/*/*_results.db gpurun_out/prof_decwrap/*_results.db 2>/dev/null | head -1)
python3 profiles/summarize.py "$db" gpurun_out/${tag}_decwrap_kernel_stats.csv
tail -n 3 gpurun_out/prof_decwrap.log
rm -rf gpurun_out/prof_decwrap
