#!/bin/bash
# Round profile of the default bench (C3, one GPU): kernel statistics, HBM traffic (PMC, separate passes) and SQ counters.
# Run on the GPU box from the repo root:  bash profiles/collect.sh r01   -> files under gpurun_out/, to be copied into profiles/
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
B="bench.py --steps 4 --warmup 1 --no-ncc --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag -- python3 $B > gpurun_out/prof_$tag.log 2>&1 &&
python3 profiles/summarize.py "$(ls gpurun_out/prof_$tag/*/*_results.db gpurun_out/prof_$tag/*_results.db 2>/dev/null | head -1)" gpurun_out/${tag}_bench_c3_kernel_stats.csv &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_F -o pmc -- python3 $B > gpurun_out/pmc_F.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_W -o pmc -- python3 $B > gpurun_out/pmc_W.log 2>&1 &&
python3 profiles/pmc_summary.py gpurun_out/pmc_F gpurun_out/pmc_W gpurun_out/${tag}_pmc_traffic.json > gpurun_out/${tag}_pmc_traffic.txt &&
bash profiles/sq_pass.sh gpurun_out/${tag}_sq_counters.txt
rm -rf gpurun_out/pmc_F gpurun_out/pmc_W gpurun_out/prof_$tag
