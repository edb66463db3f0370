#!/bin/bash
# The NCC part of profiles/collect.sh alone: kernel statistics and HBM traffic (PMC, separate passes) of the batch on the C5 grid.
#   bash profiles/collect_ncc.sh r03   -> gpurun_out/r03_ncc_c5_kernel_stats.csv, r03_ncc_pmc_traffic.{json,txt}
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
T="timeout -k 5 300"
N="profiles/ncc_batch_probe.py 3"
db() { ls gpurun_out/$1/*/*_results.db gpurun_out/$1/*_results.db 2>/dev/null | head -1; }
$T rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ncc_$tag -o $tag -- python3 $N > gpurun_out/prof_ncc_$tag.log 2>&1
python3 profiles/summarize.py "$(db prof_ncc_$tag)" gpurun_out/${tag}_ncc_c5_kernel_stats.csv
$T rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_NF -o pmc -- python3 $N > gpurun_out/pmc_NF.log 2>&1
$T rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_NW -o pmc -- python3 $N > gpurun_out/pmc_NW.log 2>&1
python3 profiles/pmc_summary.py gpurun_out/pmc_NF gpurun_out/pmc_NW gpurun_out/${tag}_ncc_pmc_traffic.json > gpurun_out/${tag}_ncc_pmc_traffic.txt
rm -rf gpurun_out/pmc_NF gpurun_out/pmc_NW gpurun_out/prof_ncc_$tag
echo "ncc done"
