#!/bin/bash
# Round profile (one GPU): kernel statistics, HBM traffic (PMC, separate passes) and SQ counters of the default bench (C3), and
# kernel statistics + HBM traffic of the NCC batch (C5 grid).  Every profiler run is bounded (a run that has written its
# results but does not exit is killed).
# Run on the GPU box from the repo root:  bash profiles/collect.sh r02 [rl]  -> files under gpurun_out/, to be copied into profiles/
# (second argument "rl": only the RL bench part)
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
T="timeout -k 5 300"
B="bench.py --steps 4 --warmup 1 --no-ncc --no-cpu-baseline --no-stages"
N="profiles/ncc_batch_probe.py 3"
db() { ls gpurun_out/$1/*/*_results.db gpurun_out/$1/*_results.db 2>/dev/null | head -1; }
$T rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag -- python3 $B > gpurun_out/prof_$tag.log 2>&1
python3 profiles/summarize.py "$(db prof_$tag)" gpurun_out/${tag}_bench_c3_kernel_stats.csv
echo "bench kernel stats done"
$T rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_F -o pmc -- python3 $B > gpurun_out/pmc_F.log 2>&1
$T rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_W -o pmc -- python3 $B > gpurun_out/pmc_W.log 2>&1
python3 profiles/pmc_summary.py gpurun_out/pmc_F gpurun_out/pmc_W gpurun_out/${tag}_pmc_traffic.json > gpurun_out/${tag}_pmc_traffic.txt
echo "bench pmc done"
bash profiles/sq_pass.sh gpurun_out/${tag}_sq_counters.txt
echo "bench sq done"
if [ "$2" = "rl" ]; then rm -rf gpurun_out/pmc_F gpurun_out/pmc_W gpurun_out/prof_$tag; exit 0; fi
$T rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ncc_$tag -o $tag -- python3 $N > gpurun_out/prof_ncc_$tag.log 2>&1
python3 profiles/summarize.py "$(db prof_ncc_$tag)" gpurun_out/${tag}_ncc_c5_kernel_stats.csv
$T rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_NF -o pmc -- python3 $N > gpurun_out/pmc_NF.log 2>&1
$T rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_NW -o pmc -- python3 $N > gpurun_out/pmc_NW.log 2>&1
python3 profiles/pmc_summary.py gpurun_out/pmc_NF gpurun_out/pmc_NW gpurun_out/${tag}_ncc_pmc_traffic.json > gpurun_out/${tag}_ncc_pmc_traffic.txt
echo "ncc done"
rm -rf gpurun_out/pmc_F gpurun_out/pmc_W gpurun_out/pmc_NF gpurun_out/pmc_NW gpurun_out/prof_$tag gpurun_out/prof_ncc_$tag
# the smaller passes: Gaussian filters (reg step, decwrap's pre-filter), the slab edge taper of a C3 block, the direct engine on
# BASELINE config 2 (dense taps) and config 1 (rank-1 PSF: single-pass separable kernel)
stats() {  # stats <name> <command...>: kernel statistics of a command -> gpurun_out/${tag}_<name>_kernel_stats.csv
  name=$1; shift
  $T rocprofv3 --kernel-trace --stats -d gpurun_out/prof_x -o $tag -- "$@" > gpurun_out/prof_x_$name.log 2>&1
  python3 profiles/summarize.py "$(db prof_x)" gpurun_out/${tag}_${name}_kernel_stats.csv
  rm -rf gpurun_out/prof_x
}
stats gauss python3 profiles/gauss_time.py
stats edgetaper python3 profiles/edgetaper_slabs_probe.py
stats direct_c2 python3 bench.py --workload c2 --engine direct --steps 3 --warmup 1 --no-ncc --no-cpu-baseline --no-stages
stats direct_c1 python3 bench.py --workload c1 --engine direct --steps 20 --warmup 2 --no-ncc --no-cpu-baseline --no-stages
echo "small passes done"
