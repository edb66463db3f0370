#!/bin/bash
# Round-5 profile (one GPU, one box for all of it).  bash profiles/collect_r5.sh r05  -> files under gpurun_out/, copied into profiles/.
# The bench line and the kernel statistics it is read against come from ONE process (the profiled one): its placement, its box.
tag=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
T="timeout -k 5 400"
B="bench.py --steps 20 --warmup 2 --no-ncc --no-cpu-baseline --no-stages"
db() { ls gpurun_out/$1/*/*_results.db gpurun_out/$1/*_results.db 2>/dev/null | head -1; }
export MI_FFT_PLACE_LOG=1
# (1) the headline loop unprofiled, then under the kernel trace: the same command, two processes
$T python3 $B > gpurun_out/${tag}_bench_c3_line_plain.json 2> gpurun_out/${tag}_bench_c3_place_plain.log
$T rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag -- python3 $B > gpurun_out/${tag}_bench_c3_line.json 2> gpurun_out/${tag}_bench_c3_place.log
python3 profiles/summarize.py "$(db prof_$tag)" gpurun_out/${tag}_bench_c3_kernel_stats.csv
echo "bench kernel stats done"
unset MI_FFT_PLACE_LOG
B4="bench.py --steps 4 --warmup 1 --no-ncc --no-cpu-baseline --no-stages"
$T rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_F -o pmc -- python3 $B4 > gpurun_out/pmc_F.log 2>&1
$T rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_W -o pmc -- python3 $B4 > gpurun_out/pmc_W.log 2>&1
python3 profiles/pmc_summary.py gpurun_out/pmc_F gpurun_out/pmc_W gpurun_out/${tag}_pmc_traffic.json > gpurun_out/${tag}_pmc_traffic.txt
echo "bench pmc done"
bash profiles/sq_pass.sh gpurun_out/${tag}_sq_counters.txt
echo "bench sq done"
# (2) NCC: kernel statistics, HBM traffic, SQ counters (MIP passes first: probes build), timelines
N="profiles/ncc_batch_probe.py 3"
$T rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ncc_$tag -o $tag -- python3 $N > gpurun_out/prof_ncc_$tag.log 2>&1
python3 profiles/summarize.py "$(db prof_ncc_$tag)" gpurun_out/${tag}_ncc_c5_kernel_stats.csv
$T rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_NF -o pmc -- python3 $N > gpurun_out/pmc_NF.log 2>&1
$T rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_NW -o pmc -- python3 $N > gpurun_out/pmc_NW.log 2>&1
python3 profiles/pmc_summary.py gpurun_out/pmc_NF gpurun_out/pmc_NW gpurun_out/${tag}_ncc_pmc_traffic.json > gpurun_out/${tag}_ncc_pmc_traffic.txt
bash profiles/sq_ncc.sh gpurun_out/${tag}_ncc_sq_counters.txt
bash profiles/r5_ncc_timeline.sh $tag > /dev/null
(export MI_IPP_PROBES=1 MI_NCC_SERIAL_MIPS=1; $T rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ncc_trace -o ncc -- python3 profiles/ncc_batch_probe.py 1 > gpurun_out/ncc_trace.log 2>&1 && python3 profiles/ncc_timeline.py gpurun_out/ncc_trace > gpurun_out/${tag}_ncc_timeline_serial.txt; rm -rf gpurun_out/ncc_trace)
PROBE_U16=1 $T rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ncc16_$tag -o $tag -- python3 $N > gpurun_out/prof_ncc16_$tag.log 2>&1
python3 profiles/summarize.py "$(db prof_ncc16_$tag)" gpurun_out/${tag}_ncc_u16_kernel_stats.csv
echo "ncc done"
rm -rf gpurun_out/pmc_F gpurun_out/pmc_W gpurun_out/pmc_NF gpurun_out/pmc_NW gpurun_out/prof_$tag gpurun_out/prof_ncc_$tag gpurun_out/prof_ncc16_$tag
# (3) the C4-shaped rank and config 4 whole
stats() {  # stats <name> <command...>
  name=$1; shift
  $T rocprofv3 --kernel-trace --stats -d gpurun_out/prof_x -o $tag -- "$@" > gpurun_out/prof_x_$name.log 2>&1
  python3 profiles/summarize.py "$(db prof_x)" gpurun_out/${tag}_${name}_kernel_stats.csv
  rm -rf gpurun_out/prof_x
}
stats c4_rank python3 profiles/shape_time.py 1024 576 4096 127 63 63
stats c4_whole python3 bench.py --workload c4 --steps 3 --warmup 1 --no-ncc --no-cpu-baseline --no-stages
stats gauss python3 profiles/gauss_time.py
stats direct_c2 python3 bench.py --workload c2 --engine direct --steps 3 --warmup 1 --no-ncc --no-cpu-baseline --no-stages
stats direct_c1 python3 bench.py --workload c1 --engine direct --steps 20 --warmup 2 --no-ncc --no-cpu-baseline --no-stages
echo "all done"
