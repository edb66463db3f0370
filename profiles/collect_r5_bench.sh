#!/bin/bash
# part (1) of collect_r5.sh with per-launch traces: the bench line and the kernel statistics it is read against from ONE process
tag=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
T="timeout -k 5 400"
B="bench.py --steps 20 --warmup 2 --no-ncc --no-cpu-baseline --no-stages"
export MI_FFT_PLACE_LOG=1
$T python3 $B > gpurun_out/${tag}_bench_c3_line_plain.json 2> gpurun_out/${tag}_bench_c3_place_plain.log
$T rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_$tag -o $tag -- python3 $B > gpurun_out/${tag}_bench_c3_line.json 2> gpurun_out/${tag}_bench_c3_place.log
python3 profiles/loop_stats.py gpurun_out/prof_$tag gpurun_out/${tag}_bench_c3_kernel_stats.csv gpurun_out/${tag}_bench_c3_loop_launches.txt
rm -rf gpurun_out/prof_$tag
