#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the passes of an arbitrary shape: profiles/shape_time.py <z y x>
#   bash profiles/pmc_shape.sh <tag> <z> <y> <x>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export MI_FFT_PLACE_CANDIDATES=1
T="timeout -k 5 300"
$T rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_SF -o pmc -- python3 profiles/shape_time.py "$@" > gpurun_out/pmc_SF.log 2>&1 || exit 1
$T rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_SW -o pmc -- python3 profiles/shape_time.py "$@" > gpurun_out/pmc_SW.log 2>&1 || exit 1
python3 profiles/pmc_summary.py gpurun_out/pmc_SF gpurun_out/pmc_SW gpurun_out/${tag}_pmc_traffic.json > gpurun_out/${tag}_pmc_traffic.txt
cat gpurun_out/${tag}_pmc_traffic.txt
rm -rf gpurun_out/pmc_SF gpurun_out/pmc_SW
