#!/bin/bash
# work-group shape of the lag transforms (MI_NCC_FFT_SHAPE=<threads>,<butterflies per thread>): per-kernel durations on the C5 grid
# with the MIP passes run first (nothing beside the chain).   usage: bash profiles/lag_shape_probe.sh <out.txt> [shapes...]
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export MI_NCC_SERIAL_MIPS=1
: > "$out"
for shape in "$@"; do
  rm -rf gpurun_out/lagshape
  MI_NCC_FFT_SHAPE=$shape timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lagshape -o t -- python3 profiles/ncc_batch_probe.py 1 > gpurun_out/lagshape.log 2>&1 || { echo "shape $shape: failed" >> "$out"; tail -n 5 gpurun_out/lagshape.log >> "$out"; continue; }
  echo "shape $shape: $(grep pairs gpurun_out/lagshape.log)" >> "$out"
  python3 profiles/kernel_durations.py gpurun_out/lagshape 'k_lag_(fwd|inv|mac)' >> "$out"
done
rm -rf gpurun_out/lagshape
cat "$out"
