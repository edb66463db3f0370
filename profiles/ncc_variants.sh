#!/bin/bash
# NCC batch on the C5 grid under sets of environment switches, each with a kernel timeline.
#   usage: bash profiles/ncc_variants.sh <out.txt> "<VAR=val VAR=val>" ...
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
: > "$out"
for v in "$@"; do
  rm -rf gpurun_out/nccv
  echo "==== $v" >> "$out"
  echo "wall: $(env $v timeout -k 5 200 python3 profiles/ncc_batch_probe.py 20 2>&1 | grep pairs)" >> "$out"
  env $v timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/nccv -o t -- python3 profiles/ncc_batch_probe.py 1 > gpurun_out/nccv.log 2>&1
  python3 profiles/ncc_timeline.py gpurun_out/nccv | head -n 6 >> "$out"
done
rm -rf gpurun_out/nccv
cat "$out"
