#!/bin/bash
# NCC batch on the C5 grid: order of MIP passes and chains.   usage: bash profiles/ncc_gate_probe.sh <out.txt>
out=$1
cd "$GRAFT_REPO_ROOT" || exit 1
: > "$out"
for round in 1 2; do
for v in "MI_NCC_GATE=1" "MI_NCC_GATE=0" "MI_NCC_SERIAL_MIPS=1"; do
  echo "$v: $(env $v timeout -k 5 200 python3 profiles/ncc_batch_probe.py 20 2>&1 | grep pairs)" >> "$out"
done
done
cat "$out"
