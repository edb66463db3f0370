#!/bin/bash
# k_mips5 at 3 / 4 work-groups per compute unit, with and without the gate that holds a MIP pass behind the previous chain's head
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export MI_IPP_PROBES=1
for wpe in 4 3; do
  MI_NCC_MIPS_WPE=$wpe timeout -k 10 300 python profiles/r5_mips_knock.py 0 12 2>&1 | grep knock | sed "s/^/wpe $wpe /"
  for gate in 1 0; do
    echo "wpe $wpe gate $gate: $(MI_NCC_MIPS_WPE=$wpe MI_NCC_GATE=$gate timeout -k 5 300 python3 profiles/ncc_batch_probe.py 10 2>&1 | grep pairs)"
  done
done
