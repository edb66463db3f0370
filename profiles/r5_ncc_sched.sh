#!/bin/bash
# schedule variants of the 112-pair batch (probes build): pieces per group x work-groups per CU of the passes that run beside a chain x gate
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export MI_IPP_PROBES=1
for pieces in 1 2 4; do for wb in 3 2; do for gate in 0 1; do
  echo "pieces $pieces wpe-beside $wb gate $gate: $(MI_NCC_PIECES=$pieces MI_NCC_MIPS_WPE_BESIDE=$wb MI_NCC_GATE=$gate timeout -k 5 200 python3 profiles/ncc_batch_probe.py 10 2>&1 | grep pairs | cut -c1-60)"
done; done; done
