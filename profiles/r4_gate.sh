cd $GRAFT_REPO_ROOT
for g in 1 0; do for u in "" 1; do echo "== MI_NCC_GATE=$g PROBE_U16=$u"; MI_IPP_PROBES=1 MI_NCC_GATE=$g PROBE_U16=$u python3 profiles/ncc_batch_probe.py 10 2>&1 | grep pairs; done; done
