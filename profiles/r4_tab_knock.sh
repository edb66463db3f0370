cd $GRAFT_REPO_ROOT
for k in 0 1 2 3; do echo "== knock $k"; MI_IPP_PROBES=1 MI_NCC_CHAIN_STREAMS=1 MI_NCC_TAB_KNOCK=$k bash profiles/r4_ncc_durations.sh r4_knock$k | grep -E "k_plane|k_band"; done
