cd $GRAFT_REPO_ROOT
for p in 1 0; do echo "==== priorities $p"; MI_IPP_PROBES=1 MI_NCC_PRIORITIES=$p python3 profiles/ncc_batch_probe.py 10; done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ncc_trace -o ncc -- python3 profiles/ncc_batch_probe.py 1 > gpurun_out/ncc_trace.log 2>&1 && python3 profiles/ncc_timeline.py gpurun_out/ncc_trace | head -20; rm -rf gpurun_out/ncc_trace
