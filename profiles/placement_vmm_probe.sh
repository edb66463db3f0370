#!/bin/bash
# the spectrum arrays built from physical chunks mapped in a chosen order (HIP virtual-memory API, probe build:
# MI_FFT_VMM=<order>[,<chunk MB>]) against the plain allocation; every line is a fresh process.
#   bash profiles/placement_vmm_probe.sh > gpurun_out/r04_placement_vmm.txt
cd "$GRAFT_REPO_ROOT" || exit 1
B="bench.py --steps 20 --warmup 5 --no-ncc --no-cpu-baseline --no-stages"
run() {
  env MI_IPP_PROBES=1 "$@" timeout -k 5 150 python3 $B 2>gpurun_out/vmm.err | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('  %.2f ms/it  mode %s  passes %s' % (d['ms_per_step'], r.get('pass_mode'), r['pass_ms']))
"
}
for rep in 1 2 3 4 5; do echo "plain hipMalloc (run $rep)"; run MI_DUMMY=1; done
# (chunks of the allocation granularity -- 10.8 thousand handles for the 21.6-GB range -- did not finish within seven minutes)
for chunk in 1024 256 64; do
  for order in 0 1 2 3; do
    for rep in 1 2; do echo "MI_FFT_VMM=$order,$chunk (run $rep)"; run MI_FFT_VMM=$order,$chunk; tail -2 gpurun_out/vmm.err | grep -i -E "error|fail|Traceback" ; done
  done
done
