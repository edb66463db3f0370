#!/bin/bash
# The two placements of the y passes seen from the L2's memory interface: per process (a fresh context each) the duration of
# k_y_pair<fwd/inv> and the EA counters of those launches -- requests, their summed residency (LEVEL / REQ = average latency in
# TCC cycles) (four TCC counters are what one pass collects).   bash profiles/ea_probe.sh <out.txt> [processes]
out=$1; n=${2:-5}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
B="bench.py --steps 4 --warmup 1 --no-ncc --no-cpu-baseline --no-stages"
: > "$out"
for i in $(seq 1 $n); do
  timeout -k 5 90 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum --kernel-trace --output-format csv -d gpurun_out/ea_$i -o ea -- python3 $B > gpurun_out/ea_$i.log 2>&1
  echo "== process $i" >> "$out"
  python3 - gpurun_out/ea_$i >> "$out" <<'PY'
import csv, os, re, sys, collections
folder = sys.argv[1]
dur = collections.defaultdict(list); cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for root, _, files in os.walk(folder):
    for f in files:
        p = os.path.join(root, f)
        if f.endswith("kernel_trace.csv"):
            for r in csv.DictReader(open(p)):
                dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        if f.endswith("counter_collection.csv"):
            for r in csv.DictReader(open(p)):
                cnt[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name in sorted(cnt):
    m = re.search(r"(k_y_pair<[^>]*>|k_x_fused_pipe<10, 1, 0>|k_z_pair_pipe<[^>]*>)", name)
    if not m:
        continue
    c = {k: sum(v) / len(v) for k, v in cnt[name].items()}
    d = dur.get(name, [0.0])
    rd, wr = c.get("TCC_EA0_RDREQ_sum", 0), c.get("TCC_EA0_WRREQ_sum", 0)
    print(f"  {m.group(1):46s} {sum(d) / len(d):6.3f} ms (under the profiler)  reads {rd / 1e6:7.1f} M, residency per read {c.get('TCC_EA0_RDREQ_LEVEL_sum', 0) / max(rd, 1):7.1f} cyc;  "
          f"writes {wr / 1e6:7.1f} M, residency per write {c.get('TCC_EA0_WRREQ_LEVEL_sum', 0) / max(wr, 1):7.1f} cyc")
PY
  rm -rf gpurun_out/ea_$i
done
cat "$out"
