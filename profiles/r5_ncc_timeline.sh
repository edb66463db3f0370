#!/bin/bash
# kernel timeline of one batch call, product build (default stream layout), plus a per-kernel list of the last call in time order
tag=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ncc_trace -o ncc -- python3 profiles/ncc_batch_probe.py 1 > gpurun_out/ncc_trace.log 2>&1 &&
python3 profiles/ncc_timeline.py gpurun_out/ncc_trace > gpurun_out/${tag}_ncc_timeline_default.txt && cat gpurun_out/${tag}_ncc_timeline_default.txt
python3 - <<'PY' > gpurun_out/${tag}_ncc_sequence.txt
import csv, glob, re
path = sorted(glob.glob("gpurun_out/ncc_trace/**/*kernel_trace.csv", recursive=True))[0]
rows = []
for r in csv.DictReader(open(path)):
    m = re.search(r"\b(k_[a-z0-9_]+)", r["Kernel_Name"])
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], m.group(1) if m else r["Kernel_Name"][:30]))
rows.sort()
calls, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if b[0] - max(r[1] for r in cur) > 200_000:
        calls.append(cur); cur = []
    cur.append(b)
calls.append(cur)
c = [x for x in calls if any("k_mips" in r[3] for r in x)][-1]
t0 = c[0][0]
for s, e, q, k in c:
    print(f"{(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f} us  ({(e - s) / 1e3:7.1f})  queue {q}  {k}")
PY
cat gpurun_out/${tag}_ncc_sequence.txt
rm -rf gpurun_out/ncc_trace
