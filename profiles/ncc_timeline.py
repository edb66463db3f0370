"""Per-queue timeline of the NCC batch from a `rocprofv3 --kernel-trace --output-format csv` run of profiles/ncc_batch_probe.py:
for the LAST batch call in the trace, busy time per queue, the union of both, and how much of the MIP pass ran while the table /
lag / refinement chain did.
    python profiles/ncc_timeline.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import os
import re
import sys

path = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[0]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        m = re.search(r"\b(k_[a-z0-9_]+)", r["Kernel_Name"])
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], m.group(1) if m else r["Kernel_Name"][:40]))
rows.sort()
# batch calls are separated by idle gaps > 200 us
calls, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if b[0] - max(r[1] for r in cur) > 200_000:
        calls.append(cur)
        cur = []
    cur.append(b)
calls.append(cur)
ncc = [c for c in calls if any("k_mips" in r[3] for r in c)]
print(f"{len(calls)} bursts of kernels, {len(ncc)} with k_mips; analysing the last one")
c = ncc[-1]
t0 = min(r[0] for r in c)
t1 = max(r[1] for r in c)


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for a, b in iv:
        if cs is None:
            cs, ce = a, b
        elif a <= ce:
            ce = max(ce, b)
        else:
            tot += ce - cs
            cs, ce = a, b
    return tot + (ce - cs if cs is not None else 0)


mip = [(r[0], r[1]) for r in c if "k_mips" in r[3]]
lag = [(r[0], r[1]) for r in c if "k_mips" not in r[3]]
print(f"span {(t1 - t0) / 1e3:.1f} us; kernels {len(c)}; MIP kernels busy {union(mip) / 1e3:.1f} us, chain kernels busy {union(lag) / 1e3:.1f} us, "
      f"either {union(mip + lag) / 1e3:.1f} us -> both at once {(union(mip) + union(lag) - union(mip + lag)) / 1e3:.1f} us")
by_q = {}
for r in c:
    by_q.setdefault(r[2], []).append((r[0], r[1]))
for q, iv in by_q.items():
    print(f"  queue {q}: {len(iv)} kernels, busy {union(iv) / 1e3:.1f} us, first start {(min(a for a, _ in iv) - t0) / 1e3:.1f} us, last end {(max(b for _, b in iv) - t0) / 1e3:.1f} us")
tot = {}
for r in c:
    k = r[3]
    tot.setdefault(k, [0, 0])
    tot[k][0] += 1
    tot[k][1] += r[1] - r[0]
for k, (n, d) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"  {k:>42}: {n:4d} launches, {d / 1e3:9.1f} us in all, {d / n / 1e3:7.1f} us each")
