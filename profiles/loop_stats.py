"""Kernel statistics of a `rocprofv3 --kernel-trace --output-format csv` run of bench.py, in two forms:
  * every launch of the process (calls, total, average per kernel) -> the CSV given as second argument -- what `--stats` prints;
  * the launches of the ITERATION LOOP alone (third argument, text): a pass kernel's launches whose predecessor in the trace is the
    kernel that precedes it in an iteration (x pass after the inverse y pass, y-forward after the x pass, z after y-forward,
    y-inverse after z).  The plan-time placement trial launches the x and y kernels 60 times on buffers that are given back, and
    mi_rl_time_pass launches each pass five times back to back: both are in the first form and in neither the bench's timed loop.
    python profiles/loop_stats.py <trace dir> <all.csv> <loop.txt>"""
import csv
import glob
import os
import re
import sys

path = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[0]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()


def short(name):
    m = re.search(r"(k_[a-z0-9_]+(<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]


tot = {}
for s, e, k in rows:
    t = tot.setdefault(k, [0, 0])
    t[0] += 1
    t[1] += e - s
grand = sum(v[1] for v in tot.values())
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "total_ms", "avg_ms", "percent"])
    for k, (n, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        w.writerow([k if len(k) < 160 else k[:157] + "...", n, round(t / 1e6, 2), round(t / n / 1e6, 3), round(100.0 * t / grand, 2)])

names = [short(k) for _, _, k in rows]
fam = lambda n: ("x" if n.startswith("k_x_fused_pipe") and n.endswith(", 0>") else "yf" if n.startswith(("k_y_pair", "k_y_pass")) and n.endswith("false>")
                 else "yi" if n.startswith(("k_y_pair", "k_y_pass")) and n.endswith("true>") else "z" if n.startswith(("k_z_pair_pipe", "k_z_conv_pipe")) else None)
prev_of = {"x": "yi", "yf": "x", "z": "yf", "yi": "z"}
loop = {}
passes = [i for i in range(len(rows)) if fam(names[i])]   # (helper launches between two passes -- counter resets -- do not count)
for a, i in zip(passes, passes[1:]):
    f = fam(names[i])
    if fam(names[a]) == prev_of[f]:
        loop.setdefault(names[i], []).append((rows[i][1] - rows[i][0]) / 1e6)
with open(sys.argv[3], "w") as out:
    out.write("# launches inside the iteration loop (predecessor rule, profiles/loop_stats.py): kernel, launches, average ms, min, max\n")
    for k, v in sorted(loop.items(), key=lambda kv: -sum(kv[1])):
        out.write(f"{k:45s} {len(v):4d}  avg {sum(v) / len(v):7.3f}  min {min(v):7.3f}  max {max(v):7.3f}\n")
    x = [v for k, v in loop.items() if k.startswith("k_x_fused_pipe")]
    if x:
        v = x[0]
        out.write(f"# x pass: even / odd launches of the loop (ratio / update step): {sum(v[0::2]) / len(v[0::2]):.3f} / {sum(v[1::2]) / len(v[1::2]):.3f} ms\n")
print(open(sys.argv[3]).read())
