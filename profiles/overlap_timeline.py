"""Timeline of `profiles/overlap_probe.py trace` under `rocprofv3 --kernel-trace --output-format csv`: for every variant the start and
end of the stand-in kernel (k_busy, second stream) and of part 2 of the x pass (k_x_fused_pipe, launch stream), in microseconds
from the start of the stand-in.
    python profiles/overlap_timeline.py <dir with *_kernel_trace.csv> > profiles/r03_overlap_timeline.txt
"""
import csv
import glob
import os
import sys

path = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[0]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        if "k_busy" in name or "k_x_fused_pipe" in name:
            rows.append(("busy" if "k_busy" in name else "x", int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                         int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0), int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)))
rows.sort(key=lambda r: r[1])
# the probe's loop order (busy > 0 only; every variant = warm-up + one repetition, the repetition is printed)
variants = [(b, us, dyn, free) for b, us in ((8, 300), (8, 1500), (32, 300), (32, 1500)) for dyn in (False, True) for free in (0, 8, 32)]
groups, cur = [], None
for kind, t0, t1, wg, grid in rows:
    if kind == "busy":
        cur = {"busy": (t0, t1, grid // max(wg, 1)), "x": []}
        groups.append(cur)
    elif cur is not None:
        cur["x"].append((t0, t1, grid // max(wg, 1)))
groups = [g for g in groups if g["x"]]
print(f"{os.path.basename(path)}: {len(groups)} stand-in launches with an x launch behind them (2 per variant: warm-up, repetition)")
print(f"{'stand-in':>16} {'tiles':>8} {'free CUs':>8} | {'busy end':>9} | {'x WGs':>6} {'x start':>8} {'x end':>8} {'x dur':>8}   (us from the stand-in's start)")
for i, v in enumerate(variants):
    k = 2 * i + 1
    if k >= len(groups):
        break
    g = groups[k]
    b0, b1, bw = g["busy"]
    x0, x1, xw = g["x"][0]
    print(f"{v[0]:>3} WGs x {v[1]:>4} us {'dynamic' if v[2] else 'static':>8} {v[3]:>8} | {(b1 - b0) / 1e3:9.1f} | {xw:>6} {(x0 - b0) / 1e3:8.1f} "
          f"{(x1 - b0) / 1e3:8.1f} {(x1 - x0) / 1e3:8.1f}")
