"""Median duration of every (kernel, grid) of a `rocprofv3 --kernel-trace --output-format csv` run, for variant scans.
    python profiles/kernel_durations.py <dir> [name filter]"""
import csv
import glob
import os
import re
import statistics
import sys

path = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
by = {}
with open(path) as f:
    for r in csv.DictReader(f):
        m = re.search(r"\b(k_[a-z0-9_]+)", r["Kernel_Name"])
        name = m.group(1) if m else r["Kernel_Name"][:40]
        if flt and not re.search(flt, name):
            continue
        key = (name, f'{r["Grid_Size_X"]}x{r["Grid_Size_Y"]}x{r["Grid_Size_Z"]}/{r["Workgroup_Size_X"]} lds {r["LDS_Block_Size"]}')
        by.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (name, grid), d in sorted(by.items()):
    print(f"  {name:18s} {grid:34s} n={len(d):3d}  median {statistics.median(d):8.1f} us  min {min(d):8.1f}  all " + " ".join(f"{v:.0f}" for v in sorted(d)))
