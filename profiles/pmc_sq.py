"""Per-kernel averages of the counters of one or more `rocprofv3 --pmc ... --output-format csv` passes.
usage: python profiles/pmc_sq.py <dir> [<dir> ...]   (each dir holds a *counter_collection.csv)"""
import collections
import csv
import os
import re
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for folder in sys.argv[1:]:
    for root, _, files in os.walk(folder):
        for f in files:
            if not f.endswith("counter_collection.csv"):
                continue
            for r in csv.DictReader(open(os.path.join(root, f))):
                n = r["Kernel_Name"]
                if "anonymous namespace)::k_" not in n and "mi::" not in n:
                    continue
                m = re.search(r"\b(k_[a-z0-9_]+(<[^>]*>)?)", n)
                short = m.group(1) if m else n.replace("void ", "").replace("mi::(anonymous namespace)::", "").split("(")[0]
                agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print(f"    {c:28s} {sum(v) / len(v):16.0f}   (n={len(v)}, largest {max(v):.0f})")
