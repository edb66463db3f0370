"""Device-busy timeline from a `rocprofv3 --kernel-trace --memory-copy-trace --output-format csv` run: per 250-ms bin the share of
time at least one kernel was running, the share with a memory copy in flight, and the top kernels of the bin.
    python profiles/busy_timeline.py <dir> [bin_ms]"""
import csv
import glob
import os
import re
import sys

d = sys.argv[1]
bin_ns = int(float(sys.argv[2]) * 1e6) if len(sys.argv) > 2 else 250_000_000


def load(pattern, name_col):
    out = []
    for path in glob.glob(os.path.join(d, "**", pattern), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                out.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get(name_col, "")))
    return sorted(out)


def union_in(iv, a, b):
    tot, cs, ce = 0, None, None
    for s, e, _ in iv:
        s, e = max(s, a), min(e, b)
        if s >= e:
            continue
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    return tot + (ce - cs if cs is not None else 0)


k = load("*kernel_trace.csv", "Kernel_Name")
m = load("*memory_copy_trace.csv", "Direction")
t0, t1 = min(x[0] for x in k + m), max(x[1] for x in k + m)
print(f"{len(k)} kernels, {len(m)} copies over {(t1 - t0) / 1e9:.2f} s; kernels busy {union_in(k, t0, t1) / 1e9:.2f} s, copies busy {union_in(m, t0, t1) / 1e9:.2f} s")
a = t0
while a < t1:
    b = min(a + bin_ns, t1)
    kk = [x for x in k if x[1] > a and x[0] < b]
    mm = [x for x in m if x[1] > a and x[0] < b]
    top = {}
    for s, e, n in kk:
        nm = re.search(r"\b(k_[a-z0-9_]+|[a-z_]*rocclr[A-Za-z_]*|fft_rtc[a-z0-9_]*|transpose[a-z0-9_]*)", n)
        nm = nm.group(1)[:28] if nm else n[:28]
        top[nm] = top.get(nm, 0) + min(e, b) - max(s, a)
    tops = ", ".join(f"{n} {v / 1e6:.0f}" for n, v in sorted(top.items(), key=lambda kv: -kv[1])[:3])
    print(f"{(a - t0) / 1e9:6.2f} s: kernels {100 * union_in(kk, a, b) / (b - a):5.1f} %  copies {100 * union_in(mm, a, b) / (b - a):5.1f} %   {tops}")
    a = b
