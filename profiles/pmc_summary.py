"""Builds profiles/<tag>_pmc_traffic.json from two rocprofv3 counter passes of the SAME bench command:
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_FETCH_SIZE -o pmc -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_WRITE_SIZE -o pmc -- python3 bench.py ...
    python profiles/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE profiles/r01_pmc_traffic.json
(separate passes: FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2 -- MI355X_MICROARCH.md "rocprofv3 PMC slots").
Units and the gfx950 correction follow MI355X_MICROARCH.md section HBM: both counters are in KiB; FETCH_SIZE reports
exactly half of the bytes of a wide coalesced read stream, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane
stores.  Values are averages per launch."""
import collections
import csv
import json
import os
import sys


def per_kernel(folder):
    agg = collections.defaultdict(list)
    path = [os.path.join(folder, f) for f in os.listdir(folder) if f.endswith("counter_collection.csv")][0]
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "mi::" not in n and "anonymous namespace)::k_" not in n:
            continue
        short = n.replace("void ", "").replace("mi::(anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
        agg[short].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


def main(fetch_dir, write_dir, out):
    f, nf = per_kernel(fetch_dir)
    w, _ = per_kernel(write_dir)
    res = {}
    for k in sorted(f):
        fetch_b = f[k] * 1024 * 2      # KiB -> B, x2 gfx950 correction for wide coalesced reads
        write_b = w.get(k, 0.0) * 1024
        res[k] = {"launches_sampled": nf[k], "fetch_bytes_corrected": fetch_b, "write_bytes": write_b,
                  "hbm_bytes_per_launch": fetch_b + write_b}
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE x2 (gfx950); per launch",
               "kernels": res}, open(out, "w"), indent=1)
    for k, v in res.items():
        print(f"{k:36s} fetch {v['fetch_bytes_corrected']/1e9:7.2f} GB  write {v['write_bytes']/1e9:7.2f} GB  total {v['hbm_bytes_per_launch']/1e9:7.2f} GB")


if __name__ == "__main__":
    main(*sys.argv[1:4])
