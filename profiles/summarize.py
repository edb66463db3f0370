"""Dumps the per-kernel summary (calls, total/avg duration) of a rocprofv3 rocpd database to CSV:
    python profiles/summarize.py gpurun_out/prof_r01/r01_results.db profiles/r01_bench_c3_kernel_stats.csv
(the same numbers `rocprofv3 --kernel-trace --stats` prints; durations in milliseconds)."""
import csv
import sqlite3
import sys


def main(db, out):
    c = sqlite3.connect(db)
    rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ms", "avg_ms", "percent"])
        for name, calls, total, avg, pct in rows:
            short = name if len(name) < 160 else name[:157] + "..."
            w.writerow([short, calls, round(total / 1e3, 2), round(avg / 1e3, 3), round(pct, 2)])
    print(f"{len(rows)} kernels -> {out}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
