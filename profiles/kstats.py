"""Prints the top kernels of a rocprofv3 rocpd database (helper for tuning runs)."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
for name, calls, total, avg, pct in c.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit ?", (n,)):
    short = name.replace("void mi::(anonymous namespace)::", "").replace("mi::(anonymous namespace)::", "")
    print(f"{short[:44]:44s} calls={calls:3d} avg_ms={avg/1e3:8.3f} pct={pct:5.1f}")
