"""rocprofv3 (--kernel-trace --stats) results .db -> per-kernel CSV summary for profiles/.
usage: python scripts/rocprof_summary.py gpurun_out/prof1/r1_results.db profiles/r01_kernel_stats.csv"""
import csv
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
    for name, calls, tot, avg, pct in rows:
        w.writerow([name, calls, round(tot, 3), round(avg, 3), round(pct, 3)])
print(f"wrote {len(rows)} kernels to {out}")
