"""Top kernels of a rocprofv3 --kernel-trace --stats run: python tools/trace_top.py <dir>/<prefix>_kernel_stats.csv [n]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.2f} ms, {sum(int(r['Calls']) for r in rows)} dispatches")
for r in rows[:n]:
    print(f"{float(r['TotalDurationNs']) / 1e6:9.2f} ms {int(r['Calls']):6d} x {float(r['AverageNs']) / 1e3:9.1f} us  {r['Name'][:140]}")
