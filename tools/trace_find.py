"""Dispatches of kernels matching a substring in a rocprofv3 kernel trace, with their neighbours:
python tools/trace_find.py <dir>/<prefix>_kernel_trace.csv <substring> [min_us]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2]
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 100.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.replace("pfv::", "").replace("void ", "")[:90]
for i, r in enumerate(rows):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if pat in r["Kernel_Name"] and d >= min_us:
        prev = short(rows[i - 1]["Kernel_Name"]) if i else ""
        nxt = short(rows[i + 1]["Kernel_Name"]) if i + 1 < len(rows) else ""
        print(f"{d:9.1f} us grid {r.get('Grid_Size', '?'):>10}  after [{prev}]  before [{nxt}]")
