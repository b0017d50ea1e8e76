"""Ordered kernel timeline of the LAST step of tools/run_step.py from a rocprofv3 --kernel-trace csv:
python tools/step_timeline.py <dir> [min_us]  -> one line per dispatch (start offset, duration, gap before, kernel), phases
cut at the marker kernels; dispatches shorter than min_us are summed into the next printed line."""
import csv
import glob
import re
import sys

d = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
path = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step starts at the last dispatch of the first kernel of build_topology
first = [i for i, r in enumerate(rows) if "build_topology" in r["Kernel_Name"] and "#1}" in r["Kernel_Name"]]
i0 = first[-1]
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
small_n, small_t = 0, 0.0


def short(n):
    n = re.sub(r"void pfv::k_(wave_for|parallel_for|block_for)<", r"\1<", n)
    n = re.sub(r"\(pfv::pfv_ctx_impl&[^)]*\)", "()", n)
    n = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::", "rocprim::", n)
    return n[:150]


busy = 0.0
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur = (e - s) / 1e3
    busy += dur
    gap = (s - prev_end) / 1e3
    prev_end = max(prev_end, e)
    if dur < min_us and gap < 20.0:
        small_n += 1
        small_t += dur
        continue
    extra = f"  (+{small_n} short: {small_t:.0f} us)" if small_n else ""
    small_n, small_t = 0, 0.0
    print(f"{(s - t0) / 1e3:10.1f} us  {dur:9.1f} us  gap {gap:7.1f}  {short(r['Kernel_Name'])}{extra}")
print(f"span {(prev_end - t0) / 1e6:.2f} ms, kernel time {busy / 1e3:.2f} ms, {len(rows) - i0} dispatches")
