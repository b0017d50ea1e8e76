"""One AMG cycle of the traced step, dispatch by dispatch: python tools/cycle_slice.py <trace dir>"""
import csv, glob, re, sys
path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last full cycle: between the last two f64 products with dots (MODE 2 / 3)
idx = [i for i, r in enumerate(rows) if "k_spmv_win<16, 5, double, 3>" in r["Kernel_Name"] or "k_spmv_win<16, 5, double, 2>" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["Start_Timestamp"])
prev = t0
def short(n):
    n = re.sub(r"void pfv::k_(wave_for|parallel_for|block_for)<", r"\1<", n)
    n = re.sub(r"\(pfv::pfv_ctx_impl&[^)]*\)", "()", n)
    n = re.sub(r"::\{lambda\(pfv::WaveCtx const&\)", "::{wave", n)
    return n[:110]
tot = 0.0
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:8.1f} us {(e - s) / 1e3:7.1f} us gap {(s - prev) / 1e3:5.1f} grid {r.get('Grid_Size_X', r.get('Grid_Size','?')):>8} {short(r['Kernel_Name'])}")
    prev = e
    tot += (e - s) / 1e3
print(f"{b - a} dispatches, kernel time {tot:.1f} us, span {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")
