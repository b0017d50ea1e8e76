"""FETCH_SIZE / WRITE_SIZE passes of tools/gpu_pmc.sh -> profiles-style JSON keyed by the kernels bench.py reports.
usage: python tools/pmc_to_json.py <pmc dir> <out.json> [n_side]"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def short(name: str):
    """(class, variant) of a dispatch: the SpMV kernels run on every AMG level with their own template arguments --
    the variant with the most bytes per launch is the finest level's (picked below)."""
    if "k_spmv_win" in name:
        a = name.replace("k_spmv_win_pre<", "k_spmv_win<").split("k_spmv_win<")[1].split(">")[0].replace(" ", "")
        parts = a.split(",")  # L, U, value type, epilogue
        if parts[2] == "double" and parts[3] in ("2", "3"):
            return "krylov", a
        if parts[2] == "float" and parts[3] == "1":
            return "smooth", a
        return None
    if "k_face_pipe" in name:
        return "face", ""
    if "launch_node_class_reg<64, 3, 40" in name:
        return "node", ""
    return None


raw = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    with open(path) as f:
        for row in csv.DictReader(f):
            k = short(row.get("Kernel_Name", ""))
            if k and row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                raw[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
agg = {}
for (cls, variant), d in raw.items():
    fe = sorted(d.get("FETCH_SIZE", [0.0]))
    if cls not in agg or fe[len(fe) // 2] > agg[cls][0]:
        agg[cls] = (fe[len(fe) // 2], d, variant)
agg = {cls: v[1] for cls, v in agg.items()}
out = {"n_side": int(sys.argv[3]) if len(sys.argv) > 3 else 69, "source_hash": bench.source_hash(),
       "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/run_step.py; counters in KiB; "
              "FETCH doubled for kernels that stream with wide coalesced loads (MI355X_MICROARCH.md, gfx950 note: the counter "
              "tallies 128-byte requests at 64 B), as reported for the node kernel (scattered 8-byte reads); per launch = median "
              "over the launches of the run; FETCH counts L2 misses (Infinity-Cache hits included)",
       "kernels": {}}
# wide coalesced streams: doubled (un-doubled, the SpMV figure is below what the kernel must read); the node kernel
# reads scattered 8-byte words: as reported
doubled = {"krylov": True, "smooth": True, "face": True, "node": False}
for k, d in agg.items():
    fe = sorted(d.get("FETCH_SIZE", [0.0]))
    wr = sorted(d.get("WRITE_SIZE", [0.0]))
    f = fe[len(fe) // 2] * 1024.0 * (2.0 if doubled[k] else 1.0)
    w = wr[len(wr) // 2] * 1024.0
    out["kernels"][k] = {"fetch_bytes_per_launch": f, "write_bytes_per_launch": w, "traffic_bytes_per_launch": f + w,
                         "fetch_doubled": doubled[k], "launches_seen": len(fe)}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
