"""Summarise rocprofv3 --pmc CSV output: per kernel (short name) average of each counter."""
import csv, glob, sys, collections
for path in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name", "")
            short = ("win " + name.split("k_spmv_win")[1].split(">")[0] + ">") if "k_spmv_win" in name else "amg_spmv_smooth" if "k_amg_spmv" in name and "true" in name.split("k_amg_spmv")[1][:40] else "amg_spmv" if "k_amg_spmv" in name else "spmv" if "k_spmv" in name else "node" if "node_body" in name or "launch_node" in name else \
                "face" if ("run_face" in name or "k_face_pipe" in name) else "symb" if "build_symbolic" in name and "wave_for" in name else \
                "assemble" if "assemble_system" in name else None
            if short is None: continue
            if short == "node" and "Li64E" not in name: continue
            agg[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print(path.split("/")[-2:])
    for k, d in agg.items():
        print("  ", k, {c: f"{max(v):.4g}" for c, v in d.items()})
