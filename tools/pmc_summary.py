"""Summarise rocprofv3 --pmc CSV output: per kernel (short name) the mean and the maximum of each counter over its
dispatches.  The interaction-region kernel has one row per launch class (node64_40 = the 8 x 8 lane-grid class of interior
nodes of a tetrahedral grid, node32, node16, ...), the symbolic kernels one row each."""
import collections
import csv
import glob
import re
import sys


def short_name(name: str):
    if "k_spmv_win" in name:
        return "win " + name.split("k_spmv_win")[1].split(">")[0] + ">"
    if "k_amg_spmv" in name:
        return "amg_spmv_smooth" if "true" in name.split("k_amg_spmv")[1][:40] else "amg_spmv"
    if "k_spmv" in name:
        return "spmv"
    if "launch_node_class" in name or "node_body" in name:
        m = re.search(r"launch_node_class_reg<\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+)", name)
        if m:
            return f"node{m.group(1)}_{m.group(3)}_gj{m.group(4)}"
        m = re.search(r"launch_node_class_regILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)", name)  # mangled form
        if m:
            return f"node{m.group(1)}_{m.group(3)}_gj{m.group(4)}"
        return "node_other"
    if "run_face" in name or "k_face_pipe" in name:
        return "face"
    if "symbolic_face_rows" in name:
        return "symb_face_rows"
    if "symbolic_cell_rows" in name:
        return "symb_cell_rows"
    if "build_symbolic" in name and "wave_for" in name:
        return "symb"
    if "assemble_system" in name:
        return "assemble"
    if "amg_galerkin" in name and "wave_for<64" in name:
        return "galerkin64"
    if "mpsa_node" in name or "mpsa_run_node" in name:
        return "mpsa_node"
    return None


def main(root):
    for path in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        with open(path) as f:
            for row in csv.DictReader(f):
                short = short_name(row.get("Kernel_Name", ""))
                if short is None:
                    continue
                agg[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print(path.split("/")[-2:])
        for k, d in sorted(agg.items()):
            n = max(len(v) for v in d.values())
            print("  ", k, f"dispatches={n}", {c: f"mean {sum(v) / len(v):.4g} max {max(v):.4g}" for c, v in d.items()})


if __name__ == "__main__":
    main(sys.argv[1])
