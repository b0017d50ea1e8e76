"""Fine whole-grid VALUE datum for MPSA (round 6; VERDICT r5 item 1c: "a per-block max |a|"): the REFERENCE run on every
cell of the grids of oracle/gen_golden_mpsa_whole_grid.py (same problem: ``tests._parity.mpsa_whole_grid_problem``; same
grid object over the device's own arrays; ``pp.Mpsa`` with the python inverter, sub-problems as there), and for each of
its FOUR matrices, per block of 256 consecutive rows: sum |a| and max |a| (``bench.fine_digest``; rows of ``stress`` that
belong to Neumann components of boundary faces left out as in the first datum -- their true entries are all zero).
-> tests/golden/mpsawhole_fine_<n>.npz

TEST INFRASTRUCTURE; build container only (n = 12: 1 min; n = 20: 3 min; n = 32 with 6 sub-problems: 16 min; n = 44 with 16: 51 min):
    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/repo/oracle/shim:/root/reference/src:/root/repo \
      python /root/repo/oracle/gen_golden_mpsa_fine.py [n_side = 12] [num_subproblems = 1]
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sps

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

KEYS = ("stress", "bound_stress", "bound_displacement_cell", "bound_displacement_face")


def main(n: int = 12, num_sub: int = 1):
    import porepy as pp

    import _reference_patch_script as rps
    import bench
    import porepy_amd as pa
    from tests._parity import mpsa_stress_rows_that_count, mpsa_whole_grid_problem

    g, mu, lam, is_dir, is_neu, bvf = mpsa_whole_grid_problem(n)
    raw = pa.grid_to_raw(g)
    gr = rps.grid_of(raw)
    bc = pp.BoundaryConditionVectorial(gr)
    bc.is_dir, bc.is_neu = is_dir.copy(), is_neu.copy()
    data = pp.initialize_data({}, "mechanics", {"fourth_order_tensor": pp.FourthOrderTensor(mu, lam), "bc": bc,
                                                "bc_values": bvf, "inverter": "python", "mpsa_eta": 1.0 / 3.0})
    if num_sub > 1:
        data[pp.PARAMETERS]["mechanics"]["partition_arguments"] = {"num_subproblems": int(num_sub)}
    t1 = time.perf_counter()
    pp.Mpsa("mechanics").discretize(gr, data)
    t2 = time.perf_counter()
    md = data[pp.DISCRETIZATION_MATRICES]["mechanics"]
    rows = mpsa_stress_rows_that_count(raw, is_neu)
    out = {k + "_fine": bench.fine_digest(sps.csr_matrix(md[k]), rows_mask=rows if k == "stress" else None) for k in KEYS}
    out["info"] = np.array(json.dumps({"n_side": n, "num_subproblems": num_sub, "cells": int(gr.num_cells),
                                       "faces": int(gr.num_faces), "rows_per_block": bench.FINE_ROWS,
                                       "discretize_s": t2 - t1, "porepy_from": os.path.dirname(pp.__file__)}))
    path = os.path.join(ROOT, "tests", "golden", f"mpsawhole_fine_{n}.npz")
    np.savez_compressed(path, **out)
    print(out["info"], os.path.getsize(path) / 1e3, "KB", flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 12, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
