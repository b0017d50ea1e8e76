"""Fine whole-grid value datum at BASELINE configs[2] size (round 6): the REFERENCE run once more on all
1 971 054 tetrahedra of ``bench.make_problem(69)`` (as oracle/gen_golden_headline_pattern.py does: same grid object over the
device's own arrays, ``pp.Mpfa`` with 12 sub-problems, python inverter), and for each of its SIX matrices, per block of 256
consecutive rows (15 511 blocks instead of the 1 024 of the first datum): sum |a| and max |a| (``bench.fine_digest``; Neumann
boundary rows of flux / vector_source left out as there).  -> tests/golden/headline_fine_digest_69.npz

TEST INFRASTRUCTURE; build container only (20 minutes of host time, ~32 GB of RAM):
    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/repo/oracle/shim:/root/reference/src:/root/repo \
      python /root/repo/oracle/gen_golden_headline_fine.py [n_side = 69] [num_subproblems = 12]
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


def main(n_side: int = 69, num_sub: int = 12, out_dir: str | None = None):
    import porepy as pp

    import _reference_patch_script as rps
    import bench
    import porepy_amd as pa

    t0 = time.perf_counter()
    g, K, bc, bv, src = bench.make_problem(n_side)
    raw = pa.grid_to_raw(g)
    gr = rps.grid_of(raw)
    bcr = pp.BoundaryCondition(gr)
    bcr.is_dir = np.asarray(bc.is_dir, bool).copy()
    bcr.is_neu = np.asarray(bc.is_neu, bool).copy()
    bcr.is_rob = np.zeros(gr.num_faces, bool)
    bcr.is_internal = np.zeros(gr.num_faces, bool)
    Kr = pp.SecondOrderTensor(np.ones(gr.num_cells))
    Kr.values = np.asarray(K.values, dtype=float).copy()
    params = {"second_order_tensor": Kr, "bc": bcr, "bc_values": bv, "mpfa_inverter": "python", "mpfa_eta": 1.0 / 3.0}
    if num_sub > 1:
        params["partition_arguments"] = {"num_subproblems": int(num_sub)}
    data = pp.initialize_data({}, "flow", params)
    t1 = time.perf_counter()
    pp.Mpfa("flow").discretize(gr, data)
    t2 = time.perf_counter()
    md = data[pp.DISCRETIZATION_MATRICES]["flow"]
    nf = gr.num_faces
    sides = np.bincount(raw["cf_indices"], minlength=nf)
    neumann_row = (sides == 1) & ~bcr.is_dir
    out = {}
    for name, msk in (("flux", ~neumann_row), ("bound_flux", None), ("bound_pressure_cell", None),
                      ("bound_pressure_face", None), ("vector_source", ~neumann_row), ("bound_pressure_vector_source", None)):
        out[name + "_fine"] = bench.fine_digest(sps.csr_matrix(md[name]), rows_mask=msk)
    out["info"] = np.array(json.dumps({"n_side": n_side, "cells": int(gr.num_cells), "faces": int(nf), "num_subproblems": num_sub,
                                       "rows_per_block": bench.FINE_ROWS, "discretize_s": t2 - t1, "grid_s": t1 - t0,
                                       "porepy_from": os.path.dirname(pp.__file__)}))
    path = os.path.join(out_dir or os.path.join(ROOT, "tests", "golden"), f"headline_fine_digest_{n_side}.npz")
    np.savez_compressed(path, **out)
    print(out["info"], os.path.getsize(path) / 1e6, "MB", flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 69, int(sys.argv[2]) if len(sys.argv) > 2 else 12,
         sys.argv[3] if len(sys.argv) > 3 else None)
