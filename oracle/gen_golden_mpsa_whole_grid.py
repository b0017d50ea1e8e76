"""Whole-grid VALUE datum for MPSA (tests/golden/mpsawhole_<n>.npz), made by running the REFERENCE on every cell of a
perturbed tetrahedral box of the BASELINE configs[3] family (rollers on the low faces, traction on top; heterogeneous
Lame parameters): ``pp.Mpsa("mechanics").discretize`` (numerics/fv/mpsa.py:63-529, python inverter) +
``assemble_matrix_rhs`` (:488-529) + a scipy solve.  Stored: per block of consecutive rows (256 blocks) the digests of
bench.value_digest -- sum |a|, sum a^2, column-weighted sum -- of stress, bound_stress, bound_displacement_cell and
bound_displacement_face, and the displacement field (norm, per-block sums).  The grid is handed to the reference as a
``pp.Grid`` over the SAME arrays the device gets (tests/_reference_patch_script.grid_of).

TEST INFRASTRUCTURE; build container only:
    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/repo/oracle/shim:/root/reference/src:/root/repo \
      python /root/repo/oracle/gen_golden_mpsa_whole_grid.py [n_side = 16] [num_subproblems = 1]
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sps
import scipy.sparse.linalg as spla

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

KEYS = ("stress", "bound_stress", "bound_displacement_cell", "bound_displacement_face")
BLOCKS = 256


def main(n: int = 16, num_sub: int = 1):
    import porepy as pp

    import _reference_patch_script as rps
    import porepy_amd as pa
    from bench import value_digest, vector_digest
    from tests._parity import mpsa_whole_grid_problem

    t0 = time.perf_counter()
    g, mu, lam, is_dir, is_neu, bvf = mpsa_whole_grid_problem(n)
    raw = pa.grid_to_raw(g)
    gr = rps.grid_of(raw)
    bc = pp.BoundaryConditionVectorial(gr)
    bc.is_dir, bc.is_neu = is_dir.copy(), is_neu.copy()
    C = pp.FourthOrderTensor(mu, lam)
    data = pp.initialize_data({}, "mechanics", {"fourth_order_tensor": C, "bc": bc, "bc_values": bvf, "inverter": "python",
                                                "mpsa_eta": 1.0 / 3.0, "source": np.zeros(3 * gr.num_cells)})
    if num_sub > 1:  # (memory-bounded run of the reference: mpsa.py:201-207, _fvutils.subproblems)
        data[pp.PARAMETERS]["mechanics"]["partition_arguments"] = {"num_subproblems": int(num_sub)}
    d = pp.Mpsa("mechanics")
    t1 = time.perf_counter()
    d.discretize(gr, data)
    t2 = time.perf_counter()
    A, b = d.assemble_matrix_rhs(gr, data)
    A = sps.csr_matrix(A)
    dg = A.diagonal()
    its = [0]
    u, flag = spla.bicgstab(A, b, rtol=1e-13, atol=0.0, maxiter=50000, M=spla.LinearOperator(A.shape, lambda v: v / dg),
                            callback=lambda _x: its.__setitem__(0, its[0] + 1))
    res = float(np.linalg.norm(b - A @ u) / np.linalg.norm(b))
    t3 = time.perf_counter()
    md = data[pp.DISCRETIZATION_MATRICES]["mechanics"]
    from tests._parity import mpsa_stress_rows_that_count

    rows = mpsa_stress_rows_that_count(raw, is_neu)  # (Neumann components of boundary faces: true entries all zero)
    out = {k + "_digest": value_digest(md[k], BLOCKS, rows_mask=rows if k == "stress" else None) for k in KEYS}
    out["u_digest"] = vector_digest(u, BLOCKS)
    out["u_norm"] = np.array([float(np.linalg.norm(u))])
    out["info"] = np.array(json.dumps({
        "n_side": n, "num_subproblems": num_sub, "cells": int(gr.num_cells), "faces": int(gr.num_faces), "dofs": int(A.shape[0]),
        "nnz": {k: int(sps.csr_matrix(md[k]).nnz) for k in KEYS}, "system_nnz": int(A.nnz),
        "discretize_s": t2 - t1, "solve_s": t3 - t2, "grid_s": t1 - t0, "iterations": its[0], "flag": int(flag),
        "true_rel_residual": res, "solver": "scipy BiCGStab + Jacobi, rtol 1e-13"}))
    path = os.path.join(ROOT, "tests", "golden", f"mpsawhole_{n}.npz")
    np.savez_compressed(path, **out)
    print(out["info"], os.path.getsize(path) / 1e3, "KB", flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 16, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
