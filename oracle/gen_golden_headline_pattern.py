"""Whole-grid pattern datum at BASELINE configs[2] size, made by running the REFERENCE on all 1 971 054 tetrahedra.

TEST INFRASTRUCTURE; build container only (minutes of host time, ~32 GB of RAM):
    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/repo/oracle/shim:/root/reference/src:/root/repo \
      python /root/repo/oracle/gen_golden_headline_pattern.py [n_side = 69] [num_subproblems = 12]
The grid is ``bench.make_problem(n_side)`` -- handed to the reference as a ``pp.Grid`` over the SAME topology and geometry
arrays the device gets (tests/_reference_patch_script.grid_of), so faces and cells carry the same numbers on both sides.
``pp.Mpfa("flow").discretize`` (numerics/fv/mpfa.py:65-508, ``partition_arguments={"num_subproblems": k}``, python
inverter) leaves the flux matrix; stored in tests/golden/headline_flux_pattern_<n>.npz:
  row_len      uint8 per face: entries the reference STORES in that row of ``flux`` (scipy drops exact zeros)
  digest_rows  uint64: order-independent digest of (row, column) over all stored entries of the rows that are not
               Neumann boundary rows (sum of splitmix64 hashes, modulo 2^64) -- ``headline_digest`` below
  neumann_row  packed bits: rows of Neumann boundary faces (their true entries are all zero: what either side stores
               there is cancellation noise, the reference's stored noise is a subset of the structural stencil)
  totals       nnz of flux / bound_flux / vector_source, seconds of the run
and (round 5, late) a VALUE datum: ``flux_value_digest`` / ``bound_flux_value_digest`` (3 x 1024: per block of consecutive
rows sum |a|, sum a^2, sum a w(column)), ``pressure_digest`` / ``pressure_norm`` of the field the reference's own
``assemble_matrix_rhs`` + scipy BiCGStab (rtol 1e-13) give on that grid.
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


def headline_digest(indptr, indices, rows_mask) -> int:
    """Sum (mod 2^64) of splitmix64((row << 32) | column) over the entries of the rows with rows_mask -- the same
    function the GPU test applies to the device's index arrays."""
    indptr = np.asarray(indptr, dtype=np.int64)
    total = np.uint64(0)
    n = indptr.size - 1
    step = 1 << 18
    with np.errstate(over="ignore"):
        for r0 in range(0, n, step):
            r1 = min(n, r0 + step)
            lens = np.diff(indptr[r0:r1 + 1])
            rows = np.repeat(np.arange(r0, r1, dtype=np.uint64), lens)
            keep = np.repeat(rows_mask[r0:r1], lens)
            cols = np.asarray(indices[indptr[r0]:indptr[r1]], dtype=np.uint64)
            z = ((rows << np.uint64(32)) | cols)[keep]
            z = z + np.uint64(0x9E3779B97F4A7C15)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
            total = total + z.sum(dtype=np.uint64)
    return int(total)


from bench import N_BLOCKS, value_digest, vector_digest  # noqa: E402  (the digest functions live with their consumer)


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
    F = sps.csr_matrix(md["flux"])
    F.sort_indices()
    nf = gr.num_faces
    sides = np.bincount(raw["cf_indices"], minlength=nf)
    neumann_row = (sides == 1) & ~bcr.is_dir
    row_len = np.diff(F.indptr)
    assert row_len.max() < 256
    dig = headline_digest(F.indptr, F.indices, ~neumann_row)
    # ---- value datum (round 5, late): block digests of flux / bound_flux, and the pressure field of the assembled system
    t3 = time.perf_counter()
    disc = pp.Mpfa("flow")
    A, b = disc.assemble_matrix_rhs(gr, data)  # (fv_elliptic.py:67-112: div @ flux, -div @ bound_flux @ bc values)
    b = b + np.asarray(src, dtype=float)       # the integrated source of make_problem, as the device's assemble adds it
    A = sps.csr_matrix(A)
    import scipy.sparse.linalg as spla

    d = A.diagonal()
    M = spla.LinearOperator(A.shape, lambda v: v / d)
    its = [0]
    p, flag = spla.bicgstab(A, b, rtol=1e-13, atol=0.0, maxiter=20000, M=M, callback=lambda _x: its.__setitem__(0, its[0] + 1))
    res = float(np.linalg.norm(b - A @ p) / np.linalg.norm(b))
    t4 = time.perf_counter()
    out = {
        "flux_value_digest": value_digest(F, rows_mask=~neumann_row),
        "bound_flux_value_digest": value_digest(md["bound_flux"]),
        # (the other four matrices of the discretization: pressure traces and the vector-source terms; the rows of
        # vector_source at Neumann boundary faces are cancellation noise like those of flux)
        "bound_pressure_cell_value_digest": value_digest(md["bound_pressure_cell"]),
        "bound_pressure_face_value_digest": value_digest(md["bound_pressure_face"]),
        "vector_source_value_digest": value_digest(md["vector_source"], rows_mask=~neumann_row),
        "bound_pressure_vector_source_value_digest": value_digest(md["bound_pressure_vector_source"]),
        "pressure_digest": vector_digest(p),
        "pressure_norm": np.array([float(np.linalg.norm(p))]),
        "solve": np.array(json.dumps({"iterations": its[0], "flag": int(flag), "true_rel_residual": res,
                                      "solver": "scipy BiCGStab + Jacobi, rtol 1e-13", "seconds": t4 - t3,
                                      "system_nnz": int(A.nnz)})),
        "row_len": row_len.astype(np.uint8),
        "neumann_row": np.packbits(neumann_row),
        "digest_rows": np.array([dig], dtype=np.uint64),
        "totals": np.array(json.dumps({
            "n_side": n_side, "cells": int(gr.num_cells), "faces": int(nf), "num_subproblems": num_sub,
            "flux_nnz": int(F.nnz), "flux_nnz_outside_neumann_rows": int(row_len[~neumann_row].sum()),
            "bound_flux_nnz": int(md["bound_flux"].nnz), "vector_source_nnz": int(md["vector_source"].nnz),
            "stored_exact_zeros_in_flux": int((F.data == 0).sum()),
            "discretize_s": t2 - t1, "grid_s": t1 - t0, "porepy_from": os.path.dirname(pp.__file__)})),
    }
    path = os.path.join(out_dir or os.path.join(ROOT, "tests", "golden"), f"headline_flux_pattern_{n_side}.npz")
    np.savez_compressed(path, **out)
    print(out["totals"], os.path.getsize(path) / 1e6, "MB", flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 69, int(sys.argv[2]) if len(sys.argv) > 2 else 12,
         sys.argv[3] if len(sys.argv) > 3 else None)
