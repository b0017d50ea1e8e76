"""The step after discretize on the headline grid: d(div q)/dp = Div @ Flux formed on the device (DeviceCsr) against the
same product by scipy on the host -- which first has to fetch the flux matrix over PCIe.

    python tools/csr_algebra_bench.py [n_side]
"""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sps

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
import porepy_amd as pa  # noqa: E402

n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 69
lp, Kvals, flags, bv, src, eta = bench.make_slab_problem(n_side, 0, 1)
ctx = pa.Context(0)
ctx.set_grid(lp.raw)
ctx.set_params(Kvals, flags, None, eta)
ctx.discretize(rebuild_topology=True)
ctx.assemble(bv, None, src)
ctx.sync()
nc, nf = lp.raw["cell_centers"].shape[1], lp.raw["face_centers"].shape[1]
# divergence = cell_faces^T (cells x faces, +-1)
cf_ptr, cf_idx, cf_sgn = lp.raw["cf_indptr"], lp.raw["cf_indices"], lp.raw["cf_sign"]
cell_faces = sps.csc_matrix((cf_sgn.astype(float), cf_idx, cf_ptr), shape=(nf, nc))
div_h = sps.csr_matrix(cell_faces.T)
out = {"n_side": n_side, "cells": int(nc), "faces": int(nf)}
t = time.perf_counter()
div = pa.DeviceCsr.from_scipy(div_h, ctx)
out["upload_div_ms"] = 1e3 * (time.perf_counter() - t)
t = time.perf_counter()
flux = pa.DeviceCsr.from_discretization(ctx, pa._lib.MAT_FLUX)
ctx.sync()
out["flux_device_to_device_ms"] = 1e3 * (time.perf_counter() - t)
out["flux_nnz"] = flux.nnz
for rep in range(3):
    t = time.perf_counter()
    J = div @ flux
    ctx.sync()
    out["device_matmul_ms"] = 1e3 * (time.perf_counter() - t)
    if rep < 2:
        J.close()
out["J_nnz"] = J.nnz
x = np.random.default_rng(0).random(nc)
A = ctx.matrix(pa._lib.MAT_SYSTEM)
y = J @ x
out["vs_library_system_matrix_rel"] = float(np.abs(y - A @ x).max() / np.abs(A @ x).max())
t = time.perf_counter()
flux_h = ctx.matrix(pa._lib.MAT_FLUX)
out["host_fetch_flux_ms"] = 1e3 * (time.perf_counter() - t)
t = time.perf_counter()
Jh = div_h @ flux_h
out["host_scipy_matmul_ms"] = 1e3 * (time.perf_counter() - t)
Jh.sort_indices()
Jd = J.to_scipy()
out["bit_identical_to_scipy"] = bool(np.array_equal(Jd.indptr, Jh.indptr) and np.array_equal(Jd.indices, Jh.indices)
                                     and np.array_equal(Jd.data, Jh.data))
print(json.dumps(out))
