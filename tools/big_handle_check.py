"""One handle beyond 6.3 M cells: the grid of the headline family at n_side = 110 (7 986 000 tetrahedra;
vector_source with 2.72e9 entries -- its CSR arrays do not exist, csrc/topology.inc: vs_implicit) discretized on ONE handle,
and the rows of two pieces of a 24-way cell partition (cells + one node ring, distributed.extract_subdomain -- what the
split path computes) compared with the same rows fetched from the whole-grid handle by pfv_get_matrix_rows: all six matrices.
GPU box:   python tools/big_handle_check.py [n_side = 110] [pieces = 24]   (the step time of the same grid: bench.py --n-side 110)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import porepy_amd as pa  # noqa: E402
from porepy_amd import distributed as D  # noqa: E402


def main():
    import torch

    torch.zeros(1, device="cuda")  # (torch's HIP context first, as in bench.py: initialised after the library's it finds no device)
    n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 110
    nparts = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    M = pa._lib
    keys = {"flux": M.MAT_FLUX, "bound_flux": M.MAT_BOUND_FLUX, "bound_pressure_cell": M.MAT_BOUND_PRESSURE_CELL,
            "bound_pressure_face": M.MAT_BOUND_PRESSURE_FACE, "vector_source": M.MAT_VECTOR_SOURCE,
            "bound_pressure_vector_source": M.MAT_BOUND_PRESSURE_VECTOR_SOURCE}
    t0 = time.time()
    lp, Kvals, flags, bv, src, eta = bench.make_slab_problem(n_side, 0, 1)
    raw = lp.raw
    nc, nf = raw["cell_centers"].shape[1], raw["face_centers"].shape[1]
    out = {"n_side": n_side, "cells": int(nc), "faces": int(nf), "grid_seconds": round(time.time() - t0, 1)}
    ctx = pa.Context(0)
    ctx.set_grid(raw)
    ctx.set_params(Kvals, flags, None, eta)
    t1 = time.time()
    ctx.discretize(rebuild_topology=True)
    ctx.sync()
    out["discretize_first_call_seconds"] = round(time.time() - t1, 2)
    info = {k: ctx.matrix_info(w) for k, w in keys.items()}
    out["nnz"] = {k: int(v[2]) for k, v in info.items()}
    out["vector_source_entries_over_2_31"] = bool(info["vector_source"][2] >= 2 ** 31)
    st = ctx.stats()
    out["phases_ms"] = {k: round(float(st[k]), 2) for k in ("topology_ms", "symbolic_ms", "node_ms", "face_ms", "discretize_ms")}
    # ---- two pieces of the split path against the rows of the whole-grid handle
    box = (raw["face_centers"].min(axis=1), raw["face_centers"].max(axis=1))
    order = D.morton_order(raw["cell_centers"], 3, box)
    owner = np.empty(nc, dtype=np.int64)
    owner[order] = (np.arange(nc) * nparts) // nc
    worst = {k: 0.0 for k in keys}
    checked = 0
    for r in (0, nparts // 2 + 1):
        sub = D.extract_subdomain(raw, owner, r)
        c2 = pa.Context(0)
        c2.set_grid(sub.raw)
        lfl = flags[sub.face_gid].copy()
        lfl[sub.artificial_boundary] = M.BC_NEU
        c2.set_params(np.ascontiguousarray(Kvals[:, :, sub.cell_gid]), lfl, None, eta)
        c2.discretize(rebuild_topology=True)
        cfp = sub.raw["cf_indptr"]
        own_faces = np.unique(sub.raw["cf_indices"][: cfp[sub.n_own]])
        # a sample of the piece's own faces (the export is what takes the time, not the comparison)
        sel = own_faces[:: max(1, own_faces.size // 20000)]
        vcol = (3 * sub.cell_gid[:, None] + np.arange(3)[None, :]).ravel()
        for name, which in keys.items():
            P = c2.matrix_rows(which, sel).tocoo()
            cmap = vcol if "vector_source" in name else (sub.face_gid if name in ("bound_flux", "bound_pressure_face") else sub.cell_gid)
            import scipy.sparse as sps

            ncols = info[name][1]
            Pg = sps.coo_matrix((P.data, (P.row, cmap[P.col])), shape=(sel.size, ncols)).tocsr()
            W = ctx.matrix_rows(which, sub.face_gid[sel])
            scale = max(abs(W).max(), 1e-300)
            worst[name] = max(worst[name], float(abs(Pg - W).max() / scale))
        checked += int(sel.size)
        c2.close()
    out["rows_compared_with_the_split_path"] = checked
    out["worst_rel_diff_vs_split_path"] = worst
    print("PARTIAL " + json.dumps(out), flush=True)
    # ---- and the step: assemble + solve on the same handle
    dev = torch.device("cuda", 0)
    d_bv = torch.from_numpy(np.ascontiguousarray(bv)).to(dev)
    d_src = torch.from_numpy(np.ascontiguousarray(src)).to(dev)
    d_x = torch.zeros(nc, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    times = []
    for _ in range(3):
        ts = time.time()
        ctx.discretize(rebuild_topology=True)
        ctx.assemble_device(d_bv.data_ptr(), 0, d_src.data_ptr())
        sinfo = ctx.solve_device(d_x.data_ptr(), "bicgstab", rtol=1e-13, maxit=20000, raise_on_fail=False, precond="amg")
        ctx.sync()
        times.append(1e3 * (time.time() - ts))
    out["step_ms"] = [round(t, 1) for t in times]
    out["iterations"] = int(sinfo["iterations"])
    out["rel_residual"] = float(sinfo["rel_residual"])
    out["cells_per_s"] = float(nc / (min(times) * 1e-3))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
