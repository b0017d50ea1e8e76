"""Windowed vs plain SpMV on the benchmark matrix, in the grid's own cell numbering (tetrahedron type
major) and with the cells renumbered lattice-cell major (PERM=1) or along a Morton curve (PERM=2)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, porepy_amd as pa
n = int(os.environ.get("TUNE_N", "69"))
g, K, bc, bv, src = bench.make_problem(n)
mode = int(os.environ.get("PERM", "0"))
if mode:
    nc = g.num_cells
    if mode == 1:
        ncube = nc // 6
        perm = (np.arange(nc).reshape(6, ncube).T).ravel()  # new cell k = old cell perm[k]
    else:
        q = np.floor((g.cell_centers - g.cell_centers.min(1, keepdims=True)) /
                     (np.ptp(g.cell_centers, axis=1)[:, None] + 1e-12) * 1024).astype(np.int64)
        key = np.zeros(nc, dtype=np.int64)
        for b in range(10):
            for a in range(3):
                key |= ((q[a] >> b) & 1) << (3 * b + a)
        perm = np.argsort(key, kind="stable")
    g.cell_faces = g.cell_faces.tocsc()[:, perm]
    g.cell_centers = g.cell_centers[:, perm]
    g.cell_volumes = g.cell_volumes[perm]
    K.values = K.values[:, :, perm]
    src = src[perm]
ctx = pa.Context(0); ctx.set_grid(pa.grid_to_raw(g))
ctx.set_params(K.values, pa.bc_flags(bc), bc.robin_weight, pa.determine_eta(g))
ctx.discretize(rebuild_topology=True)
ctx.assemble(bv, None, src)
os.environ["PFV_DEBUG_WIN"] = "1"
for flag in ("1", "0"):
    os.environ["PFV_SPMV_WINDOW"] = flag
    print("PERM", mode, "window", flag, "spmv ms", ctx.time_kernel(0, 50), flush=True)
    x, info = ctx.solve(method="bicgstab", rtol=1e-10, maxit=200, precond="amg")
    x, info = ctx.solve(method="bicgstab", rtol=1e-10, maxit=200, precond="amg")
    st = ctx.stats()
    print(info, "amg smooth ms", ctx.time_kernel(3, 50), "setup", st["amg_setup_ms"],
          {k: round(st[k], 2) for k in ("topology_ms", "symbolic_ms", "node_ms", "face_ms", "assemble_ms")}, flush=True)
