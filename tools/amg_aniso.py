import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import porepy_amd as pa
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
kz = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-2
g = pa.StructuredTetrahedralGrid([n] * 3, [1, 1, 1]); g.compute_geometry(); g = pa.perturb_interior_nodes(g, 0.2 / n)
nc = g.num_cells
K = pa.SecondOrderTensor(kxx=np.ones(nc), kyy=np.ones(nc), kzz=kz * np.ones(nc)).values
bf = g.get_all_boundary_faces(); xf = g.face_centers[0, bf]; dirf = bf[(xf < 1e-9) | (xf > 1 - 1e-9)]
flags = np.zeros(g.num_faces, dtype=np.uint8); flags[bf] = 2; flags[dirf] = 1
bv = np.zeros(g.num_faces); bv[dirf] = 1 + g.face_centers[1, dirf]
ctx = pa.Context(0); ctx.set_grid(pa.grid_to_raw(g)); ctx.set_params(K, flags, None, 1 / 3); ctx.discretize(skip_vector_source=True); ctx.assemble(bv, None, g.cell_volumes)
b = ctx.rhs()
for meth, pre, env in (("bicgstab", "jacobi", {}), ("bicgstab", "amg", {}), ("gmres", "amg", {}), ("bicgstab", "amg", {"PFV_AMG_ALPHA_PCT": "100"}),
                       ("bicgstab", "amg", {"PFV_AMG_ALPHA_PCT": "100", "PFV_AMG_GAMMA": "1"}), ("bicgstab", "amg", {"PFV_AMG_OMEGA_PCT": "60", "PFV_AMG_ALPHA_PCT": "100"})):
    for k in ("PFV_AMG_ALPHA_PCT", "PFV_AMG_GAMMA", "PFV_AMG_OMEGA_PCT"):
        os.environ.pop(k, None)
    os.environ.update(env)
    ctx.assemble(bv, None, g.cell_volumes)
    x, info = ctx.solve(meth, rtol=1e-10, maxit=3000, precond=pre, raise_on_fail=False, restart=50)
    r = float(np.linalg.norm(b - ctx.spmv(pa._lib.MAT_SYSTEM, x)) / np.linalg.norm(b))
    print(f"n={n} kz={kz} {meth}+{pre} {env}: its {info['iterations']} res {r:.2e}", flush=True)
