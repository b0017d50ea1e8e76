"""Time the MPSA path on BASELINE config C4-like grids (structured tets, mu = lambda = 1,
rollers on the low faces, unit traction on top)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import porepy_amd as pa
n = int(sys.argv[1]) if len(sys.argv) > 1 else 44
g = pa.StructuredTetrahedralGrid([n, n, n], [1, 1, 1]); g.compute_geometry()
g = pa.perturb_interior_nodes(g, 0.2 / n)
nd, nc, nf = 3, g.num_cells, g.num_faces
C = pa.FourthOrderTensor(np.ones(nc), np.ones(nc))
bc = pa.BoundaryConditionVectorial(g)
bf = g.get_all_boundary_faces(); fc = g.face_centers
for axis in range(3):
    roll = bf[fc[axis, bf] < 1e-9]; bc.is_dir[axis, roll] = True; bc.is_neu[axis, roll] = False
bv = np.zeros((3, nf)); top = bf[fc[2, bf] > 1 - 1e-9]; bv[2, top] = -g.face_areas[top]
ctx = pa.Context(0); ctx.set_grid(pa.grid_to_raw(g))
ctx.mpsa_set_params(C.values, g.cell_volumes, bc.is_dir, bc.is_neu, 1.0 / 3.0)
for it in range(2):
    t = time.perf_counter(); ctx.mpsa_discretize(rebuild_topology=True); ctx.sync(); dt = time.perf_counter() - t
    st = ctx.stats()
    print(f"cells {nc}: discretize {dt*1e3:.1f} ms  (topology {st['topology_ms']:.1f} symbolic {st['symbolic_ms']:.1f} node {st['node_ms']:.1f} face {st['face_ms']:.1f})  {nc/dt/1e6:.2f} Mcells/s", flush=True)
t = time.perf_counter(); ctx.mpsa_assemble(bv.ravel("F"), None); ctx.sync(); print("assemble ms", (time.perf_counter()-t)*1e3, ctx.stats()["assemble_ms"])
pre = sys.argv[2] if len(sys.argv) > 2 else "amg"
u, info = ctx.solve("bicgstab", rtol=1e-10, maxit=50000, n=3*nc, raise_on_fail=False, precond=pre)
st = ctx.stats()
print("solve", pre, info, {k: st[k] for k in st if k.startswith("amg")})
u = u.reshape(3, -1, order="F"); cc = g.cell_centers; E, nu = 2.5, 0.25
print("max error vs exact uniaxial solution", np.max(np.abs(u - np.vstack((nu*cc[0]/E, nu*cc[1]/E, -cc[2]/E)))))
print("A nnz", ctx.matrix_info(11))
