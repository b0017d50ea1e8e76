"""Stage timing of the MPSA node kernel: PFV_MPSA_ABLATE=k leaves the kernel after stage k
(1 sub-cell setup, 2 row assembly + scaling, 3 Gauss-Jordan, 4 AG/PLA/PLAG, 5 Et/Ptot/Etb/Pb, 0 all)."""
import os, subprocess, sys
n = sys.argv[1] if len(sys.argv) > 1 else "24"
code = r'''
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import porepy_amd as pa
n = int(sys.argv[1])
g = pa.StructuredTetrahedralGrid([n, n, n], [1, 1, 1]); g.compute_geometry()
g = pa.perturb_interior_nodes(g, 0.2 / n)
nc, nf = g.num_cells, g.num_faces
C = pa.FourthOrderTensor(np.ones(nc), np.ones(nc))
bc = pa.BoundaryConditionVectorial(g)
bf = g.get_all_boundary_faces(); fc = g.face_centers
for axis in range(3):
    roll = bf[fc[axis, bf] < 1e-9]; bc.is_dir[axis, roll] = True; bc.is_neu[axis, roll] = False
ctx = pa.Context(0); ctx.set_grid(pa.grid_to_raw(g))
ctx.mpsa_set_params(C.values, g.cell_volumes, bc.is_dir, bc.is_neu, 1.0 / 3.0)
best = 1e9
for it in range(3):
    ctx.mpsa_discretize(rebuild_topology=(it == 0)); ctx.sync()
    best = min(best, ctx.stats()["node_ms"])
print("ABLATE", os.environ.get("PFV_MPSA_ABLATE", "0"), "cells", nc, "node_ms", round(best, 2), "face_ms", round(ctx.stats()["face_ms"], 2), flush=True)
'''
for k in ("1", "2", "3", "4", "5", "0"):
    env = dict(os.environ, PFV_MPSA_ABLATE=k)
    subprocess.run([sys.executable, "-c", code, n], env=env, check=False)
