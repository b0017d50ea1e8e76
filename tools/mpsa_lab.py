"""MPSA interaction-region kernel at BASELINE configs[3] (511 104 tetrahedra): time of the kernel when it is left
after stage k (PFV_MPSA_ABLATE, set per run by tools/gpu_mpsa_lab.sh); prints node_ms."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import porepy_amd as pa  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 44
g = pa.StructuredTetrahedralGrid([n, n, n], [1.0, 1.0, 1.0])
g.compute_geometry()
g = pa.perturb_interior_nodes(g, 0.2 / n)
nc, nf = g.num_cells, g.num_faces
C = pa.FourthOrderTensor(np.ones(nc), np.ones(nc))
bc = pa.BoundaryConditionVectorial(g)
bf = g.get_all_boundary_faces()
fc = g.face_centers
for axis in range(3):
    roll = bf[fc[axis, bf] < 1e-9]
    bc.is_dir[axis, roll] = True
    bc.is_neu[axis, roll] = False
ctx = pa.Context(0)
ctx.set_grid(pa.grid_to_raw(g))
ctx.mpsa_set_params(C.values, g.cell_volumes, bc.is_dir, bc.is_neu, 1.0 / 3.0)
abl = os.environ.get("PFV_MPSA_ABLATE", "0")
try:
    ctx.mpsa_discretize(rebuild_topology=True)
except Exception as e:  # ablated kernels leave no usable tables; the timing is still there
    print("  (", type(e).__name__, ")")
ts = []
for _ in range(3):
    try:
        ctx.mpsa_discretize(rebuild_topology=False)
    except Exception:
        pass
    ts.append(ctx.stats()["node_ms"])
st = ctx.stats()
print(f"ablate {abl}: node_ms {min(ts):.2f}  face_ms {st['face_ms']:.2f}")
