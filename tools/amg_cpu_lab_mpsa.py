"""CPU lab (host-emulation build): iteration counts of block-AMG BiCGStab on the MPSA workload (BASELINE configs[3]
family) for sweeps of PFV_AMG_* switches.   python tools/amg_cpu_lab_mpsa.py N_SIDE VAR=v1,v2 ..."""
import itertools
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import porepy_amd as pa  # noqa: E402
from tests import _parity as P  # noqa: E402

n = int(sys.argv[1])
sweeps = [(a.split("=")[0], a.split("=")[1].split(",")) for a in sys.argv[2:]]
g = pa.StructuredTetrahedralGrid([n, n, n], [1, 1, 1])
g.compute_geometry()
g = pa.perturb_interior_nodes(g, 0.2 / n)
nd, nc, nf = 3, g.num_cells, g.num_faces
C = pa.FourthOrderTensor(np.ones(nc), np.ones(nc))
bc = pa.BoundaryConditionVectorial(g)
bf = g.get_all_boundary_faces()
fc = g.face_centers
for axis in range(3):
    roll = bf[fc[axis, bf] < 1e-9]
    bc.is_dir[axis, roll] = True
    bc.is_neu[axis, roll] = False
bv = np.zeros((3, nf))
top = bf[fc[2, bf] > 1 - 1e-9]
bv[2, top] = -g.face_areas[top]
ctx = pa.Context(0, P.emulation_library())
ctx.set_grid(pa.grid_to_raw(g))
ctx.mpsa_set_params(C.values, g.cell_volumes, bc.is_dir, bc.is_neu, 1.0 / 3.0)
print(f"{nc} cells, {3 * nc} dofs", flush=True)
os.environ["PFV_AMG_REUSE"] = "0"
for combo in itertools.product(*[v for _, v in sweeps]):
    for (k, _), v in zip(sweeps, combo):
        os.environ[k] = v
    ctx.mpsa_discretize(rebuild_topology=False)
    ctx.mpsa_assemble(bv.ravel("F"), None)
    t0 = time.perf_counter()
    u, info = ctx.solve("bicgstab", rtol=1e-10, maxit=400, n=3 * nc, raise_on_fail=False, precond="amg")
    st = ctx.stats()
    u = u.reshape(3, -1, order="F")
    cc = g.cell_centers
    err = np.max(np.abs(u - np.vstack((0.1 * cc[0], 0.1 * cc[1], -0.4 * cc[2]))))
    print(" ".join(f"{k}={v}" for (k, _), v in zip(sweeps, combo)), "-> iterations", info["iterations"], "converged",
          info["converged"], f"res {info['rel_residual']:.1e} err {err:.1e} levels {int(st['amg_levels'])} opc "
          f"{st['amg_operator_complexity']:.3f} ({time.perf_counter() - t0:.1f} s)", flush=True)
