"""Iteration counts of the AMG-preconditioned BiCGStab with the handshake rounds choosing among all
neighbours (PFV_AMG_CAND=0) or among the K strongest (4, 8).  Runs on the host-emulation build (the
aggregation is deterministic and the same on the device): python tools/amg_cand_compare.py 12 16 24"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import porepy_amd as pa
from tests import _parity as P

lib = P.emulation_library()
for n in [int(a) for a in sys.argv[1:]] or [12]:
    for name in ("aniso", "iso"):
        if name == "aniso":
            g, K, bc, bv, src = bench.make_problem(n)
            eta = 1.0 / 3.0
        else:
            g = pa.StructuredTetrahedralGrid([n] * 3, [1.0] * 3)
            g.compute_geometry()
            K = pa.SecondOrderTensor(np.ones(g.num_cells))
            bf = g.get_all_boundary_faces()
            bc = pa.BoundaryCondition(g, bf, ["dir"] * bf.size)
            bv = np.zeros(g.num_faces); bv[bf] = g.face_centers[0, bf]
            src = np.zeros(g.num_cells); eta = 1.0 / 3.0
        ctx = pa.Context(0, lib)
        ctx.set_grid(pa.grid_to_raw(g))
        ctx.set_params(K.values, pa.bc_flags(bc), None, eta)
        for k in (0, 4, 8):
            os.environ["PFV_AMG_CAND"] = str(k)
            ctx.discretize(skip_vector_source=True)  # new matrix -> new hierarchy
            ctx.assemble(bv, None, src)
            t0 = time.time()
            x, info = ctx.solve("bicgstab", rtol=1e-10, maxit=3000, raise_on_fail=False, precond="amg")
            st = ctx.stats()
            print(f"n={n} {name:5s} cells {g.num_cells:7d} K={k}: its {info['iterations']:3d} conv {info['converged']} levels {st['amg_levels']} "
                  f"cx {st['amg_operator_complexity']:.3f} coarsest {st['amg_coarsest_rows']} ({time.time()-t0:.1f} s)", flush=True)
