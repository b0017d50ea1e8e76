"""Iteration counts of the AMG-preconditioned BiCGStab under different settings of the hierarchy, on the
host-emulation build (the aggregation is deterministic and the same on the device):
python tools/amg_compare.py "PFV_AMG_PASSES_COARSE=3" "PFV_AMG_PASSES_COARSE=2" -- 16 24 32"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import porepy_amd as pa
from tests import _parity as P

args = sys.argv[1:]
cut = args.index("--") if "--" in args else len(args)
variants = args[:cut] or [""]
sizes = [int(a) for a in args[cut + 1:]] or [12]
lib = P.emulation_library()
for n in sizes:
    for name in ("aniso", "iso", "mpsa"):
        if name == "mpsa" and n > 16:
            continue
        if name == "aniso":
            g, K, bc, bv, src = bench.make_problem(n)
        else:
            g = pa.StructuredTetrahedralGrid([n] * 3, [1.0] * 3)
            g.compute_geometry()
            if name == "mpsa":
                g = pa.perturb_interior_nodes(g, 0.2 / n)
            K = pa.SecondOrderTensor(np.ones(g.num_cells))
            bf = g.get_all_boundary_faces()
            bc = pa.BoundaryCondition(g, bf, ["dir"] * bf.size)
            bv = np.zeros(g.num_faces); bv[bf] = g.face_centers[0, bf]
            src = np.zeros(g.num_cells)
        ctx = pa.Context(0, lib)
        ctx.set_grid(pa.grid_to_raw(g))
        if name == "mpsa":
            nc, nf = g.num_cells, g.num_faces
            C = pa.FourthOrderTensor(np.ones(nc), np.ones(nc))
            bcv = pa.BoundaryConditionVectorial(g)
            bf = g.get_all_boundary_faces()
            for axis in range(3):
                roll = bf[g.face_centers[axis, bf] < 1e-9]
                bcv.is_dir[axis, roll] = True
                bcv.is_neu[axis, roll] = False
            bvv = np.zeros((3, nf))
            top = bf[g.face_centers[2, bf] > 1 - 1e-9]
            bvv[2, top] = -g.face_areas[top]
            ctx.mpsa_set_params(C.values, g.cell_volumes, bcv.is_dir, bcv.is_neu, 1.0 / 3.0)
        else:
            ctx.set_params(K.values, pa.bc_flags(bc), None, 1.0 / 3.0)
        for v in variants:
            for kv in v.split():
                k, val = kv.split("=")
                os.environ[k] = val
            t0 = time.time()
            if name == "mpsa":
                ctx.mpsa_discretize(rebuild_topology=False)
                ctx.mpsa_assemble(bvv.ravel("F"), None)
                x, info = ctx.solve("bicgstab", rtol=1e-10, maxit=3000, n=3 * g.num_cells, raise_on_fail=False, precond="amg")
            else:
                ctx.discretize(skip_vector_source=True)
                ctx.assemble(bv, None, src)
                x, info = ctx.solve("bicgstab", rtol=1e-10, maxit=3000, raise_on_fail=False, precond="amg")
            st = ctx.stats()
            print(f"n={n} {name:5s} cells {g.num_cells:7d} [{v}]: its {info['iterations']:3d} conv {info['converged']} levels {st['amg_levels']} "
                  f"cx {st['amg_operator_complexity']:.3f} coarsest {st['amg_coarsest_rows']} ({time.time()-t0:.1f} s)", flush=True)
            for kv in v.split():
                os.environ.pop(kv.split("=")[0], None)
