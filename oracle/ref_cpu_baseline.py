"""The REFERENCE's own CPU path, timed (bench.py: cpu_baseline, kind "reference").

TEST INFRASTRUCTURE; run as a subprocess with the reference importable (``oracle.ref_env()``):
    python oracle/ref_cpu_baseline.py N_SIDE [NUM_SUBPROBLEMS [SOLVE_CAP_S]]
NUM_SUBPROBLEMS > 1: ``partition_arguments={"num_subproblems": k}`` (mpfa.py:160-161, 246-372: the reference's own
memory-bounded split -- what SURVEY 8(d) prescribes for configs[2] / [3] sizes); SOLVE_CAP_S bounds the Krylov solve
(the line then reports the residual reached).
Times, on one grid object and with the reference's own classes only,
  (i)   ``pp.Mpfa("flow").discretize(g, data)`` with ``mpfa_inverter="python"`` (numba is absent)
        — /root/reference/src/porepy/numerics/fv/mpfa.py:65-508;
  (ii)  ``assemble_matrix_rhs`` — numerics/fv/fv_elliptic.py:67-112;
  (iii) the linear solve.  The reference's model solves directly (models/solution_strategy.py:830-884:
        pypardiso, else scipy ``spsolve``); SuperLU does not finish a 3-D system of this size (BASELINE.md:
        > 20 min at 197 k cells), pypardiso is absent, so above ``DIRECT_MAX`` cells the system is solved with
        scipy BiCGStab + Jacobi to rtol 1e-10, and the line says so.
The workload is the bench's headline family (bench.py: make_problem): structured tetrahedra, interior nodes
perturbed, full-tensor anisotropic heterogeneous K, Dirichlet p = x on the x-faces, Neumann elsewhere, unit
source.  Prints one line ``RESULT {json}``.
"""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sps
import scipy.sparse.linalg as spla

import porepy as pp

DIRECT_MAX = 30000


def main(n_side: int, num_sub: int = 1, solve_cap_s: float = 0.0):
    t00 = time.perf_counter()
    g = pp.StructuredTetrahedralGrid([n_side] * 3, [1.0, 1.0, 1.0])
    g.compute_geometry()
    rng = np.random.default_rng(1)
    x = g.nodes.copy()
    lo, hi = x.min(axis=1, keepdims=True), x.max(axis=1, keepdims=True)
    tol = 1e-9 * (hi - lo)
    interior = np.all((x > lo + tol) & (x < hi - tol), axis=0)
    x[:, interior] += (rng.random((3, int(interior.sum()))) - 0.5) * (0.2 / n_side)
    g.nodes = x
    g.compute_geometry()
    nc = g.num_cells
    scale = np.exp(0.5 * np.random.default_rng(2).standard_normal(nc))
    K = pp.SecondOrderTensor(kxx=1.0 * scale, kyy=10.0 * scale, kzz=0.1 * scale, kxy=0.5 * scale,
                             kxz=0.05 * scale, kyz=0.2 * scale)
    bf = g.get_all_boundary_faces()
    xf = g.face_centers[0, bf]
    dirf = bf[(xf < 1e-9) | (xf > 1 - 1e-9)]
    bc = pp.BoundaryCondition(g, dirf, ["dir"] * dirf.size)
    bv = np.zeros(g.num_faces)
    bv[dirf] = g.face_centers[0, dirf]
    params = {"second_order_tensor": K, "bc": bc, "bc_values": bv, "mpfa_inverter": "python"}
    if num_sub > 1:
        params["partition_arguments"] = {"num_subproblems": int(num_sub)}
    data = pp.initialize_data({}, "flow", params)
    t_grid = time.perf_counter() - t00
    discr = pp.Mpfa("flow")
    c0 = time.process_time()
    t0 = time.perf_counter()
    discr.discretize(g, data)
    t1 = time.perf_counter()
    A, b = discr.assemble_matrix_rhs(g, data)
    b = b + g.cell_volumes
    t2 = time.perf_counter()
    if nc <= DIRECT_MAX:
        p = spla.spsolve(A.tocsc(), b)
        solver, its = "scipy spsolve (SuperLU), as models/solution_strategy.py:873", 0
    else:
        A = A.tocsr()
        M = sps.diags(1.0 / A.diagonal())
        its = 0

        class _Cap(Exception):
            pass

        last = [None]

        def cb(xk):
            nonlocal its
            its += 1
            last[0] = xk
            if solve_cap_s > 0 and time.perf_counter() - t2 > solve_cap_s:
                raise _Cap()

        try:
            p, flag = spla.bicgstab(A, b, rtol=1e-10, atol=0.0, maxiter=20000, M=M, callback=cb)
        except _Cap:
            p, flag = np.array(last[0], copy=True), f"stopped at the {solve_cap_s:.0f} s cap"
        solver = f"scipy BiCGStab+Jacobi rtol 1e-10 (flag {flag}); the reference's direct solve does not finish at this size"
    t3 = time.perf_counter()
    cpu_s = time.process_time() - c0  # all threads of this process: cpu_s / wall = threads effectively busy
    res = float(np.linalg.norm(b - A @ p) / np.linalg.norm(b))
    try:
        import resource
        rss_gb = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6
    except Exception:
        rss_gb = None
    out = {"cells": int(nc), "n_side": n_side, "grid_s": t_grid, "discretize_s": t1 - t0, "assemble_s": t2 - t1,
           "solve_s": t3 - t2, "cpu_s": cpu_s, "effective_threads": cpu_s / (t3 - t0), "solver": solver, "iterations": its, "rel_residual": res,
           "flux_nnz": int(data[pp.DISCRETIZATION_MATRICES]["flow"]["flux"].nnz),
           "p_norm": float(np.linalg.norm(p)), "peak_rss_gb": rss_gb,
           "host_cores": os.cpu_count(), "porepy_from": os.path.dirname(pp.__file__),
           "numpy": np.__version__, "num_subproblems": int(num_sub),
           "threads_env": {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS")}}
    print("RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 32, int(sys.argv[2]) if len(sys.argv) > 2 else 1,
         float(sys.argv[3]) if len(sys.argv) > 3 else 0.0)
