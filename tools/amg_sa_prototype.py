"""Prototype (scipy, CPU; decides a round-5 item, nothing of the product): smoothed aggregation against the plain
(piecewise-constant) aggregation csrc/amg.inc uses, on the library's own MPFA flow system of the headline family
(perturbed tetrahedra, full-tensor anisotropic heterogeneous K; host-emulation build).  Same greedy aggregates, same
damped-Jacobi smoother (one sweep before and after), same V-cycle inside BiCGStab to 1e-10:

  plain    : P = tentative prolongator (one constant per aggregate)
  smoothed : P = (I - w D^-1 A_f) P_t, A_f = A with weak entries lumped to the diagonal, w = 2/3

    python tools/amg_sa_prototype.py [n_side]
"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sps
import scipy.sparse.linalg as spla

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import porepy_amd as pa  # noqa: E402
from tests import _parity as P  # noqa: E402


def flow_system(n):
    lp, K, flags, bv, src, eta = bench.make_slab_problem(n, 0, 1)
    ctx = pa.Context(0, P.emulation_library())
    ctx.set_grid(lp.raw)
    ctx.set_params(K, flags, None, eta)
    ctx.discretize(rebuild_topology=False)
    ctx.assemble(bv, None, src)
    return ctx.matrix(pa._lib.MAT_SYSTEM).tocsr(), np.asarray(ctx.rhs())


def strong_graph(A, theta):
    d = np.abs(A.diagonal())
    coo = A.tocoo()
    keep = (coo.row != coo.col) & (np.abs(coo.data) >= theta * np.sqrt(d[coo.row] * d[coo.col]))
    return sps.csr_matrix((np.abs(coo.data[keep]), (coo.row[keep], coo.col[keep])), shape=A.shape)


def aggregate(S):
    n = S.shape[0]
    agg = -np.ones(n, dtype=int)
    ip, ix = S.indptr, S.indices
    na = 0
    for i in range(n):
        nb = ix[ip[i]:ip[i + 1]]
        if agg[i] < 0 and np.all(agg[nb] < 0):
            agg[i] = na
            agg[nb] = na
            na += 1
    for i in range(n):
        if agg[i] < 0:
            nb = ix[ip[i]:ip[i + 1]]
            nb = nb[agg[nb] >= 0]
            if nb.size:
                agg[i] = agg[nb[np.argmax(S.data[ip[i]:ip[i + 1]][np.isin(ix[ip[i]:ip[i + 1]], nb)])]]
            else:
                agg[i] = na
                na += 1
    return agg, na


def hierarchy(A, smoothed, theta=0.08, coarsest=400, omega_p=2.0 / 3.0):
    levels = []
    while True:
        n = A.shape[0]
        lev = {"A": A, "dinv": 1.0 / A.diagonal()}
        levels.append(lev)
        if n <= coarsest or len(levels) >= 10:
            lev["lu"] = spla.splu(sps.csc_matrix(A))
            return levels
        S = strong_graph(A, theta)
        agg, na = aggregate(S)
        if na >= n:
            lev["lu"] = spla.splu(sps.csc_matrix(A))
            return levels
        cnt = np.bincount(agg, minlength=na).astype(float)
        Pt = sps.csr_matrix((1.0 / np.sqrt(cnt[agg]), (np.arange(n), agg)), shape=(n, na))
        if smoothed:
            # filtered matrix: weak entries lumped to the diagonal
            mask = strong_graph(A, theta)
            mask.data[:] = 1.0
            Aoff = A.multiply(mask).tocsr()
            dlump = np.asarray(A.sum(axis=1)).ravel() - np.asarray(Aoff.sum(axis=1)).ravel()
            Af = (Aoff + sps.diags(dlump)).tocsr()
            Pm = Pt - omega_p * (sps.diags(1.0 / Af.diagonal()) @ (Af @ Pt))
        else:
            Pm = Pt
        lev["P"] = sps.csr_matrix(Pm)
        lev["R"] = sps.csr_matrix(Pm.T)
        A = sps.csr_matrix(lev["R"] @ A @ lev["P"])


def vcycle(levels, l, r, omega=0.7):
    lev = levels[l]
    if "lu" in lev:
        return lev["lu"].solve(r)
    x = omega * lev["dinv"] * r
    x = x + lev["P"] @ vcycle(levels, l + 1, lev["R"] @ (r - lev["A"] @ x), omega)
    return x + omega * lev["dinv"] * (r - lev["A"] @ x)


def run(A, b, smoothed, label):
    t = time.perf_counter()
    lv = hierarchy(A, smoothed)
    opc = sum(L["A"].nnz for L in lv) / A.nnz
    its = [0]
    M = spla.LinearOperator(A.shape, lambda r: vcycle(lv, 0, r))
    x, info = spla.bicgstab(A, b, rtol=1e-10, atol=0.0, maxiter=500, M=M, callback=lambda _x: its.__setitem__(0, its[0] + 1))
    res = np.linalg.norm(b - A @ x) / np.linalg.norm(b)
    print(f"{label:9s}: levels {len(lv)} (rows {[L['A'].shape[0] for L in lv]}; entries per row {[round(L['A'].nnz / L['A'].shape[0], 1) for L in lv]}), "
          f"operator complexity {opc:.2f}, BiCGStab iterations {its[0]}, residual {res:.1e}, flag {info}  ({time.perf_counter() - t:.1f} s)", flush=True)
    return its[0]


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    t = time.perf_counter()
    A, b = flow_system(n)
    print(f"{A.shape[0]} cells, {A.nnz} entries ({time.perf_counter() - t:.1f} s to discretize on the emulation build)", flush=True)
    a = run(A, b, False, "plain")
    s = run(A, b, True, "smoothed")
    print(f"iterations {a} -> {s}")
