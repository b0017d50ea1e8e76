"""Prototype (scipy, CPU; decides a round-5 item, nothing of the product): what rigid-body-mode aware aggregation would
buy the MPSA solve.  The MPSA system of the BASELINE configs[3] family (structured tetrahedra, rollers + top traction) is
assembled by the library (host-emulation build); then two aggregation-AMG preconditioners built here in scipy are
compared inside BiCGStab (rtol 1e-10) on it:

  translations : the tentative prolongator carries the 3 translations per aggregate (piecewise constant per component --
                 what csrc/amg.inc does for block systems: coarse block size 3);
  rigid bodies : the 6 rigid-body modes per aggregate (3 translations + 3 rotations about the aggregate's centre,
                 orthonormalised per aggregate: coarse block size 6).

Same aggregates (greedy, on the cell graph of the block strength), same smoother (one damped block-Jacobi sweep before
and after), same V-cycle, unsmoothed prolongators, direct solve on the coarsest level.

    python tools/amg_rbm_prototype.py [n_side]
"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sps
import scipy.sparse.linalg as spla

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import porepy_amd as pa  # noqa: E402
from tests import _parity as P  # noqa: E402


def mpsa_system(n):
    g = pa.StructuredTetrahedralGrid([n, n, n], [1, 1, 1])
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
    bv = np.zeros((3, nf))
    top = bf[fc[2, bf] > 1 - 1e-9]
    bv[2, top] = -g.face_areas[top]
    ctx = pa.Context(0, P.emulation_library())
    ctx.set_grid(pa.grid_to_raw(g))
    ctx.mpsa_set_params(C.values, g.cell_volumes, bc.is_dir, bc.is_neu, 1.0 / 3.0)
    ctx.mpsa_discretize(rebuild_topology=False)
    ctx.mpsa_assemble(bv.ravel("F"), None)
    A = ctx.matrix(pa._lib.MAT_MECH_SYSTEM).tocsr()
    b = ctx.rhs()
    return g, A, np.asarray(b)


def cell_strength(A, nc):
    """|| . ||_F of the 3 x 3 blocks: a cell graph."""
    coo = A.tocoo()
    S = sps.coo_matrix((coo.data ** 2, (coo.row // 3, coo.col // 3)), shape=(nc, nc)).tocsr()
    S.data = np.sqrt(S.data)
    return S


def aggregate(S, theta=0.08):
    """Greedy aggregation (root + its strong neighbours, leftovers join the strongest neighbouring aggregate)."""
    n = S.shape[0]
    d = S.diagonal()
    agg = -np.ones(n, dtype=int)
    indptr, indices, data = S.indptr, S.indices, S.data
    strong = [indices[indptr[i]:indptr[i + 1]][(data[indptr[i]:indptr[i + 1]] >= theta * np.sqrt(abs(d[i] * d[indices[indptr[i]:indptr[i + 1]]]))) &
                                                (indices[indptr[i]:indptr[i + 1]] != i)] for i in range(n)]
    na = 0
    for i in range(n):
        if agg[i] < 0 and all(agg[j] < 0 for j in strong[i]):
            agg[i] = na
            agg[strong[i]] = na
            na += 1
    for i in range(n):
        if agg[i] < 0:
            cand = [j for j in strong[i] if agg[j] >= 0]
            if cand:
                agg[i] = agg[cand[0]]
            else:
                agg[i] = na
                na += 1
    return agg, na


def tentative(agg, na, B, bs_fine):
    """Block prolongator from near-null-space vectors B (n_fine_dofs x k): per aggregate the QR of its rows of B."""
    k = B.shape[1]
    rows, cols, vals = [], [], []
    Bc = np.zeros((na * k, k))
    order = np.argsort(agg, kind="stable")
    bounds = np.searchsorted(agg[order], np.arange(na + 1))
    for a in range(na):
        cells = order[bounds[a]:bounds[a + 1]]
        dofs = (cells[:, None] * bs_fine + np.arange(bs_fine)[None, :]).ravel()
        Q, R = np.linalg.qr(B[dofs])
        kk = Q.shape[1]
        if kk < k:  # (an aggregate with fewer dofs than modes)
            Q = np.hstack((Q, np.zeros((Q.shape[0], k - kk))))
            R = np.vstack((R, np.zeros((k - kk, k))))
        rows.append(np.repeat(dofs, k))
        cols.append(np.tile(a * k + np.arange(k), dofs.size))
        vals.append(Q.ravel())
        Bc[a * k:(a + 1) * k] = R
    Pm = sps.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(B.shape[0], na * k))
    return Pm, Bc


def block_jacobi_inverse(A, bs):
    n = A.shape[0] // bs
    D = np.zeros((n, bs, bs))
    coo = A.tocoo()
    m = (coo.row // bs) == (coo.col // bs)
    D[coo.row[m] // bs, coo.row[m] % bs, coo.col[m] % bs] = coo.data[m]
    Di = np.linalg.inv(D + 1e-300 * np.eye(bs))
    return sps.block_diag([sps.csr_matrix(x) for x in Di], format="csr") if n < 2000 else sps.bsr_matrix(
        (Di, np.arange(n), np.arange(n + 1)), shape=A.shape).tocsr()


def hierarchy(A, B, bs, coarsest=600, omega=0.7):
    levels = []
    while True:
        nc = A.shape[0] // bs
        lev = {"A": A, "Dinv": block_jacobi_inverse(A, bs), "omega": omega}
        levels.append(lev)
        if A.shape[0] <= coarsest or len(levels) >= 8:
            lev["lu"] = spla.splu(sps.csc_matrix(A))
            return levels
        S = cell_strength(A, nc) if bs == 3 else cell_strength_bs(A, nc, bs)
        agg, na = aggregate(S)
        if na >= nc:
            lev["lu"] = spla.splu(sps.csc_matrix(A))
            return levels
        Pm, Bc = tentative(agg, na, B, bs)
        lev["P"] = Pm
        A = sps.csr_matrix(Pm.T @ A @ Pm)
        B = Bc
        bs = B.shape[1]


def cell_strength_bs(A, nc, bs):
    coo = A.tocoo()
    S = sps.coo_matrix((coo.data ** 2, (coo.row // bs, coo.col // bs)), shape=(nc, nc)).tocsr()
    S.data = np.sqrt(S.data)
    return S


def vcycle(levels, l, r):
    lev = levels[l]
    if "lu" in lev:
        return lev["lu"].solve(r)
    x = lev["omega"] * (lev["Dinv"] @ r)
    rc = lev["P"].T @ (r - lev["A"] @ x)
    x = x + lev["P"] @ vcycle(levels, l + 1, rc)
    return x + lev["omega"] * (lev["Dinv"] @ (r - lev["A"] @ x))


def run(A, b, B, label):
    t = time.perf_counter()
    lv = hierarchy(A, B, 3)
    opc = sum(L["A"].nnz for L in lv) / A.nnz
    its = [0]
    M = spla.LinearOperator(A.shape, lambda r: vcycle(lv, 0, r))
    x, info = spla.bicgstab(A, b, rtol=1e-10, atol=0.0, maxiter=400, M=M, callback=lambda _x: its.__setitem__(0, its[0] + 1))
    res = np.linalg.norm(b - A @ x) / np.linalg.norm(b)
    print(f"{label:13s}: levels {len(lv)} (rows {[L['A'].shape[0] for L in lv]}), operator complexity {opc:.2f}, "
          f"BiCGStab iterations {its[0]}, residual {res:.1e}, flag {info}  ({time.perf_counter() - t:.1f} s)", flush=True)
    return its[0]


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    g, A, b = mpsa_system(n)
    nc = g.num_cells
    print(f"{nc} cells, {3 * nc} dofs, {A.nnz} entries", flush=True)
    cc = g.cell_centers
    T = np.zeros((3 * nc, 3))
    for a in range(3):
        T[a::3, a] = 1.0
    R = np.zeros((3 * nc, 3))
    x, y, z = cc
    # rotations about the axes: u = w x r
    R[0::3, 0], R[1::3, 0], R[2::3, 0] = 0.0, -z, y
    R[0::3, 1], R[1::3, 1], R[2::3, 1] = z, 0.0, -x
    R[0::3, 2], R[1::3, 2], R[2::3, 2] = -y, x, 0.0
    i3 = run(A, b, T, "translations")
    i6 = run(A, b, np.hstack((T, R)), "rigid bodies")
    print(f"iterations {i3} -> {i6}")
