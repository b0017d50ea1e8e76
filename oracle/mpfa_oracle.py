"""CPU oracle for the MPFA-O hot path — TEST INFRASTRUCTURE ONLY.

This is a node-centric numpy restatement of what the reference computes with ~50
global sparse products.  Nothing in ``porepy_amd/`` may import this module; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
use it, and only as the checker.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks this oracle against
matrices produced by the reference itself (``oracle/gen_golden.py`` imports
``/root/reference/src`` through ``oracle/shim`` and stores them under
``tests/golden/``), including the reference's own known-answer cases.

Reference code followed (paths relative to /root/reference/src/porepy):
  * sub-half-face enumeration (cell, face, node)      numerics/fv/_fvutils.py:51-172
  * nK product, n_h = face_normal / #nodes(face)      numerics/fv/_fvutils.py:697-762
  * continuity point / distance d_h, eta rule         numerics/fv/_fvutils.py:222-277
  * default eta from the grid name                    numerics/fv/_fvutils.py:280-305
  * local system rows (flux / Robin / pressure)       numerics/fv/mpfa.py:809-997
  * row 1-norm scaling before the inverse             numerics/fv/mpfa.py:1013-1045,
                                                      numerics/linalg/matrix_operations.py:1880-1906
  * right-hand sides for cells and boundary values    numerics/fv/mpfa.py:1080-1105,1414-1578
  * pressure-trace reconstruction                     numerics/fv/mpfa.py:1108-1125,1628-1690
  * vector source (gravity) terms                     numerics/fv/mpfa.py:1158-1307
  * A = div @ flux, b = -div @ bound_flux @ bc        numerics/fv/fv_elliptic.py:67-112

The reference assembles one big block-diagonal system over all grid nodes and inverts
its blocks; per node the blocks decouple, which is what is looped over here.  The
unknowns at node v are one gradient g (nd numbers) per cell touching v.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sps

MATRIX_KEYS = (
    "flux",
    "bound_flux",
    "bound_pressure_cell",
    "bound_pressure_face",
    "vector_source",
    "bound_pressure_vector_source",
)


def default_eta(name: str) -> float:
    """1/3 for (structured) triangle / tetrahedral grids, else 0 (_fvutils.py:296-305)."""
    return 1.0 / 3.0 if ("TriangleGrid" in name or "TetrahedralGrid" in name) else 0.0


def sub_half_faces(grid: dict):
    """All (cell, face, node, sign) tuples, sorted by (node, cell, face).

    ``grid`` holds raw arrays: cf_indptr/cf_indices/cf_sign (cell_faces in CSC, one
    column per cell) and fn_indptr/fn_indices (face_nodes in CSC, one column per face).
    """
    cf_ptr, cf_idx, cf_sgn = grid["cf_indptr"], grid["cf_indices"], grid["cf_sign"]
    fn_ptr, fn_idx = grid["fn_indptr"], grid["fn_indices"]
    nc = cf_ptr.size - 1
    cell_of_cf = np.repeat(np.arange(nc), np.diff(cf_ptr))
    nn_face = np.diff(fn_ptr)
    reps = nn_face[cf_idx]
    h_c = np.repeat(cell_of_cf, reps)
    h_f = np.repeat(cf_idx, reps)
    h_s = np.repeat(cf_sgn, reps).astype(np.float64)
    # node of each sub-half-face: walk the node list of the face
    start = np.repeat(fn_ptr[cf_idx], reps)
    within = np.arange(reps.sum()) - np.repeat(np.cumsum(reps) - reps, reps)
    h_v = fn_idx[start + within]
    order = np.lexsort((h_f, h_c, h_v))
    return h_c[order], h_f[order], h_v[order], h_s[order]


def subface_ids(grid: dict, faces, node) -> np.ndarray:
    """Global sub-face id (position of the (face, node) pair in the face_nodes CSC arrays, sorted indices) of the
    sub-faces `faces` x `node` -- the numbering of SubcellTopology.subfno_unique (_fvutils.py:78-90, 160-172)."""
    fn_ptr, fn_idx = grid["fn_indptr"], grid["fn_indices"]
    out = np.empty(len(faces), dtype=np.int64)
    for i, f in enumerate(faces):
        seg = fn_idx[fn_ptr[f]:fn_ptr[f + 1]]
        out[i] = fn_ptr[f] + int(np.flatnonzero(seg == node)[0])
    return out


def discretize(
    grid: dict,
    perm: np.ndarray,
    bc: dict,
    eta: float | None = None,
    vector_dim: int | None = None,
    real=None,
) -> dict:
    """MPFA-O discretization; returns the six matrices of the reference as csr.

    perm: (3, 3, Nc) permeability (reference layout, params/tensor.py:68-157).
    bc: dict with boolean arrays is_dir / is_neu / is_rob / is_internal (Nf,) and
        robin_weight (Nf,) — the fields of the reference's BoundaryCondition
        (params/bc.py:68-190).  Boundary values are given per face.
    real: None = FP64 (numpy), as the reference; a scalar constructor (``mpmath.mpf``) = every arithmetic step of the
        node-local computation in that type from the FP64 inputs on (object arrays, local inverse by mpmath): the exact
        answer to the problem the inputs pose -- the arbiter of tools/fuzz_vs_reference.py (round 6; small grids).
    """
    nd = int(grid["dim"])
    if nd not in (2, 3):
        raise ValueError("oracle covers nd = 2, 3 (1-D delegates to TPFA in the reference)")
    vd = nd if vector_dim is None else int(vector_dim)
    if eta is None:
        eta = default_eta(grid.get("name", ""))
    nodes, fc, cc = grid["nodes"], grid["face_centers"], grid["cell_centers"]
    fnrm, farea = grid["face_normals"], grid["face_areas"]
    nf, nc = fc.shape[1], cc.shape[1]
    nn_face = np.diff(grid["fn_indptr"])

    h_c, h_f, h_v, h_s = sub_half_faces(grid)
    sides_per_face = np.bincount(h_f, minlength=nf) // np.maximum(nn_face, 1)
    is_bnd_face = sides_per_face == 1

    # Row classes, mpfa.py:1452-1454: internal (fracture) faces are treated as Neumann.
    # Boundary conditions per face, or per sub-face (face_nodes CSC position) when the arrays have one
    # entry per sub-face (mpfa.py:761-768): then flux / bound_flux / the two trace matrices keep
    # sub-face rows (no collapse, mpfa.py:1117-1125) and the Neumann data are not divided by the
    # number of face nodes (mpfa.py:1516-1523)
    nsub = int(grid["fn_indices"].size)
    subface_bc = np.asarray(bc["is_dir"]).size == nsub and nsub != nf
    nbc = nsub if subface_bc else nf
    is_int = np.asarray(bc.get("is_internal", np.zeros(nbc, bool)), bool)
    is_dir = np.asarray(bc["is_dir"], bool) & ~is_int
    is_rob = np.asarray(bc["is_rob"], bool) & ~is_int
    is_neu = np.asarray(bc["is_neu"], bool) | is_int
    rw = np.asarray(bc.get("robin_weight", np.ones(nbc)), dtype=float)
    fn_ptr_, fn_idx_ = grid["fn_indptr"], grid["fn_indices"]
    perm = np.asarray(perm, dtype=float)
    if real is not None:
        import mpmath as _mp

        def _conv(arr):
            arr = np.asarray(arr, dtype=float)
            out_ = np.empty(arr.shape, dtype=object)
            flat, of = arr.ravel(), out_.ravel()
            for i_ in range(flat.size):
                of[i_] = real(float(flat[i_]))
            return out_

        nodes, fc, cc, fnrm, farea, perm, rw = (_conv(x) for x in (nodes, fc, cc, fnrm, farea, perm, rw))
        eta = real(float(eta)) if np.ndim(eta) == 0 else _conv(eta)

        def zeros(shape):
            z = np.empty(shape, dtype=object)
            z.fill(real(0))
            return z

        def invert(M):
            return np.array((_mp.matrix(M.tolist()) ** -1).tolist(), dtype=object).reshape(M.shape)
    else:
        zeros = np.zeros
        invert = None

    node_start = np.flatnonzero(np.r_[True, h_v[1:] != h_v[:-1], True])
    acc = {k: ([], [], []) for k in MATRIX_KEYS}

    def emit(key, r, c, v):
        acc[key][0].append(np.asarray(r).ravel())
        acc[key][1].append(np.asarray(c).ravel())
        acc[key][2].append(np.asarray(v, dtype=float).ravel())

    for a, b in zip(node_start[:-1], node_start[1:]):
        v = h_v[a]
        hc, hf, hs = h_c[a:b], h_f[a:b], h_s[a:b]
        cells, jloc = np.unique(hc, return_inverse=True)
        faces, sloc = np.unique(hf, return_inverse=True)
        deg, nsf, nh = cells.size, faces.size, hc.size
        if nh != nd * deg:
            # _fvutils.py:735-736: every (cell, node) pair must have exactly nd faces
            raise AssertionError("cell with != nd faces meeting in a node (e.g. pyramid)")
        m = nd * deg
        # geometry per sub-half-face
        n_h = fnrm[:nd, hf] / nn_face[hf]  # (nd, nh), stored orientation
        if real is None:
            nK = np.einsum("ih,ijh->jh", n_h, perm[:nd, :nd, hc])  # n^T K -> (nd, nh)
        else:
            nK = zeros((nd, nh))
            for h_ in range(nh):
                for j_ in range(nd):
                    nK[j_, h_] = sum(n_h[i_, h_] * perm[i_, j_, hc[h_]] for i_ in range(nd))
        if np.ndim(eta) == 0:
            eta_h = np.where(is_bnd_face[hf], 0.0 if real is None else real(0), eta)  # scalar: forced to 0 on the boundary (_fvutils.py:257-268)
        else:
            eta_h = np.asarray(eta, float)[subface_ids(grid, hf, v)]  # one value per sub-face, used as given
        fch = fc[:nd, hf]
        if "periodic_native" in grid:
            # merged periodic faces (porepy_amd/periodic.py; reference _fvutils.py:91-137): the side
            # that is not the face's own cell sees the face centre displaced by the period
            nat = np.asarray(grid["periodic_native"])[hf]
            far = (nat >= 0) & (nat != hc)
            fch = fch - np.asarray(grid["periodic_shift"])[:nd, hf] * far
        xcp = fch + eta_h * (nodes[:nd, [v]] - fch)
        d_h = xcp - cc[:nd, hc]
        col0 = nd * jloc  # first gradient column of the subcell of each h

        # global sub-face id of every local sub-face, and the index its boundary data live under
        sfid = np.array([fn_ptr_[f] + int(np.flatnonzero(fn_idx_[fn_ptr_[f]: fn_ptr_[f + 1]] == v)[0]) for f in faces])
        bid = sfid if subface_bc else faces
        neu_scale = 1.0 if subface_bc else None
        rows_F, rows_R, rows_P = [], [], []  # local subface index per row, per class
        for s, f in enumerate(faces):
            if not is_bnd_face[f]:
                rows_F.append(s)
                rows_P.append(s)
            elif is_dir[bid[s]]:
                rows_P.append(s)
            elif is_rob[bid[s]]:
                rows_R.append(s)
            elif is_neu[bid[s]]:
                rows_F.append(s)
            else:
                raise ValueError("boundary face without a boundary condition type")
        nF, nR, nP = len(rows_F), len(rows_R), len(rows_P)
        if nF + nR + nP != m:
            raise ValueError("local system is not square")
        G = zeros((m, m))
        Rc = zeros((m, deg))  # cell-pressure right-hand side
        bfaces = [s for s in range(nsf) if is_bnd_face[faces[s]]]
        bcol = {s: i for i, s in enumerate(bfaces)}
        Rb = zeros((m, len(bfaces)))
        E = zeros((m, deg * nd))  # vector-source right-hand side, column nd*j+k
        row_of = {}
        for r, s in enumerate(rows_F):
            row_of[("F", s)] = r
        for r, s in enumerate(rows_R):
            row_of[("R", s)] = nF + r
        for r, s in enumerate(rows_P):
            row_of[("P", s)] = nF + nR + r

        for h in range(nh):
            s, j, f, sg = sloc[h], jloc[h], hf[h], hs[h]
            cols = slice(col0[h], col0[h] + nd)
            if ("F", s) in row_of:
                r = row_of[("F", s)]
                G[r, cols] += sg * nK[:, h]
                E[r, nd * j : nd * j + nd] += sg * nK[:, h]
                if is_bnd_face[f]:
                    Rb[r, bcol[s]] = -(neu_scale or 1.0 / nn_face[f])
            if ("R", s) in row_of:
                r = row_of[("R", s)]
                a_s = farea[f] / nn_face[f]
                G[r, cols] += sg * nK[:, h] - rw[bid[s]] * a_s * d_h[:, h]
                Rc[r, j] += rw[bid[s]] * a_s
                E[r, nd * j : nd * j + nd] += sg * nK[:, h]
                Rb[r, bcol[s]] = -(neu_scale or 1.0 / nn_face[f])
            if ("P", s) in row_of:
                r = row_of[("P", s)]
                G[r, cols] += sg * d_h[:, h]
                Rc[r, j] += -sg
                if is_bnd_face[f]:
                    Rb[r, bcol[s]] = sg

        scale = 1.0 / np.abs(G).sum(axis=1)
        try:
            Gs = scale[:, None] * G
            if invert is not None:
                igrad = invert(Gs) * scale[None, :]
            else:
                if np.linalg.cond(Gs) > 1e14:  # singular up to rounding: the reference's LAPACK inverse raises on these
                    raise np.linalg.LinAlgError("Singular matrix")
                igrad = np.linalg.inv(Gs) * scale[None, :]
        except (np.linalg.LinAlgError, ZeroDivisionError) as exc:  # matrix_operations.py:1487-1490
            raise ValueError("Error in inversion of local linear systems") from exc

        # first-sorted side of every subface = the side fluxes are evaluated from
        first_h = np.full(nsf, -1)
        for h in range(nh - 1, -1, -1):
            first_h[sloc[h]] = h
        W = zeros((nsf, m))  # Darcy rows: -nK on the subcell of h*
        for s in range(nsf):
            h = first_h[s]
            W[s, col0[h] : col0[h] + nd] = -nK[:, h]
        # trace rows: average over the sides of the subface of p_c + d_h . g
        nsides = np.bincount(sloc, minlength=nsf)
        D = zeros((nsf, m))
        Dc = zeros((nsf, deg))
        for h in range(nh):
            s = sloc[h]
            D[s, col0[h] : col0[h] + nd] += d_h[:, h] / nsides[s]
            Dc[s, jloc[h]] += 1.0 / nsides[s]

        Wi, Di = W @ igrad, D @ igrad
        q_cell, q_bnd, q_vs = Wi @ Rc, Wi @ Rb, Wi @ E
        for s in range(nsf):  # direct part of the vector-source flux (mpfa.py:1292-1305)
            h = first_h[s]
            q_vs[s, nd * jloc[h] : nd * jloc[h] + nd] += nK[:, h]
        t_cell, t_bnd, t_vs = Di @ Rc + Dc, Di @ Rb, Di @ E

        wf = 1.0 / nn_face[faces]  # subface -> face averaging of traces
        wrow = (np.ones(nsf) if real is None else np.array([real(1)] * nsf, dtype=object)) if subface_bc else wf
        rid = sfid if subface_bc else faces
        frow = np.repeat(rid, deg)
        ccol = np.tile(cells, nsf)
        emit("flux", frow, ccol, q_cell)
        emit("bound_pressure_cell", frow, ccol, t_cell * wrow[:, None])
        if bfaces:
            bf_ids = rid[bfaces]
            frow_b = np.repeat(rid, len(bfaces))
            bcol_g = np.tile(bf_ids, nsf)
            emit("bound_flux", frow_b, bcol_g, q_bnd)
            emit("bound_pressure_face", frow_b, bcol_g, t_bnd * wrow[:, None])
        vcol = (cells[:, None] * vd + np.arange(nd)[None, :]).ravel()
        frow_v = np.repeat(faces, deg * nd)
        emit("vector_source", frow_v, np.tile(vcol, nsf), q_vs)
        emit("bound_pressure_vector_source", frow_v, np.tile(vcol, nsf), t_vs * wf[:, None])

    nr = nsub if subface_bc else nf
    shapes = {
        "flux": (nr, nc),
        "bound_flux": (nr, nr),
        "bound_pressure_cell": (nr, nc),
        "bound_pressure_face": (nr, nr),
        "vector_source": (nf, nc * vd),
        "bound_pressure_vector_source": (nf, nc * vd),
    }
    out = {}
    for k in MATRIX_KEYS:
        r, c, val = acc[k]
        if r:
            mat = sps.coo_matrix(
                (np.concatenate(val), (np.concatenate(r), np.concatenate(c))), shape=shapes[k]
            ).tocsr()
        else:
            mat = sps.csr_matrix(shapes[k])
        mat.sort_indices()
        out[k] = mat
    return out


def divergence(grid: dict) -> sps.csr_matrix:
    """div = cell_faces^T (grids/grid.py:1237-1266)."""
    nf = grid["face_centers"].shape[1]
    nc = grid["cell_centers"].shape[1]
    cf = sps.csc_matrix(
        (grid["cf_sign"].astype(float), grid["cf_indices"], grid["cf_indptr"]), shape=(nf, nc)
    )
    return cf.T.tocsr()


def assemble_matrix_rhs(grid: dict, mats: dict, bc_values: np.ndarray, vector_source=None):
    """A = div @ flux; b = -div @ bound_flux @ bc_values [- div @ vector_source @ g]
    (fv_elliptic.py:67-112)."""
    div = divergence(grid)
    A = (div @ mats["flux"]).tocsr()
    b = -div @ (mats["bound_flux"] @ bc_values)
    if vector_source is not None:
        b = b - div @ (mats["vector_source"] @ vector_source)
    return A, b
