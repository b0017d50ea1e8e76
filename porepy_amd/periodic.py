"""Periodic faces (``Grid.set_periodic_map``, reference grids/grid.py:879-911) as a grid transformation.

The reference treats a pair (left face, right face) of ``periodic_face_map`` as ONE face
topologically: ``SubcellTopology`` renames the nodes of the right sub-faces to those of the left
ones and gives the right sub-faces the sub-face numbers of the left ones
(numerics/fv/_fvutils.py:91-137); the geometry stays per side (face centre / normal of the face a
cell really has); ``Mpfa`` finally copies the face rows of the left faces to the right faces
(numerics/fv/mpfa.py:900-917).

Here the same statement is a *merged grid* handed to the device library: the right cell lists the
left face, the right nodes are renamed, the right faces keep neither cells nor nodes, and
``pfv_set_periodic`` tells the kernels that the far side of a merged face sees the face centre
displaced by the period.  The device then discretizes, assembles and solves on the merged grid with
its ordinary kernels; ``copy_rows`` puts the rows of the left faces at the right faces afterwards.

Faithful to the reference including its quirk that a right-hand sub-cell computes its continuity
points with the coordinates of the *left* node when eta != 0 (the renamed node ids index
``sd.nodes``, _fvutils.py:257-269).  Restrictions, checked: maps sorted (as the reference,
_fvutils.py:96-112), left and right faces translates of each other (equal normals), and the cell of
a left face numbered below the cell of its right face (otherwise the reference drops the flux rows
of the pair: ``hf2f`` keeps the sub-face under the face of the lower cell, mpfa.py:892-917).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sps


class PeriodicMerge:
    """Result of :func:`merge_periodic`."""

    def __init__(self, raw, native, shift, left, right, label=None):
        self.label = label      # (Nn,) node renaming (right nodes carry the labels of the left ones)
        self.raw = raw          # merged raw grid (see grid.grid_to_raw)
        self.native = native    # (Nf,) int32: cell that sees a merged face in place, -1 elsewhere
        self.shift = shift      # (3, Nf): x_f(left) - x_f(right) at the left faces
        self.left, self.right = left, right

    def copy_rows(self, M: sps.spmatrix, trace: bool = False) -> sps.csr_matrix:
        """Rows of the right faces := rows of the left faces (mpfa.py:907-917).  ``trace``: the
        matrix is one of the pressure-trace reconstructions; the reference's sub-face average
        ``area_mat`` (mpfa.py:1115-1124) is built from ``hf2f`` *after* the rows were duplicated,
        which weights a periodic sub-face 2 / #nodes instead of 1 / #nodes - reproduced here."""
        nf = self.native.size
        if M.shape[0] != nf or self.left.size == 0:
            return M
        S = sps.csr_matrix((np.ones(self.left.size), (self.right, self.left)), shape=(nf, nf))
        out = sps.csr_matrix(M) + S @ M
        if trace:
            w = np.ones(nf)
            w[self.left] = 2.0
            w[self.right] = 2.0
            out = sps.diags(w) @ out
        out = out.tocsr()
        out.sort_indices()
        return out


def merged_subface_order(face_nodes, merge: PeriodicMerge) -> np.ndarray:
    """Conditions per sub-face on a grid with periodic faces: the reference numbers the merged sub-faces by their
    position in the caller's face_nodes arrays, the right sub-faces taking the numbers of the left ones and the gaps
    closed (SubcellTopology, numerics/fv/_fvutils.py:82-160); the device numbers them by the sorted CSC arrays of the
    merged grid (right faces emptied, nodes renamed).  order[d] = the caller's number of device sub-face d."""
    fn = sps.csc_matrix(face_nodes)
    nf = fn.shape[1]
    nnf = np.diff(fn.indptr)
    keep_face = np.ones(nf, dtype=bool)
    keep_face[merge.right] = False
    keep = np.repeat(keep_face, nnf)                      # caller positions that survive
    face = np.repeat(np.arange(nf), nnf)[keep]
    lbl = merge.label[fn.indices[keep]]
    caller = np.arange(int(keep.sum()))                   # compact numbers, in the caller's storage order
    return caller[np.lexsort((lbl, face))]


def merge_periodic(raw: dict, periodic_face_map) -> PeriodicMerge:
    pm = np.asarray(periodic_face_map)
    if pm.ndim != 2 or pm.shape[0] != 2:
        raise ValueError("dimension 0 of periodic_face_map must be of size 2")
    left, right = pm[0].astype(np.int64), pm[1].astype(np.int64)
    if not (np.array_equal(np.sort(left), left) and np.array_equal(np.sort(right), right)):
        # same restriction and text as the reference (_fvutils.py:103-112)
        raise NotImplementedError("Can not create subcell topology for periodic faces that are not sorted")
    nf = raw["face_centers"].shape[1]
    nc = raw["cell_centers"].shape[1]
    nn = raw["nodes"].shape[1]
    cf_ptr, cf_idx, cf_sgn = (np.asarray(raw[k]) for k in ("cf_indptr", "cf_indices", "cf_sign"))
    fn_ptr, fn_idx = np.asarray(raw["fn_indptr"]), np.asarray(raw["fn_indices"])
    cell_of_e = np.repeat(np.arange(nc), np.diff(cf_ptr))
    sides = np.bincount(cf_idx, minlength=nf)
    if np.any(sides[left] != 1) or np.any(sides[right] != 1):
        raise ValueError("periodic faces must be boundary faces")
    cell_of_face = np.full(nf, -1, dtype=np.int64)
    cell_of_face[cf_idx] = cell_of_e  # single-sided faces: their one cell
    cl, cr = cell_of_face[left], cell_of_face[right]
    if np.any(cl >= cr):
        raise NotImplementedError(
            "periodic_face_map[0] must hold the faces of the lower-numbered cells: the reference keeps a "
            "merged sub-face under the face of its lower cell and copies the rows of periodic_face_map[0]")
    nrm = np.asarray(raw["face_normals"], dtype=float)
    scale = np.linalg.norm(nrm[:, left], axis=0)
    if np.any(np.linalg.norm(nrm[:, left] - nrm[:, right], axis=0) > 1e-9 * scale):
        raise NotImplementedError("periodic faces must be translates of each other (equal normals)")
    nnf = np.diff(fn_ptr)
    if np.any(nnf[left] != nnf[right]):
        raise ValueError("periodic faces with different numbers of nodes")

    # --- node renaming, in the order of the reference's loop over the right sub-faces (sorted by
    # face, then by position in the face's node list): every node carrying the label of the right
    # node gets the label of the left node
    label = np.arange(nn)
    members = {}
    for fl, fr in zip(left, right):
        for j in range(nnf[fl]):
            rn = label[fn_idx[fn_ptr[fr] + j]]
            ln = label[fn_idx[fn_ptr[fl] + j]]
            if rn == ln:
                continue
            mr = members.pop(rn, [rn])
            ml = members.setdefault(ln, [ln])
            ml.extend(mr)
            label[mr] = ln

    # --- merged face_nodes: nodes renamed, right faces emptied
    keep = np.ones(nf, dtype=bool)
    keep[right] = False
    counts = np.where(keep, nnf, 0)
    new_fn_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    sel = np.repeat(keep, nnf)
    new_fn_idx = label[fn_idx[sel]].astype(np.int32)
    fn = sps.csc_matrix((np.ones(new_fn_idx.size, dtype=np.int8), new_fn_idx, new_fn_ptr), shape=(nn, nf))
    if fn.has_canonical_format is False:
        fn.sort_indices()
    fn.sort_indices()
    # --- merged cell_faces: the right cell lists the left face
    to_left = np.arange(nf)
    to_left[right] = left
    cf = sps.csc_matrix((np.asarray(cf_sgn).astype(np.int8), to_left[cf_idx], cf_ptr), shape=(nf, nc))
    cf.sort_indices()
    out = dict(raw)
    out["fn_indptr"], out["fn_indices"] = fn.indptr.astype(np.int32), fn.indices.astype(np.int32)
    out["cf_indptr"], out["cf_indices"] = cf.indptr.astype(np.int32), cf.indices.astype(np.int32)
    out["cf_sign"] = np.asarray(cf.data).astype(np.int8)
    native = np.full(nf, -1, dtype=np.int32)
    native[left] = cl
    shift = np.zeros((3, nf))
    fc = np.asarray(raw["face_centers"], dtype=float)
    shift[:, left] = fc[:, left] - fc[:, right]
    out["periodic_native"] = native
    out["periodic_shift"] = shift
    return PeriodicMerge(out, native, shift, left, right, label=label)
