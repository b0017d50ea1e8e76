"""Index bookkeeping for partial (re)discretization — which faces get new rows.

Host-side set logic only (the arithmetic stays on the device): the same active sets as the
reference's ``cell_ind_for_partial_update`` (numerics/fv/_fvutils.py:1260-1462), written
with boolean incidence products.

* ``cells``: faces sharing a node with a listed cell are rediscretized; the cell set is
  everything sharing a node with those faces.
* ``faces``: faces sharing a node with a listed face; cells two node-rings out.
* ``nodes``: cells around the listed nodes; only faces whose nodes are *all* listed.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sps


def _incidence(sd):
    fn = sps.csc_matrix(sd.face_nodes).astype(bool).astype(np.int32)  # nodes x faces
    cf = sps.csc_matrix(sd.cell_faces)
    cf = sps.csc_matrix((np.ones(cf.data.size, dtype=np.int32), cf.indices, cf.indptr), shape=cf.shape)
    cn = (fn @ cf).tocsc()  # nodes x cells (counts > 0)
    return fn, cn


def _flag(n, idx):
    v = np.zeros(n, dtype=np.int32)
    v[np.asarray(idx, dtype=np.int64)] = 1
    return v


def active_indices(sd, cells=None, faces=None, nodes=None):
    """(active_cells, active_faces), both sorted int arrays."""
    fn, cn = _incidence(sd)
    nn, nf = fn.shape
    nc = cn.shape[1]
    face_on = np.zeros(nf, dtype=bool)
    cell_on = np.zeros(nc, dtype=bool)
    if cells is not None:
        v = cn @ _flag(nc, cells) > 0
        f = fn.T @ v.astype(np.int32) > 0
        face_on |= f
        v |= fn @ f.astype(np.int32) > 0
        cell_on |= cn.T @ v.astype(np.int32) > 0
    if faces is not None:
        pv = fn @ _flag(nf, faces) > 0
        f = fn.T @ pv.astype(np.int32) > 0
        face_on |= f
        an = fn @ f.astype(np.int32) > 0
        c1 = cn.T @ an.astype(np.int32) > 0
        an |= cn @ c1.astype(np.int32) > 0
        cell_on |= cn.T @ an.astype(np.int32) > 0
    if nodes is not None:
        v = _flag(nn, nodes)
        cell_on |= cn.T @ v > 0
        per_face = np.asarray(fn.sum(axis=0)).ravel()
        face_on |= (fn.T @ v) == per_face
    return np.flatnonzero(cell_on), np.flatnonzero(face_on)
