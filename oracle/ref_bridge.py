"""Helpers that touch the *reference* PorePy objects (only usable in the build
container through oracle/shim).  TEST INFRASTRUCTURE — golden generation / oracle pinning.
"""
from __future__ import annotations

import numpy as np


def grid_to_raw(g) -> dict:
    """Flatten a reference pp.Grid into the raw-array dict used by oracle and product."""
    cf = g.cell_faces.tocsc().copy()  # the reference grid is left as it is
    cf.sort_indices()
    fn = g.face_nodes.tocsc().copy()
    fn.sort_indices()
    frac = np.zeros(g.num_faces, dtype=bool)
    for tag in ("fracture_faces",):
        if tag in g.tags:
            frac |= np.asarray(g.tags[tag], dtype=bool)
    return {
        "dim": int(g.dim),
        "name": str(g.name),
        "nodes": np.ascontiguousarray(g.nodes, dtype=np.float64),
        "cf_indptr": cf.indptr.astype(np.int32),
        "cf_indices": cf.indices.astype(np.int32),
        "cf_sign": cf.data.astype(np.int8),
        "fn_indptr": fn.indptr.astype(np.int32),
        "fn_indices": fn.indices.astype(np.int32),
        "face_normals": np.ascontiguousarray(g.face_normals, dtype=np.float64),
        "face_centers": np.ascontiguousarray(g.face_centers, dtype=np.float64),
        "cell_centers": np.ascontiguousarray(g.cell_centers, dtype=np.float64),
        "face_areas": np.ascontiguousarray(g.face_areas, dtype=np.float64),
        "cell_volumes": np.ascontiguousarray(g.cell_volumes, dtype=np.float64),
        "fracture_faces": frac,
    }


def bc_to_raw(bc) -> dict:
    return {
        "is_dir": np.asarray(bc.is_dir, bool).copy(),
        "is_neu": np.asarray(bc.is_neu, bool).copy(),
        "is_rob": np.asarray(bc.is_rob, bool).copy(),
        "is_internal": np.asarray(bc.is_internal, bool).copy(),
        "robin_weight": np.asarray(bc.robin_weight, float).copy(),
    }
