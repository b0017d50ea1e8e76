"""``Mpfa`` — the reference's MPFA-O discretization operator, executed on an MI355X.

Same operator API as ``pp.Mpfa`` / ``FVElliptic`` / ``Discretization`` of the reference
(numerics/fv/mpfa.py:65-508, numerics/fv/fv_elliptic.py:30-112,
numerics/discretization.py:12-121): ``Mpfa(keyword)``, ``ndof(sd)``,
``discretize(sd, data)``, ``update_discretization(sd, data)``,
``assemble_matrix_rhs(sd, data)``, the six ``*_matrix_key`` attributes, parameters read
from ``data[PARAMETERS][keyword]`` and results written as scipy csr matrices into
``data[DISCRETIZATION_MATRICES][keyword]``.  All arithmetic happens in the HIP kernels
behind the C ABI (include/porefv.h); this file is argument marshalling only.

When the reference package is importable, :func:`as_porepy_discretization` returns a
subclass of ``pp.Mpfa`` with ``discretize`` / ``assemble_matrix_rhs`` routed here, so
``pp.Mpfa = as_porepy_discretization()`` swaps the operator under existing models
(INTEGRATION.md).
"""
from __future__ import annotations

import os

import numpy as np

from . import _lib
from .lazy import LazyCsr
from .grid import grid_to_raw
from .partial import active_indices
from .periodic import merge_periodic, merged_subface_order
from .params import DISCRETIZATION_MATRICES, PARAMETERS, bc_flags

_KEYS = (
    ("flux", _lib.MAT_FLUX),
    ("bound_flux", _lib.MAT_BOUND_FLUX),
    ("bound_pressure_cell", _lib.MAT_BOUND_PRESSURE_CELL),
    ("bound_pressure_face", _lib.MAT_BOUND_PRESSURE_FACE),
    ("vector_source", _lib.MAT_VECTOR_SOURCE),
    ("bound_pressure_vector_source", _lib.MAT_BOUND_PRESSURE_VECTOR_SOURCE),
)


def determine_eta(sd) -> float:
    """Default continuity point (numerics/fv/_fvutils.py:280-305): 1/3 on simplex grids
    (recognised by the grid name), 0 otherwise."""
    name = str(getattr(sd, "name", ""))
    return 1.0 / 3.0 if ("TriangleGrid" in name or "TetrahedralGrid" in name) else 0.0


def sps_nnz(m) -> int:
    return int(m.nnz)


def subface_order(face_nodes) -> np.ndarray:
    """order[p_sorted] = position of the same (face, node) pair in the caller's face_nodes arrays
    (identity when the indices are already sorted inside every column)."""
    import scipy.sparse as sps

    fn = sps.csc_matrix(face_nodes)
    col = np.repeat(np.arange(fn.shape[1]), np.diff(fn.indptr))
    return np.lexsort((fn.indices, col))


def renumber_subfaces(M, order: np.ndarray, cols: bool):
    """Rows (and columns) from the device's sub-face numbering back to the caller's."""
    import scipy.sparse as sps

    coo = sps.coo_matrix(M)
    r = order[coo.row]
    c = order[coo.col] if cols else coo.col
    out = sps.csr_matrix((coo.data, (r, c)), shape=M.shape)
    out.sort_indices()
    return out


class _FaceBC:
    """All-Neumann per-face placeholder used while the real conditions are given per sub-face."""

    def __init__(self, nf: int):
        self.is_dir = np.zeros(nf, bool)
        self.is_neu = np.ones(nf, bool)
        self.is_rob = np.zeros(nf, bool)
        self.is_internal = np.zeros(nf, bool)
        self.robin_weight = np.ones(nf)


def plane_basis(nodes: np.ndarray, tol: float = 1e-5):
    """(2, 3) orthonormal basis of the plane holding the nodes of a 2-D grid, or None when the
    grid already lies in a plane z = const (then the first two coordinates are used as is).
    The role of ``map_geometry.project_plane_matrix`` (geometry/map_geometry.py:215-270); the
    discretization does not depend on which in-plane basis is picked."""
    x = np.asarray(nodes, dtype=float)
    if np.ptp(x[2]) <= 1e-12 * max(1.0, np.ptp(x[0]), np.ptp(x[1])):
        return None
    xc = x - x.mean(axis=1, keepdims=True)
    w, v = np.linalg.eigh(xc @ xc.T)  # ascending: v[:, 0] is the plane normal
    if w[0] > (tol ** 2) * w[2] * x.shape[1]:
        raise AssertionError("the nodes of a 2-D grid must be planar")
    return np.ascontiguousarray(v[:, [2, 1]].T)


def reference_plane_rotation(nodes: np.ndarray, face_centers: np.ndarray, tol: float = 1e-5):
    """``(R, dim)`` of the reference's ``map_geometry.map_grid`` for a 2-D grid (geometry/map_geometry.py:118-135):
    the 3 x 3 rotation that turns the plane of the nodes onto a coordinate plane and the mask of the two coordinates
    that vary afterwards.  Restated because the reference's un-rotation of the vector-source matrices with
    ``ambient_dimension == 2`` (numerics/fv/mpfa.py:425-462) depends on WHICH rotation was picked: the normal is the
    longest cross product of the longest centred node vector with the others (``compute_normal``, :440-517), R the
    Rodrigues rotation about ``normal x e_z`` by the angle between the two (``project_plane_matrix`` :215-270,
    ``rotation_matrix`` :327-358)."""
    x = np.asarray(nodes, dtype=float)
    v = x - x.mean(axis=1).reshape((-1, 1))
    nrm = np.linalg.norm(v, axis=0)
    v1 = v[:, int(np.argmax(nrm))]
    cross = np.array([v1[1] * v[2] - v1[2] * v[1], v1[2] * v[0] - v1[0] * v[2], v1[0] * v[1] - v1[1] * v[0]])
    normal = cross[:, int(np.argmax(np.linalg.norm(cross, axis=0)))]
    normal = normal / np.linalg.norm(normal)
    ref = np.array([0.0, 0.0, 1.0])
    angle = np.arccos(np.dot(normal, ref))
    axis = np.cross(normal, ref)
    if np.allclose(axis, np.zeros(3)):
        R = np.identity(3)
    else:
        axis = axis / np.linalg.norm(axis)
        W = np.array([[0.0, -axis[2], axis[1]], [axis[2], 0.0, -axis[0]], [-axis[1], axis[0], 0.0]])
        R = np.identity(3) + np.sin(angle) * W + (1.0 - np.cos(angle)) * (W @ W)
    fc = R @ np.asarray(face_centers, dtype=float)
    check = np.sum(np.abs(fc.T - fc[:, 0]), axis=0)
    check = check / np.sum(check)
    dim = np.logical_not(np.isclose(check, 0, atol=tol, rtol=0))
    return R, dim


def planar_source_map(sd, T: np.ndarray):
    """Tilted 2-D grid with ``ambient_dimension == 2``: the (2 Nc x 2 Nc) matrix the reference leaves on the right of
    the vector-source matrices (mpfa.py:425-462), expressed for matrices computed in the in-plane basis ``T`` of this
    package.  The reference multiplies its local matrices (basis ``R[dim]``) by rows AND columns ``dim + 2 c`` of
    blockdiag(R, ..., R) -- the leading 2 Nc x 2 Nc corner of a matrix of 3 x 3 blocks; reproduced as it is."""
    import scipy.sparse as sps

    R, dim = reference_plane_rotation(sd.nodes, sd.face_centers)
    nc = sd.num_cells
    to_ref_basis = sps.kron(sps.identity(nc, format="csr"), sps.csr_matrix(T @ R[dim].T), format="csr")
    full = sps.kron(sps.identity(nc, format="csr"), sps.csr_matrix(R), format="csr")
    idx = (np.where(dim)[0].reshape((-1, 1)) + 2 * np.arange(nc)).ravel("F")
    corner = full[idx][:, idx]
    return (to_ref_basis @ corner).tocsr()


def grid_fingerprint(sd) -> tuple:
    """Cheap digest of what `_upload_grid` reads from ``sd``: sizes, a strided checksum of the node and
    face-centre coordinates, and the periodic map.  Not cryptographic -- it catches the cases the
    reference handles by re-reading ``sd`` in every call (moved nodes + compute_geometry, a periodic map
    set or cleared after the first discretize)."""
    nodes = np.asarray(sd.nodes)
    fc = np.asarray(sd.face_centers)
    step_n = max(1, nodes.shape[1] // 4096)
    step_f = max(1, fc.shape[1] // 4096)
    per = getattr(sd, "periodic_face_map", None)
    per_key = None if per is None else (np.asarray(per).shape, int(np.asarray(per).sum()))
    return (sd.num_cells, sd.num_faces, sd.num_nodes, float(nodes[:, ::step_n].sum()), float(nodes.sum()),
            float(fc[:, ::step_f].sum()), float(np.asarray(sd.face_areas).sum()), per_key)


_IGNORED_NOTED: set = set()


def note_ignored_parameters(pd: dict, keyword: str, notes: dict | None = None) -> None:
    """Keys the reference's Mpfa reads (mpfa.py:119-167; ``notes``: another class's) that have no effect here: say
    so once per key."""
    import logging

    notes = notes if notes is not None else {
        "mpfa_inverter": "the local systems are inverted by the device kernel (register Gauss-Jordan); "
                         "the reference's numba / python / cython choice does not apply",
        "reconstruction_eta": "pressure traces are reconstructed at the continuity points of `mpfa_eta`, "
                              "as the reference does when the key is absent",
    }
    for key, why in notes.items():
        if key in pd and (keyword, key) not in _IGNORED_NOTED:
            _IGNORED_NOTED.add((keyword, key))
            logging.getLogger("porepy_amd").warning("parameter %r of %r is ignored: %s", key, keyword, why)


def estimate_device_bytes(sd) -> int:
    """Upper estimate of the HBM one MPFA handle needs for ``sd`` (the reference estimates its peak memory
    to choose a sub-problem count, mpfa.py:1315-1355): grid, sub-cell topology, per-node response tables,
    the four CSR patterns with their value arrays, and the transient work space of the symbolic phase."""
    nd = sd.dim
    nc, nf = sd.num_cells, sd.num_faces
    nsf = sps_nnz(sd.face_nodes)                     # sub-faces
    fpc = sps_nnz(sd.cell_faces) / max(nc, 1)        # faces per cell
    npf = nsf / max(nf, 1)                           # nodes per face
    nh = sps_nnz(sd.cell_faces) * npf                # sub-half-faces
    nn = sd.num_nodes
    deg = nh / nd / max(nn, 1)                       # cells per node
    n_loc = nsf / max(nn, 1)                         # sub-faces per node
    row = npf * deg * 0.8                            # flux row length (structural stencil, shared cells merged)
    nnz_flux = nf * row
    nnz_A = nc * min(fpc * row * 0.35, nc)
    tables = 8 * nn * (n_loc + 1) * (nd * deg + 1)
    patterns = 4 * (nnz_flux * (1 + nd) + nnz_A) + 3 * nnz_flux
    values = 8 * (2 * nnz_flux * (1 + nd) + nnz_A) + 8 * 2 * 0.05 * nnz_flux
    topo = 12 * nh + 64 * nsf + 32 * nf + 40 * nn
    grid = 8 * (3 * nn + 3 * nc + 7 * nf) + 9 * 8 * nc + 5 * (sps_nnz(sd.cell_faces) + nsf)
    staging = nf * npf * deg * (4 + npf) + 4 * nc * fpc * npf * deg
    solver = 12 * nnz_A * 2.3 + 8 * nc * 24
    return int(1.15 * (tables + patterns + values + topo + grid + staging + solver))


def plan_subproblems(sd, partition_arguments, free_bytes, need=None, what: str = "MPFA") -> int:
    """Number of overlapping sub-grids the discretization is done in (reference: mpfa.py:157-161 with
    _fvutils.parse_partition_arguments, the peak-memory estimate of mpfa.py:1315-1355 and
    _fvutils.subproblems, _fvutils.py:414-539).  ``num_subproblems`` is taken as given; ``max_memory`` bounds the
    estimated device footprint of one piece in BYTES (the reference compares it with an element-count estimate of
    its host peak, mpfa.py:1329-1355 -- the key keeps its role, the unit is the device's); without either the grid is split only when it does
    not fit the free HBM of the device."""
    import logging
    import math

    pa = dict(partition_arguments or {})
    if need is None:
        need = estimate_device_bytes(sd)
    if "num_subproblems" in pa:
        n = max(1, int(pa["num_subproblems"]))
    elif "max_memory" in pa:
        n = max(1, math.ceil(need / float(pa["max_memory"])))
    elif free_bytes is not None and need > 0.9 * free_bytes:
        n = math.ceil(need / (0.6 * free_bytes))  # pieces carry their overlap and the merge buffers
    else:
        n = 1
    n = min(n, max(1, sd.num_cells))
    if n > 1:
        logging.getLogger("porepy_amd").info(
            "%s on %d cells in %d overlapping pieces (estimated %.1f GB in one piece, %s GB free)", what, sd.num_cells,
            n, need / 1e9, "?" if free_bytes is None else f"{free_bytes / 1e9:.1f}")
    return n


def partition_cells(sd, nparts: int) -> np.ndarray:
    """Owner piece of every cell: equal chunks of the cells along a Morton curve of their centres (compact
    pieces, small overlaps; the reference partitions with metis / structured blocks / coordinates,
    grids/partition.py:269-297 -- any partition yields the same matrices)."""
    from .distributed import morton_order

    order = morton_order(np.asarray(sd.cell_centers), sd.dim)
    owner = np.empty(sd.num_cells, dtype=np.int32)
    owner[order] = (np.arange(sd.num_cells, dtype=np.int64) * nparts) // max(sd.num_cells, 1)
    return owner


class Mpfa:
    """MPFA-O flux discretization for ``keyword`` on the device."""

    def __init__(self, keyword: str, device: int = 0, library=None, lazy: bool = False):
        self.keyword = keyword
        self.device = device
        self._library = library  # None -> the gfx950 product library
        # lazy: data[DISCRETIZATION_MATRICES][kw] holds LazyCsr proxies (lazy.py) instead of host copies
        self.lazy = bool(lazy)
        self.flux_matrix_key = "flux"
        self.bound_flux_matrix_key = "bound_flux"
        self.bound_pressure_cell_matrix_key = "bound_pressure_cell"
        self.bound_pressure_face_matrix_key = "bound_pressure_face"
        self.vector_source_matrix_key = "vector_source"
        self.bound_pressure_vector_source_matrix_key = "bound_pressure_vector_source"
        self._contexts: dict = {}
        self._fingerprints: dict = {}  # id(sd) -> cheap digest of the uploaded geometry / topology
        self._tpfa_discr = None  # grids of dimension < 2
        self._plane: dict = {}  # id(sd) -> (2, 3) in-plane basis of a tilted 2-D grid, or None
        self._periodic: dict = {}  # id(sd) -> PeriodicMerge of a grid with periodic faces, or None
        self._split: dict = {}  # id(sd) -> (sd, A) of a grid discretized in pieces (no whole-grid handle exists)
        self._split_ctx = None  # handle the systems of a discretization in pieces are solved on (kept between solves)

    # ---- Discretization API ---------------------------------------------------------
    def ndof(self, sd) -> int:
        return sd.num_cells

    def context(self, sd) -> _lib.Context:
        """Device handle holding ``sd`` (created and uploaded on first use)."""
        key = id(sd)
        ent = self._contexts.get(key)
        if ent is None or ent[0] is not sd:
            ctx = _lib.Context(self.device, self._library)
            self._upload_grid(ctx, sd)
            self._contexts[key] = (sd, ctx)
            self._fingerprints[key] = grid_fingerprint(sd)
            return ctx
        # the reference reads sd afresh in every discretize call: if the caller moved nodes, recomputed
        # the geometry or (un)set a periodic map since the upload, upload again
        fp = grid_fingerprint(sd)
        if fp != self._fingerprints.get(key):
            self._upload_grid(ent[1], sd)
            self._fingerprints[key] = fp
        return ent[1]

    def invalidate(self, sd=None) -> None:
        """Forget the device copy of ``sd`` (all grids if None); the next call uploads it again."""
        keys = list(self._contexts) if sd is None else [id(sd)]
        for k in keys:
            self._contexts.pop(k, None)
            self._fingerprints.pop(k, None)
            self._plane.pop(k, None)      # (a later grid object may reuse the id: no stale basis / merge map)
            self._periodic.pop(k, None)

    def _upload_grid(self, ctx, sd):
        raw = grid_to_raw(sd)
        T = None
        if sd.dim == 2:
            # 2-D grids embedded in 3-D are discretized in local in-plane coordinates
            # (mpfa.py:733-754 via map_geometry.map_grid); any orthonormal in-plane basis
            # gives the same matrices once the vector source is mapped back
            T = plane_basis(raw["nodes"])
            if T is not None:
                for k in ("nodes", "face_normals", "face_centers", "cell_centers"):
                    loc = np.zeros_like(raw[k])
                    loc[:2] = T @ raw[k]
                    raw[k] = loc
        self._plane[id(sd)] = T
        merge = None
        if hasattr(sd, "periodic_face_map"):
            # periodic faces: discretize the merged grid, copy the rows of the left faces to the
            # right faces afterwards (_fvutils.py:91-137, mpfa.py:900-917; periodic.py)
            merge = merge_periodic(raw, sd.periodic_face_map)
            raw = merge.raw
        self._periodic[id(sd)] = merge
        ctx.set_grid(raw)
        if merge is not None:
            ctx.set_periodic(merge.native, merge.shift)

    def _free_device_bytes(self):
        return _lib.free_device_bytes(self.device, self._library)

    def _discretize_in_pieces(self, sd, data: dict, nparts: int, eta: float) -> None:
        """Memory-bounded discretization (mpfa.py:246-372, _fvutils.py:414-539): the cells are partitioned, every
        piece is extended by one node-ring of cells (all interaction regions of its own cells' faces are then
        complete), discretized on the device on its own, and the rows of the faces of its own cells are merged
        into the global matrices on the host; a face between two pieces is computed by both and averaged, as
        the reference does.  One piece is resident in HBM at a time.  The system matrix div @ flux is taken
        from the pieces too (rows of a piece's own cells are complete there).

        The same two halves serve the cell-sharded discretization of ONE large subdomain across ranks
        (``discretize_piece`` on the rank that owns a piece, ``merge_pieces`` on every rank after the exchange:
        porepy_amd/md_sharding.py)."""
        owner = partition_cells(sd, nparts)
        payloads = [self._piece_payload(sd, data, owner, r, eta) for r in range(nparts) if np.any(owner == r)]
        self.merge_pieces(sd, data, payloads)

    def _piece_payload(self, sd, data: dict, owner, r: int, eta: float) -> dict:
        """Rows of the faces (and of ``div @ flux``: of the cells) piece ``r`` of the cell partition ``owner`` owns, as
        COO triplets in GLOBAL numbering -- plain numpy arrays: what a rank sends to the others."""
        from .distributed import extract_subdomain

        pd = data[PARAMETERS][self.keyword]
        raw = grid_to_raw(sd)
        nd = sd.dim
        kval = np.asarray(pd["second_order_tensor"].values, dtype=float)
        bnd = pd["bc"]
        flags = bc_flags(bnd)
        robin = np.asarray(bnd.robin_weight, dtype=float)
        out = {"piece": int(r), "mats": {}, "faces": np.zeros(0, np.int64), "system": None}
        if not np.any(owner == r):
            return out
        lp = extract_subdomain(raw, owner, r)
        ctx = _lib.Context(self.device, self._library)
        try:
            ctx.set_grid(lp.raw)
            lfl = flags[lp.face_gid].copy()
            lfl[lp.artificial_boundary] = _lib.BC_NEU  # never touches a node of an own cell
            ctx.set_params(np.ascontiguousarray(kval[:, :, lp.cell_gid]), lfl,
                           np.ascontiguousarray(robin[lp.face_gid]), eta, None)
            try:
                ctx.discretize(rebuild_topology=True)
            except _lib.PorefvError as e:
                if e.status == 1:
                    raise ValueError("Error in inversion of local linear systems") from e
                if e.status == 2:
                    raise AssertionError(e.message) from e
                raise
            cfp = lp.raw["cf_indptr"]
            own_faces = np.unique(lp.raw["cf_indices"][: cfp[lp.n_own]])  # faces of the piece's own cells
            out["faces"] = np.asarray(lp.face_gid[own_faces], dtype=np.int64)
            vcol = (nd * lp.cell_gid[:, None] + np.arange(nd)[None, :]).ravel()
            for name, which in _KEYS:
                M = ctx.matrix_rows(which, own_faces).tocoo()
                cmap = (vcol if "vector_source" in name else
                        lp.face_gid if name in ("bound_flux", "bound_pressure_face") else lp.cell_gid)
                out["mats"][name] = (np.asarray(lp.face_gid[own_faces][M.row], dtype=np.int64),
                                     np.asarray(cmap[M.col], dtype=np.int64), np.asarray(M.data, dtype=float))
            ctx.assemble(np.zeros(lp.face_gid.size), None, None)
            S = ctx.matrix_rows(_lib.MAT_SYSTEM, np.arange(lp.n_own)).tocoo()
            out["system"] = (np.asarray(lp.cell_gid[S.row], dtype=np.int64), np.asarray(lp.cell_gid[S.col], dtype=np.int64),
                             np.asarray(S.data, dtype=float))
        finally:
            ctx.close()
        return out

    def merge_pieces(self, sd, data: dict, payloads) -> None:
        """The global matrices from the pieces' rows (host scipy, as the reference's own merge mpfa.py:298-372): rows of
        a face two pieces computed are averaged; pieces in ascending order, so every rank that merges the same payloads
        stores the same bits."""
        import scipy.sparse as sps

        pd = data[PARAMETERS][self.keyword]
        md = data.setdefault(DISCRETIZATION_MATRICES, {}).setdefault(self.keyword, {})
        nd, nc, nf = sd.dim, sd.num_cells, sd.num_faces
        ncols = {"flux": nc, "bound_flux": nf, "bound_pressure_cell": nc, "bound_pressure_face": nf,
                 "vector_source": nd * nc, "bound_pressure_vector_source": nd * nc}
        payloads = sorted((p for p in payloads if p.get("system") is not None), key=lambda p: p["piece"])
        count = np.zeros(nf, dtype=np.int64)
        for p in payloads:
            count[p["faces"]] += 1
        scale = 1.0 / np.maximum(count, 1)
        for name, _ in _KEYS:
            if payloads:
                rr = np.concatenate([p["mats"][name][0] for p in payloads])
                cc = np.concatenate([p["mats"][name][1] for p in payloads])
                vv = np.concatenate([p["mats"][name][2] for p in payloads])
            else:
                rr = cc = np.zeros(0, np.int64)
                vv = np.zeros(0)
            M = sps.coo_matrix((vv * scale[rr], (rr, cc)), shape=(nf, ncols[name])).tocsr()
            M.sum_duplicates()
            M.sort_indices()
            md[name] = M
        A = sps.coo_matrix((np.concatenate([p["system"][2] for p in payloads]),
                            (np.concatenate([p["system"][0] for p in payloads]),
                             np.concatenate([p["system"][1] for p in payloads]))), shape=(nc, nc)).tocsr()
        A.sum_duplicates()
        A.sort_indices()
        self._split[id(sd)] = (sd, A)
        self._contexts.pop(id(sd), None)
        self._fingerprints.pop(id(sd), None)
        pd["active_cells"] = np.arange(nc)
        pd["active_faces"] = np.arange(nf)

    def pieces_supported(self, sd, data: dict) -> bool:
        """Whether this (grid, parameters) pair can be discretized piece by piece: a full discretization of a grid of
        dimension >= 2 in its own coordinates, conditions per face, one continuity point for all sub-faces, no periodic
        faces -- the cases ``discretize`` itself splits under ``partition_arguments`` (the same predicate)."""
        pd = data[PARAMETERS][self.keyword]
        if sd.dim < 2:
            return False
        partial = any(pd.get(k) is not None for k in ("specified_cells", "specified_faces", "specified_nodes"))
        update = bool(pd.get("update_discretization", False))
        vdim = pd.get("ambient_dimension", sd.dim)
        subface = np.asarray(pd["bc"].is_dir).size != sd.num_faces
        eta = pd.get("mpfa_eta", None)
        eta_sub = eta is not None and np.asarray(eta).size != 1
        return not (partial or update or subface or eta_sub or hasattr(sd, "periodic_face_map") or vdim != sd.dim
                    or (sd.dim == 2 and plane_basis(grid_to_raw(sd)["nodes"]) is not None))

    def discretize_piece(self, sd, data: dict, piece: int, nparts: int) -> dict:
        """Piece ``piece`` of ``nparts`` of the cell partition of ``sd`` (``partition_cells``: along the Morton curve of the
        cell centres, the same on every rank), discretized on this object's device; returns the payload ``merge_pieces``
        takes.  Where the pair cannot be split (``pieces_supported``), piece 0 carries the whole discretization."""
        if not self.pieces_supported(sd, data):
            if piece != 0:
                return {"piece": int(piece), "mats": {}, "faces": np.zeros(0, np.int64), "system": None, "whole": None}
            self.discretize(sd, data)
            md = data[DISCRETIZATION_MATRICES][self.keyword]
            return {"piece": 0, "mats": {}, "faces": np.zeros(0, np.int64), "system": None,
                    "whole": {name: md[name].tocsr() if hasattr(md[name], "tocsr") else md[name] for name, _ in _KEYS}}
        pd = data[PARAMETERS][self.keyword]
        eta = pd.get("mpfa_eta", None)
        if eta is None:
            eta = determine_eta(sd)
        note_ignored_parameters(pd, self.keyword)
        owner = partition_cells(sd, int(nparts))
        return self._piece_payload(sd, data, owner, int(piece), float(eta))

    def merge_piece_payloads(self, sd, data: dict, payloads) -> None:
        """``merge_pieces`` for payloads that may carry an undivided discretization (``discretize_piece`` on a pair that
        cannot be split)."""
        whole = [p for p in payloads if p.get("whole")]
        if whole:
            md = data.setdefault(DISCRETIZATION_MATRICES, {}).setdefault(self.keyword, {})
            md.update(whole[0]["whole"])
            pd = data[PARAMETERS][self.keyword]
            pd["active_cells"] = np.arange(sd.num_cells)
            pd["active_faces"] = np.arange(sd.num_faces)
            return
        self.merge_pieces(sd, data, payloads)

    def _split_system(self, sd, data: dict, source=None):
        """(A, b) of a grid discretized in pieces: A from the pieces' device-side div @ flux, b from the merged
        boundary / vector-source matrices (two host SpMVs, fv_elliptic.py:98-112)."""
        pd = data[PARAMETERS][self.keyword]
        md = data[DISCRETIZATION_MATRICES][self.keyword]
        div = sd.cell_faces.T.tocsr()
        q = md["bound_flux"] @ np.asarray(pd["bc_values"], dtype=float)
        # the merged / split-off matrices are in the CALLER's coordinates (a plane of a union carries the lift to the
        # ambient space in its vector-source columns already): the vector source is taken as given, not projected
        vs = pd.get("vector_source", None)
        if vs is not None:
            q = q + md["vector_source"] @ np.asarray(vs, dtype=float)
        b = -(div @ q)
        if source is not None:
            b = b + np.asarray(source, dtype=float)
        return self._split[id(sd)][1], b

    def _tpfa(self):
        if self._tpfa_discr is None:
            from .tpfa import Tpfa

            self._tpfa_discr = Tpfa(self.keyword, self.device, self._library)
        return self._tpfa_discr

    def discretize(self, sd, data: dict) -> None:
        if sd.dim < 2:
            # 1-D grids go to TPFA, 0-D grids get empty matrices (mpfa.py:690-723, 129-149); the
            # reference's post-processing products drop the explicit zeros Tpfa stores (:360-372)
            self._tpfa().discretize(sd, data)
            md = data[DISCRETIZATION_MATRICES][self.keyword]
            for name, _ in _KEYS:
                md[name].eliminate_zeros()
            return
        pd = data[PARAMETERS][self.keyword]
        md = data.setdefault(DISCRETIZATION_MATRICES, {}).setdefault(self.keyword, {})
        k = pd["second_order_tensor"]
        bnd = pd["bc"]
        vdim = pd.get("ambient_dimension", sd.dim)
        if vdim != sd.dim and not (sd.dim == 2 and vdim == 3):
            raise NotImplementedError("ambient_dimension must be the grid dimension (or 3 for a 2-D grid)")
        spec = [pd.get(k) for k in ("specified_cells", "specified_faces", "specified_nodes")]
        partial = any(v is not None for v in spec)
        update = bool(pd.get("update_discretization", False))
        nsub = sps_nnz(sd.face_nodes)
        if hasattr(sd, "periodic_face_map"):
            # the right sub-faces share the numbers of the left ones (SubcellTopology.num_subfno_unique, _fvutils.py:91-160)
            import scipy.sparse as sps

            nnf = np.diff(sps.csc_matrix(sd.face_nodes).indptr)
            nsub -= int(nnf[np.asarray(sd.periodic_face_map)[1]].sum())
        subface = np.asarray(bnd.is_dir).size == nsub and nsub != sd.num_faces
        if not subface and np.asarray(bnd.is_dir).size != sd.num_faces:
            raise ValueError("boundary condition arrays must have one entry per face or per sub-face")
        if subface and (partial or update):
            raise NotImplementedError("partial discretization with conditions per sub-face is not covered")
        eta = pd.get("mpfa_eta", None)
        eta_sub = None
        if eta is None:
            eta = determine_eta(sd)
        elif np.asarray(eta).size != 1:
            # (the values follow the storage order of the caller's face_nodes, the device numbers sub-faces by the
            # sorted CSC arrays: _fvutils.py:78-90, 222-277)
            eta_sub = np.asarray(eta, dtype=float).ravel()
            if eta_sub.size != sps_nnz(sd.face_nodes):
                raise ValueError("size of eta must either be 1 or number of subfaces")
            eta_sub = eta_sub[subface_order(sd.face_nodes)]
            eta = 0.0
        note_ignored_parameters(pd, self.keyword)
        self._split.pop(id(sd), None)
        ent = self._contexts.get(id(sd))
        if not (ent is not None and ent[0] is sd and ent[1].has_discretization):
            # (a handle that already holds this grid's discretization reuses its buffers: it fits)
            nparts = plan_subproblems(sd, pd.get("partition_arguments"), self._free_device_bytes())
            if nparts > 1:
                plain = not (partial or update or subface or eta_sub is not None or hasattr(sd, "periodic_face_map")
                             or vdim != sd.dim or (sd.dim == 2 and plane_basis(grid_to_raw(sd)["nodes"]) is not None))
                if plain:
                    return self._discretize_in_pieces(sd, data, nparts, float(eta))
                import logging

                logging.getLogger("porepy_amd").warning(
                    "partition_arguments: partial updates, conditions per sub-face, per-sub-face eta, periodic or "
                    "tilted 2-D grids are discretized in one piece")
        ctx = self.context(sd)
        T = self._plane.get(id(sd))
        merge = self._periodic.get(id(sd))
        if merge is not None and (partial or update):
            raise NotImplementedError("periodic faces: full discretization only")
        kval = np.asarray(k.values, dtype=float)
        if T is not None:
            # rotate the tensor into the plane (mpfa.py:748-754)
            k2 = np.einsum("ia,abn,jb->ijn", T, kval, T)
            kval = np.zeros_like(kval)
            kval[:2, :2] = k2
            kval[2, 2] = 1.0
        order = None
        if subface:
            # conditions per sub-face follow the storage order of the caller's face_nodes; the device
            # numbers sub-faces by the sorted CSC arrays (mpfa.py:761-768, _fvutils.py:78-90)
            order = subface_order(sd.face_nodes) if merge is None else merged_subface_order(sd.face_nodes, merge)
            flags_sub = bc_flags(bnd)[order]
            robin_sub = np.asarray(bnd.robin_weight, dtype=float)[order]
            bnd = _FaceBC(sd.num_faces)  # per-face placeholders; the sub-face arrays take over below
        ctx.set_params(kval, bc_flags(bnd), np.asarray(bnd.robin_weight, dtype=float),
                       float(eta), eta_sub)
        if subface:
            ctx.set_subface_bc(flags_sub, robin_sub)
        if self.lazy:
            # proxies of the previous discretization that only `data` still refers to are dropped, not fetched
            # (a proxy the caller holds elsewhere stays alive and fetches its values before they are overwritten)
            for name, _ in _KEYS:
                if isinstance(md.get(name), LazyCsr):
                    del md[name]
        rows = None
        try:
            if partial:
                # node-list launch: only the interaction regions around the active faces
                # (mpfa.py:178-204; active sets as _fvutils.py:1260-1462)
                active_cells, active_faces = active_indices(sd, *spec)
                keep = update and ctx.has_discretization
                ctx.discretize_faces(active_faces, keep_other_rows=keep)
                rows = None if keep else active_faces
            else:
                ctx.discretize(rebuild_topology=bool(pd.get("hip_rebuild_topology", False)))
                active_cells, active_faces = np.arange(sd.num_cells), np.arange(sd.num_faces)
        except _lib.PorefvError as e:
            if e.status == 1:  # same exception type and text as the reference
                raise ValueError("Error in inversion of local linear systems") from e
            if e.status == 2:
                raise AssertionError(e.message) from e
            raise
        lift = None
        if sd.dim == 2 and vdim == 3:
            # vector sources live in the ambient space: append the map onto the plane of the
            # grid, column block by column block (mpfa.py:422-463)
            import scipy.sparse as sps

            basis = T if T is not None else np.eye(2, 3)
            lift = sps.kron(sps.identity(sd.num_cells, format="csr"), sps.csr_matrix(basis), format="csr")
        elif sd.dim == 2 and T is not None:
            # ambient_dimension == 2 on a grid outside the xy-plane: what the reference's un-rotation leaves
            lift = planar_source_map(sd, T)
        simple = merge is None and order is None and rows is None and not (partial and update)

        def lifted(m, L=lift):
            m = (m @ L).tocsr()
            m.sort_indices()
            return m

        for name, which in _KEYS:
            if self.lazy and simple:
                # (a tilted 2-D grid: only the vector-source matrices need the lift -- applied when one of them is
                # fetched; flux, bound_flux and the pressure traces stay plain device-resident proxies)
                if lift is not None and "vector_source" in name:
                    nr, _, _ = ctx.matrix_info(which)
                    md[name] = LazyCsr(ctx, which, post=lifted, shape=(nr, lift.shape[1]), right=lift)
                else:
                    md[name] = LazyCsr(ctx, which)
                continue
            new = ctx.matrix(which, rows=rows)
            if merge is not None and (order is None or "vector_source" in name):
                # (with conditions per sub-face only the vector-source matrices have face rows: mpfa.py:1117-1147)
                new = merge.copy_rows(new, trace=name.startswith("bound_pressure"))
            if lift is not None and "vector_source" in name:
                new = (new @ lift).tocsr()
                new.sort_indices()
            if order is not None and "vector_source" not in name and not np.array_equal(order, np.arange(order.size)):
                new = renumber_subfaces(new, order, cols=name in ("bound_flux", "bound_pressure_face"))
            if partial and update and rows is not None and name in md:
                # update without device history: splice the recomputed rows into the caller's
                # matrices (mpfa.py:466-485)
                old = md[name].tolil()
                old[active_faces] = new[active_faces]
                new = old.tocsr()
            md[name] = new
        # side effect of the reference's find_active_indices (_fvutils.py:346-353)
        pd["active_cells"] = active_cells
        pd["active_faces"] = active_faces

    # ---- many small grids at once -------------------------------------------------------------------------------
    def _batchable(self, sd, data: dict):
        """eta if (sd, data) can join a disjoint union -- plain inputs: full discretization, conditions per face, one
        continuity point, no periodic faces, no ``partition_arguments`` -- else None (the single-grid path takes it)."""
        if sd.dim < 2 or self.lazy or hasattr(sd, "periodic_face_map"):
            return None
        pd = data[PARAMETERS][self.keyword]
        if any(pd.get(k) is not None for k in ("specified_cells", "specified_faces", "specified_nodes")):
            return None
        if pd.get("update_discretization") or pd.get("partition_arguments") or pd.get("hip_rebuild_topology"):
            return None
        vdim = pd.get("ambient_dimension", sd.dim)
        if vdim != sd.dim and not (sd.dim == 2 and vdim == 3):
            return None
        if np.asarray(pd["bc"].is_dir).size != sd.num_faces:
            return None
        eta = pd.get("mpfa_eta", None)
        if eta is None:
            return float(determine_eta(sd))
        return float(np.asarray(eta).ravel()[0]) if np.asarray(eta).size == 1 else None

    def discretize_batch(self, items) -> dict:
        """``discretize(sd, data)`` for every pair of ``items`` -- the grids of a mixed-dimensional model (52 fracture
        planes of a few dozen cells each: numerics/ad/ad_utils.py:288-308 calls them one by one) -- with all grids of
        one dimension and one continuity point discretized as ONE disjoint union on the device: one upload, one
        topology + symbolic phase, one launch of each kernel over all interaction regions, one fetch per matrix,
        instead of a handle and ~40 launches of ~5 us per grid.  The interaction regions of different grids share
        nothing, so every block of the union's matrices is the matrix of its grid (same kernels, same inputs per
        region: the same bits as the single-grid path).  Pairs with special inputs go through ``discretize``.
        Returns {"unions": number of device discretizations, "batched": grids in them, "single": grids taken alone}."""
        import scipy.sparse as sps

        groups: dict = {}
        single = []
        for sd, data in items:
            eta = self._batchable(sd, data)
            if eta is None:
                single.append((sd, data))
            else:
                groups.setdefault((int(sd.dim), eta), []).append((sd, data))
        stats = {"unions": 0, "batched": 0, "single": 0}
        for (dim, eta), members in groups.items():
            if len(members) < 2:
                single.extend(members)
                continue
            raws, planes, kvals, flags, robin = [], [], [], [], []
            for sd, data in members:
                pd = data[PARAMETERS][self.keyword]
                note_ignored_parameters(pd, self.keyword)
                raw = grid_to_raw(sd)
                T = None
                if dim == 2:
                    T = plane_basis(raw["nodes"])
                    if T is not None:
                        for k in ("nodes", "face_normals", "face_centers", "cell_centers"):
                            loc = np.zeros_like(raw[k])
                            loc[:2] = T @ raw[k]
                            raw[k] = loc
                kv = np.asarray(pd["second_order_tensor"].values, dtype=float)
                if T is not None:
                    k2 = np.einsum("ia,abn,jb->ijn", T, kv, T)
                    kv = np.zeros_like(kv)
                    kv[:2, :2] = k2
                    kv[2, 2] = 1.0
                raws.append(raw)
                planes.append(T)
                kvals.append(kv)
                flags.append(bc_flags(pd["bc"]))
                robin.append(np.asarray(pd["bc"].robin_weight, dtype=float))
            nn = np.cumsum([0] + [r["nodes"].shape[1] for r in raws])
            nf = np.cumsum([0] + [r["face_centers"].shape[1] for r in raws])
            nc = np.cumsum([0] + [r["cell_centers"].shape[1] for r in raws])

            def cat_csc(ptr_key, idx_key, row_offsets):
                ptr, idx, base = [np.zeros(1, dtype=np.int64)], [], 0
                for r, off in zip(raws, row_offsets):
                    ptr.append(np.asarray(r[ptr_key][1:], dtype=np.int64) + base)
                    idx.append(np.asarray(r[idx_key], dtype=np.int64) + off)
                    base += int(r[ptr_key][-1])
                return np.concatenate(ptr).astype(np.int32), np.concatenate(idx).astype(np.int32)

            cf_ptr, cf_idx = cat_csc("cf_indptr", "cf_indices", nf[:-1])
            fn_ptr, fn_idx = cat_csc("fn_indptr", "fn_indices", nn[:-1])
            union = {"dim": dim, "name": "disjoint union of %d grids" % len(members),
                     "cf_indptr": cf_ptr, "cf_indices": cf_idx, "cf_sign": np.concatenate([r["cf_sign"] for r in raws]),
                     "fn_indptr": fn_ptr, "fn_indices": fn_idx,
                     "fracture_faces": np.concatenate([r["fracture_faces"] for r in raws])}
            for k in ("nodes", "face_normals", "face_centers", "cell_centers"):
                union[k] = np.ascontiguousarray(np.concatenate([r[k] for r in raws], axis=1))
            for k in ("face_areas", "cell_volumes"):
                union[k] = np.ascontiguousarray(np.concatenate([r[k] for r in raws]))
            ctx = _lib.Context(self.device, self._library)
            try:
                ctx.set_grid(union)
                ctx.set_params(np.ascontiguousarray(np.concatenate(kvals, axis=2)), np.concatenate(flags),
                               np.concatenate(robin), float(eta), None)
                try:
                    ctx.discretize(rebuild_topology=False)
                except _lib.PorefvError as e:
                    if e.status == 1:
                        raise ValueError("Error in inversion of local linear systems") from e
                    if e.status == 2:
                        raise AssertionError(e.message) from e
                    raise
                whole = {name: ctx.matrix(which) for name, which in _KEYS}
                # A = div @ flux of the union, on the device; its diagonal blocks are the systems of the grids
                ctx.assemble(np.zeros(int(nf[-1])), None, None)
                A_union = ctx.matrix(_lib.MAT_SYSTEM)
            finally:
                ctx.close()
            for g, (sd, data) in enumerate(members):
                pd = data[PARAMETERS][self.keyword]
                md = data.setdefault(DISCRETIZATION_MATRICES, {}).setdefault(self.keyword, {})
                f0, f1, c0, c1 = int(nf[g]), int(nf[g + 1]), int(nc[g]), int(nc[g + 1])
                vdim = pd.get("ambient_dimension", sd.dim)
                T = planes[g]
                lift = None
                if sd.dim == 2 and vdim == 3:
                    basis = T if T is not None else np.eye(2, 3)
                    lift = sps.kron(sps.identity(sd.num_cells, format="csr"), sps.csr_matrix(basis), format="csr")
                elif sd.dim == 2 and T is not None:
                    lift = planar_source_map(sd, T)
                cols = {"flux": (c0, c1), "bound_flux": (f0, f1), "bound_pressure_cell": (c0, c1),
                        "bound_pressure_face": (f0, f1), "vector_source": (dim * c0, dim * c1),
                        "bound_pressure_vector_source": (dim * c0, dim * c1)}
                for name, _ in _KEYS:
                    a, b = cols[name]
                    new = sps.csr_matrix(whole[name][f0:f1][:, a:b])
                    if lift is not None and "vector_source" in name:
                        new = (new @ lift).tocsr()
                    new.sort_indices()
                    md[name] = new
                pd["active_cells"] = np.arange(sd.num_cells)
                pd["active_faces"] = np.arange(sd.num_faces)
                self.invalidate(sd)  # (no handle of its own holds this grid's discretization:
                self._plane[id(sd)] = T  # the basis this member was discretized in
                A_g = sps.csr_matrix(A_union[c0:c1][:, c0:c1])  # assemble_matrix_rhs goes the way of a grid in pieces)
                A_g.sort_indices()
                self._split[id(sd)] = (sd, A_g)
            stats["unions"] += 1
            stats["batched"] += len(members)
        for sd, data in single:
            self.discretize(sd, data)
            stats["single"] += 1
        return stats

    def update_discretization(self, sd, data: dict) -> None:
        """Rediscretize around ``data["update_discretization"]["modified_cells" / "modified_faces"]``
        and keep every other row (mpfa.py:510-590 via _fvutils.partial_update_discretization,
        _fvutils.py:1090-1257).  Renumbering maps (``map_cells`` / ``map_faces``: the grid itself
        changed) lead to a full rediscretization of the new grid, which yields the same matrices."""
        if sd.dim < 2:
            return self._tpfa().discretize(sd, data)
        info = data.get("update_discretization", {})
        pd = data[PARAMETERS][self.keyword]
        cells = np.asarray(info.get("modified_cells", []), dtype=int)
        faces = np.asarray(info.get("modified_faces", []), dtype=int)
        ent = self._contexts.get(id(sd))
        remapped = "map_cells" in info or "map_faces" in info
        if remapped or ent is None or ent[0] is not sd or not ent[1].has_discretization:
            saved = {k: pd.pop(k, None) for k in ("specified_cells", "specified_faces", "specified_nodes")}
            if remapped:
                self._contexts.pop(id(sd), None)  # the grid itself changed: upload it again
                self._fingerprints.pop(id(sd), None)
            try:
                self.discretize(sd, data)
            finally:
                pd.update({k: v for k, v in saved.items() if v is not None})
            return
        if cells.size == 0 and faces.size == 0:
            return
        # the reference leaves these in the parameter dictionary (_fvutils.py:1175-1180)
        if cells.size:
            pd["specified_cells"] = cells
        if faces.size:
            pd["specified_faces"] = faces
        was = pd.get("update_discretization", False)
        pd["update_discretization"] = True
        try:
            self.discretize(sd, data)
        finally:
            pd["update_discretization"] = was

    def assemble_matrix_rhs(self, sd, data: dict):
        """(A, b) with A = div @ flux and b = -div @ bound_flux @ bc_values
        (- div @ vector_source @ g), computed on the device from the device-resident
        discretization (fv_elliptic.py:67-112)."""
        if sd.dim < 2:
            return self._tpfa().assemble_matrix_rhs(sd, data)
        pd = data[PARAMETERS][self.keyword]
        sp = self._split.get(id(sd))
        if sp is not None and sp[0] is sd:
            return self._split_system(sd, data)
        ent = self._contexts.get(id(sd))
        if ent is None or ent[0] is not sd:
            raise RuntimeError("discretize(sd, data) must run on this object before assemble_matrix_rhs")
        ctx = ent[1]
        ctx.assemble(np.asarray(pd["bc_values"], dtype=float), self._vector_source(sd, pd), None)
        return ctx.matrix(_lib.MAT_SYSTEM), ctx.rhs()

    def ad_flux_system(self, sd, data: dict, p, dk_dp=None, source=None, flux_jacobian: bool = False):
        """Residual and Jacobian of the flow equation for a pressure-dependent permeability, assembled on the
        device from the discretization of the last ``discretize(sd, data)`` (which must have run with
        K = K(p)): what the reference evaluates through ``AdTpfaFlux.diffusive_flux`` with an Mpfa base
        discretization and its forward AD (models/constitutive_laws.py:1195-1336, 1580-1721).  ``dk_dp``:
        d K_rs(c) / d p_c as a (3, 3, Nc) array.  Returns the face fluxes q; J = d(div q)/dp and
        -(div q - source) stay on the device as the active system -- ``newton_increment`` solves it there,
        ``context(sd).matrix(MAT_SYSTEM)`` / ``.rhs()`` copy them out; with ``flux_jacobian`` dq/dp is
        ``context(sd).matrix(MAT_FLUX_JACOBIAN)``."""
        if sd.dim < 2:
            raise NotImplementedError("the differentiable MPFA flux needs a 2-D or 3-D grid")
        pd = data[PARAMETERS][self.keyword]
        ent = self._contexts.get(id(sd))
        if ent is None or ent[0] is not sd or not ent[1].has_discretization:
            raise RuntimeError("discretize(sd, data) must run on this object before ad_flux_system")
        if self._plane.get(id(sd)) is not None or self._periodic.get(id(sd)) is not None:
            raise NotImplementedError("tilted or periodic grids are not covered by the differentiable flux")
        return ent[1].ad_flux_system(p, dk_dp, np.asarray(pd["bc_values"], dtype=float), self._vector_source(sd, pd),
                                     source, flux_jacobian=flux_jacobian)

    def newton_increment(self, sd, method: str = "bicgstab", rtol: float = 1e-10, maxit: int = 20000,
                         precond: str = "amg"):
        """Solve J dp = -(div q - source) left by ``ad_flux_system`` on the device.  Returns (dp, info)."""
        ent = self._contexts.get(id(sd))
        if ent is None or ent[0] is not sd:
            raise RuntimeError("ad_flux_system(sd, ...) first")
        return ent[1].solve(method=method, rtol=rtol, maxit=maxit, precond=precond)

    def _vector_source(self, sd, pd):
        """Cell-wise vector source in the coordinates the device discretized in."""
        vs = pd.get("vector_source", None)
        if vs is None:
            return None
        vs = np.asarray(vs, dtype=float)
        vdim = pd.get("ambient_dimension", sd.dim)
        if sd.dim == 2 and vdim == 3:
            T = self._plane.get(id(sd))
            basis = T if T is not None else np.eye(2, 3)
            vs = (vs.reshape(sd.num_cells, 3) @ basis.T).ravel()
        elif sd.dim == 2 and self._plane.get(id(sd)) is not None:
            vs = planar_source_map(sd, self._plane[id(sd)]) @ vs
        return vs

    # ---- solve (stand-in for SolutionStrategy.solve_linear_system) --------------------
    def solve(self, sd, data: dict, source=None, method: str = "bicgstab", rtol: float = 1e-12,
              maxit: int = 20000, x0=None, restart: int = 0, precond: str = "jacobi"):
        """Solve A p = b + source with a Krylov solver on the device (method: "bicgstab", "gmres"
        (restart = cycle length) or "cg"; precond: "jacobi" or "amg" = aggregation multigrid V-cycle),
        re-using the device-resident system.  Returns (p, info)."""
        if sd.dim < 2:
            return self._tpfa().solve(sd, data, source=source, method=method, rtol=rtol, maxit=maxit, x0=x0,
                                      restart=restart, precond=precond)
        pd = data[PARAMETERS][self.keyword]
        sp = self._split.get(id(sd))
        if sp is not None and sp[0] is sd:
            from .solvers import solve_csr

            A, b = self._split_system(sd, data, source)
            if self._split_ctx is None:
                self._split_ctx = _lib.Context(self.device, self._library)
            return solve_csr(A, b, method=method, rtol=rtol, maxit=maxit, restart=restart, precond=precond, x0=x0,
                             context=self._split_ctx)
        ent = self._contexts.get(id(sd))
        if ent is None or ent[0] is not sd:
            raise RuntimeError("discretize(sd, data) must run on this object before solve")
        ctx = ent[1]
        ctx.assemble(np.asarray(pd["bc_values"], dtype=float), self._vector_source(sd, pd), source)
        return ctx.solve(method=method, rtol=rtol, maxit=maxit, x0=x0, restart=restart, precond=precond)


def as_porepy_discretization(device: int = 0, library=None, lazy: bool = False):
    """Subclass of the reference's ``pp.Mpfa`` whose hot path runs on the MI355X.
    (``library`` is for tests that bind the host-emulation build; the product default is the
    gfx950 library.)  ``lazy``: the matrices stay on the device behind ``LazyCsr`` proxies (lazy.py) -- for models that
    assemble on the device (``porepy_amd.DeviceAssembly``); whatever else reads a matrix fetches it then."""
    import porepy as pp  # the reference; absent on the GPU box

    _device, _library, _lazy = device, library, bool(lazy)
    _RefMpfa = pp.Mpfa

    class HipMpfa(_RefMpfa):  # type: ignore[misc]
        def __init__(self, keyword: str):
            # pp.Mpfa.__init__ resolves ``super(pp.Mpfa, self)`` through the module attribute
            # (mpfa.py:62-63), which recurses once pp.Mpfa is rebound to this class — go to
            # its base (FVElliptic) directly
            super(_RefMpfa, self).__init__(keyword)
            self._hip = Mpfa(keyword, _device, _library, lazy=_lazy)

        def discretize(self, sd, data):
            # 1-D subdomains (fracture intersections: 103 of the 190 subdomains of the 52-fracture model) go to the
            # in-tree device TPFA through the host mirror, 0-D ones get its empty matrices (porepy_amd.Mpfa.discretize:
            # mpfa.py:690-723, 129-149 of the reference); PFV_DROPIN_LOWDIM_HOST=1 hands them to the reference's host code
            if sd.dim < 2 and os.environ.get("PFV_DROPIN_LOWDIM_HOST", "0") == "1":
                return super().discretize(sd, data)
            return self._hip.discretize(sd, data)

        def discretize_batch(self, items):
            """All (sd, data) pairs of one loop over a mixed-dimensional grid: grids of dimension >= 2 as disjoint
            unions on the device (``porepy_amd.Mpfa.discretize_batch``), the rest as upstream."""
            items = list(items)
            host = os.environ.get("PFV_DROPIN_LOWDIM_HOST", "0") == "1"
            for sd, data in items:
                if sd.dim < 2:
                    if host:
                        super().discretize(sd, data)
                    else:
                        self._hip.discretize(sd, data)
            return self._hip.discretize_batch([(sd, data) for sd, data in items if sd.dim >= 2])

        def discretize_piece(self, sd, data, piece, nparts):
            return self._hip.discretize_piece(sd, data, piece, nparts)

        def merge_piece_payloads(self, sd, data, payloads):
            return self._hip.merge_piece_payloads(sd, data, payloads)

        def update_discretization(self, sd, data):
            if sd.dim < 2:
                # (no partial path for the two-point scheme: re-discretized whole, as cheap as it gets)
                if os.environ.get("PFV_DROPIN_LOWDIM_HOST", "0") == "1":
                    return super().update_discretization(sd, data)
                return self._hip.discretize(sd, data)
            return self._hip.update_discretization(sd, data)

        def assemble_matrix_rhs(self, sd, data):
            if sd.dim < 2:
                # (host algebra on the stored matrices either way: fv_elliptic.py:67-112)
                return super().assemble_matrix_rhs(sd, data)
            return self._hip.assemble_matrix_rhs(sd, data)

    return HipMpfa
