"""``Biot`` — the reference's poro-elastic discretization (numerics/fv/biot.py:39-712) on the
device: the four MPSA matrices plus, per coupling tensor in ``scalar_vector_mappings``, the five
coupling terms ``scalar_gradient``, ``displacement_divergence``,
``boundary_displacement_divergence``, ``mpsa_consistency`` and ``bound_displacement_pressure``
(dictionaries keyed like ``scalar_vector_mappings``), all from one pass of the interaction-region
kernel.  Like the reference class it does not assemble; the poromechanics model combines the terms.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from .grid import grid_to_raw
from .mpfa import (determine_eta, estimate_device_bytes, note_ignored_parameters, partition_cells, plan_subproblems,
                   sps_nnz, subface_order)
from .partial import active_indices
from .mpsa import Mpsa
from .mpsa import _KEYS as _MECH_KEYS
from .params import DISCRETIZATION_MATRICES, PARAMETERS, SecondOrderTensor

_TERMS = (
    ("scalar_gradient", 0),
    ("displacement_divergence", 1),
    ("boundary_displacement_divergence", 2),
    ("mpsa_consistency", 3),
    ("bound_displacement_pressure", 4),
)


class Biot(Mpsa):
    def __init__(self, keyword: str = "mechanics", device: int = 0, library=None):
        super().__init__(keyword, device, library)
        self.displacement_divergence_matrix_key = "displacement_divergence"
        self.bound_displacement_divergence_matrix_key = "boundary_displacement_divergence"
        self.scalar_gradient_matrix_key = "scalar_gradient"
        self.consistency_matrix_key = "mpsa_consistency"
        self.bound_pressure_matrix_key = "bound_displacement_pressure"

    def ndof(self, sd) -> int:
        return sd.num_cells * (1 + sd.dim)

    def assemble_matrix_rhs(self, sd, data):
        raise NotImplementedError("This class cannot be used for assembly (as in the reference, biot.py:125-149)")

    def discretize(self, sd, data: dict) -> None:
        pd = data[PARAMETERS][self.keyword]
        md = data.setdefault(DISCRETIZATION_MATRICES, {}).setdefault(self.keyword, {})
        spec = [pd.get(k) for k in ("specified_cells", "specified_faces", "specified_nodes")]
        partial = any(v is not None for v in spec)
        update = bool(pd.get("update_discretization", False))
        C = pd["fourth_order_tensor"]
        bnd = pd["bc"]
        if np.asarray(bnd.is_dir).ndim != 2:
            raise AttributeError("MPSA must be given a vectorial boundary condition")
        basis = getattr(bnd, "basis", None)
        if basis is not None and np.asarray(basis).ndim != 3:
            basis = None
        mappings = pd["scalar_vector_mappings"]
        keys = list(mappings.keys())
        alphas = []
        for k in keys:
            a = mappings[k]
            if isinstance(a, (float, int)):
                a = SecondOrderTensor(float(a) * np.ones(sd.num_cells))
            alphas.append(np.asarray(a.values, dtype=float))
        eta = pd.get("mpsa_eta", None)
        eta_sub = None
        if eta is None:
            eta = determine_eta(sd)
        elif np.asarray(eta).size != 1:
            # one continuity point per sub-face (_fvutils.py:222-277), in the storage order of the caller's face_nodes
            eta_sub = np.asarray(eta, dtype=float).ravel()
            if eta_sub.size != sps_nnz(sd.face_nodes):
                raise ValueError("size of eta must either be 1 or number of subfaces")
            if partial or update or pd.get("partition_arguments"):
                raise NotImplementedError("continuity points per sub-face: full Biot discretization in one piece only")
            eta_sub = eta_sub[subface_order(sd.face_nodes)]
            eta = 0.0
        # the reference's Biot reconstructs the displacement traces at the continuity points whatever
        # ``reconstruction_eta`` says (biot.py:803-805 calls _reconstruct_displacement with eta; the key is never read):
        # the same here, with a note
        note_ignored_parameters(pd, self.keyword, {
            "reconstruction_eta": "Biot reconstructs the displacement traces at the continuity points of `mpsa_eta`, as "
                                  "the reference's Biot does (it never reads the key)"})
        self._split.pop(id(sd), None)
        ent = self._contexts.get(id(sd))
        if alphas and not (partial or update) and not (
                ent is not None and ent[0] is sd and ent[1].has_biot_discretization):
            nparts = plan_subproblems(sd, pd.get("partition_arguments"), _lib.free_device_bytes(self.device, self._library),
                                      need=(sd.dim + 1) * estimate_device_bytes(sd), what="Biot")
            if nparts > 1:
                return self._biot_in_pieces(sd, data, nparts, float(eta), basis, keys, alphas)
        ctx = self.context(sd)
        is_rob = getattr(bnd, "is_rob", None)
        ctx.mpsa_set_params(np.asarray(C.values), sd.cell_volumes, bnd.is_dir, bnd.is_neu, float(eta), is_rob=is_rob,
                            robin_weight=getattr(bnd, "robin_weight", None) if is_rob is not None else None,
                            basis=basis)
        ctx.mpsa_set_subface_eta(eta_sub)  # (None: the scalar eta of mpsa_set_params)
        ctx.biot_set_alphas(alphas)
        if partial and not alphas:
            return Mpsa.discretize(self, sd, data)
        if partial and update and not ctx.has_biot_discretization:
            # an update needs a complete set of coupling terms on the handle (an MPSA-only discretization,
            # or none, came before): discretize everything -- the rows outside the active set must not
            # come back as zeros
            partial = False
        active_cells, active_faces = np.arange(sd.num_cells), np.arange(sd.num_faces)
        try:
            if partial:
                # Node-list launch around the active faces (biot.py:326-345, 400-560: the reference
                # discretizes the sub-grid of the active cells and keeps the rows of the active faces
                # / cells).  Face rows are complete for the active faces.  A cell row sums over the
                # nodes of the cell: it is complete for cells all of whose nodes were recomputed -
                # only those are written (the reference's rows of the other active cells carry the
                # artificial boundary of its sub-grid); under an update the nodes that were not
                # recomputed still hold their tables, so every cell touching a recomputed node is
                # rewritten.
                active_cells, active_faces = active_indices(sd, *spec)
                keep = update and ctx.has_biot_discretization
                fn = sd.face_nodes.tocsc()
                mark = np.zeros(sd.num_nodes)
                for f in active_faces:
                    mark[fn.indices[fn.indptr[f]: fn.indptr[f + 1]]] = 1.0
                cn = sd.cell_nodes().astype(float)  # (Nn, Nc)
                hit = np.asarray(cn.T @ mark).ravel()
                tot = np.asarray(cn.sum(axis=0)).ravel()
                cells = np.flatnonzero(hit > 0) if keep else np.flatnonzero(hit == tot)
                ctx.biot_discretize_faces(active_faces, cells, keep_other_rows=keep)
            elif alphas:
                ctx.biot_discretize()
            else:
                ctx.mpsa_discretize()
        except _lib.PorefvError as e:
            if e.status == 1:
                raise ValueError("Error in inversion of local linear systems") from e
            if e.status == 2:
                raise AssertionError(e.message) from e
            raise
        from .mpsa import _note_contrast_regions

        _note_contrast_regions(ctx)
        for name, which in _MECH_KEYS:
            md[name] = ctx.matrix(which)
        for name, term in _TERMS:
            md[name] = {k: ctx.biot_matrix(term, i) for i, k in enumerate(keys)}
        pd["active_cells"] = active_cells
        pd["active_faces"] = active_faces

    def _biot_in_pieces(self, sd, data: dict, nparts: int, eta: float, basis, keys, alphas) -> None:
        """Memory-bounded discretization (biot.py:246-398 with _fvutils.subproblems): as
        ``Mpsa._discretize_in_pieces`` for the four MPSA matrices and the two coupling terms with face rows
        (rows of the faces of a piece's own cells, faces between two pieces averaged); the three terms with cell
        rows take the rows of a piece's own cells (all nodes of an own cell are complete in the piece)."""
        import scipy.sparse as sps

        from .distributed import extract_subdomain

        pd = data[PARAMETERS][self.keyword]
        md = data.setdefault(DISCRETIZATION_MATRICES, {}).setdefault(self.keyword, {})
        raw = grid_to_raw(sd)
        nd, nc, nf = sd.dim, sd.num_cells, sd.num_faces
        Cv = np.asarray(pd["fourth_order_tensor"].values, dtype=float)
        bnd = pd["bc"]
        is_dir, is_neu = np.asarray(bnd.is_dir, bool), np.asarray(bnd.is_neu, bool)
        is_rob = getattr(bnd, "is_rob", None)
        is_rob = None if is_rob is None or not np.any(is_rob) else np.asarray(is_rob, bool)
        robw = getattr(bnd, "robin_weight", None)
        owner = partition_cells(sd, nparts)
        comp = np.arange(nd)[None, :]
        # (rows per face?, columns: "c" cells, "C" nd x cells, "F" nd x faces)
        kinds = {"stress": (True, "C"), "bound_stress": (True, "F"), "bound_displacement_cell": (True, "C"),
                 "bound_displacement_face": (True, "F"), "scalar_gradient": (True, "c"),
                 "bound_displacement_pressure": (True, "c"), "displacement_divergence": (False, "C"),
                 "boundary_displacement_divergence": (False, "F"), "mpsa_consistency": (False, "c")}
        width = {"c": nc, "C": nd * nc, "F": nd * nf}
        acc = {(name, None): ([], [], []) for name, _ in _MECH_KEYS}
        acc.update({(name, k): ([], [], []) for name, _ in _TERMS for k in keys})
        count = np.zeros(nf, dtype=np.int64)
        for r in range(nparts):
            if not np.any(owner == r):
                continue
            lp = extract_subdomain(raw, owner, r)
            ctx = _lib.Context(self.device, self._library)
            try:
                ctx.set_grid(lp.raw)
                art = lp.artificial_boundary
                ldir, lneu = is_dir[:, lp.face_gid].copy(), is_neu[:, lp.face_gid].copy()
                ldir[:, art], lneu[:, art] = False, True
                lrob = None
                if is_rob is not None:
                    lrob = is_rob[:, lp.face_gid].copy()
                    lrob[:, art] = False
                ctx.mpsa_set_params(np.ascontiguousarray(Cv[:, :, lp.cell_gid]), lp.raw["cell_volumes"], ldir, lneu, eta,
                                    is_rob=lrob,
                                    robin_weight=None if (lrob is None or robw is None) else
                                    np.ascontiguousarray(np.asarray(robw, float)[:, :, lp.face_gid]),
                                    basis=None if basis is None else np.ascontiguousarray(np.asarray(basis, float)[:, :, lp.face_gid]))
                ctx.biot_set_alphas([np.ascontiguousarray(a[:, :, lp.cell_gid]) for a in alphas])
                try:
                    ctx.biot_discretize(rebuild_topology=True)
                except _lib.PorefvError as e:
                    if e.status == 1:
                        raise ValueError("Error in inversion of local linear systems") from e
                    if e.status == 2:
                        raise AssertionError(e.message) from e
                    raise
                cfp = lp.raw["cf_indptr"]
                own_faces = np.unique(lp.raw["cf_indices"][: cfp[lp.n_own]])
                count[lp.face_gid[own_faces]] += 1
                frows_l = (nd * own_faces[:, None] + comp).ravel()
                frows_g = (nd * lp.face_gid[own_faces][:, None] + comp).ravel()
                cmap = {"c": lp.cell_gid, "C": (nd * lp.cell_gid[:, None] + comp).ravel(),
                        "F": (nd * lp.face_gid[:, None] + comp).ravel()}

                def take(M, name, key):
                    face_rows, ck = kinds[name]
                    if face_rows:
                        sub = M[frows_l].tocoo()
                        rows = frows_g[sub.row]
                    else:
                        sub = M[: lp.n_own].tocoo()
                        rows = lp.cell_gid[sub.row]
                    rr, cc, vv = acc[(name, key)]
                    rr.append(rows)
                    cc.append(cmap[ck][sub.col])
                    vv.append(sub.data)

                for name, which in _MECH_KEYS:
                    take(ctx.matrix(which), name, None)
                for name, term in _TERMS:
                    for i, k in enumerate(keys):
                        take(ctx.biot_matrix(term, i), name, k)
            finally:
                ctx.close()
        fscale = np.repeat(1.0 / np.maximum(count, 1), nd)

        def merged(name, key):
            face_rows, ck = kinds[name]
            rr, cc, vv = (np.concatenate(x) if x else np.zeros(0) for x in acc[(name, key)])
            nrows = nd * nf if face_rows else nc
            if face_rows:
                vv = vv * fscale[rr.astype(np.int64)]
            M = sps.coo_matrix((vv, (rr, cc)), shape=(nrows, width[ck])).tocsr()
            M.sum_duplicates()
            M.sort_indices()
            return M

        for name, _ in _MECH_KEYS:
            md[name] = merged(name, None)
        for name, _ in _TERMS:
            md[name] = {k: merged(name, k) for k in keys}
        self._contexts.pop(id(sd), None)
        # (Mpsa.solve answers with a clear NotImplementedError instead of "discretize first")
        if not hasattr(self, "_pieces_without_system"):
            self._pieces_without_system = {}
        self._pieces_without_system[id(sd)] = sd
        pd["active_cells"] = np.arange(nc)
        pd["active_faces"] = np.arange(nf)

    # update_discretization: Mpsa's (modified_cells / modified_faces -> specified_* + update flag)


def as_porepy_biot(device: int = 0, library=None):
    """Subclass of the reference's ``pp.Biot`` running on the MI355X (``pp.Biot = as_porepy_biot()``;
    ``pp.ad.BiotAd`` resolves ``pp.Biot`` at call time, numerics/ad/discretizations.py:87-131)."""
    import porepy as pp  # the reference; absent on the GPU box

    _device, _library = device, library
    _Ref = pp.Biot

    class HipBiot(_Ref):  # type: ignore[misc]
        def __init__(self, keyword: str = "mechanics"):
            _Ref.__init__(self, keyword)
            self._hip = Biot(keyword, _device, _library)

        def discretize(self, sd, sd_data):
            if sd.dim < 2:
                return _Ref.discretize(self, sd, sd_data)
            return self._hip.discretize(sd, sd_data)

        def update_discretization(self, sd, sd_data):
            if sd.dim < 2:
                return _Ref.update_discretization(self, sd, sd_data)
            return self._hip.update_discretization(sd, sd_data)

    return HipBiot
