"""``Mpsa`` — the reference's MPSA-W stress discretization operator on an MI355X.

Operator API of ``pp.Mpsa`` (numerics/fv/mpsa.py:63-529): ``Mpsa(keyword)``, ``ndof``,
``discretize(sd, data)``, ``assemble_matrix_rhs(sd, data)``, four ``*_matrix_key`` attributes;
parameters ``fourth_order_tensor``, ``bc`` (vectorial), ``bc_values`` ((nd, Nf) raveled "F"),
``source``, ``mpsa_eta``.  Unknown ordering is cell-major, component-minor (u[nd*c + a]).
Covered: component-wise Dirichlet / Neumann / Robin conditions, also in a face-wise rotated or skewed
basis (``bc.basis``).  Conditions given per sub-face raise an error.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from .grid import grid_to_raw
from .mpfa import (determine_eta, estimate_device_bytes, note_ignored_parameters, partition_cells, plan_subproblems,
                   sps_nnz, subface_order)
from .partial import active_indices
from .params import DISCRETIZATION_MATRICES, PARAMETERS

_KEYS = (
    ("stress", _lib.MAT_STRESS),
    ("bound_stress", _lib.MAT_BOUND_STRESS),
    ("bound_displacement_cell", _lib.MAT_BOUND_DISPLACEMENT_CELL),
    ("bound_displacement_face", _lib.MAT_BOUND_DISPLACEMENT_FACE),
)



def _note_contrast_regions(ctx) -> None:
    """Say what the library did about interaction regions whose sub-cells' stiffnesses are more than 1e6 apart
    (``pfv_stats.mpsa_contrast_regions`` / ``mpsa_max_contrast``, csrc/mpsa.inc: mpsa_contrast_scan): assembled and
    eliminated in double-double arithmetic (the default; equal to the reference's result to 1e-10 and better) --
    or, with ``PFV_MPSA_DD=0``, left to the FP64 body, which is then only within ~eps x contrast of the reference."""
    import logging
    import os

    st = ctx.stats()
    n = int(st.get("mpsa_contrast_regions", 0))
    if n <= 0:
        return
    log = logging.getLogger("porepy_amd")
    if os.environ.get("PFV_MPSA_DD", "1") == "0":
        log.warning("Mpsa: %d interaction regions with stiffness contrasts up to %.1e between cells sharing a node and "
                    "PFV_MPSA_DD=0: the FP64 condensed systems agree with the reference only to ~1e-16 x contrast there",
                    n, st.get("mpsa_max_contrast", float("nan")))
    else:
        log.info("Mpsa: %d interaction regions with stiffness contrasts up to %.1e between cells sharing a node were "
                 "assembled and eliminated in double-double arithmetic", n, st.get("mpsa_max_contrast", float("nan")))

class Mpsa:
    def __init__(self, keyword: str, device: int = 0, library=None):
        self.keyword = keyword
        self.device = device
        self._library = library
        self.stress_matrix_key = "stress"
        self.bound_stress_matrix_key = "bound_stress"
        self.bound_displacement_cell_matrix_key = "bound_displacement_cell"
        self.bound_displacement_face_matrix_key = "bound_displacement_face"
        self._contexts: dict = {}
        self._split: dict = {}  # id(sd) -> (sd, A) of a grid discretized in pieces (no whole-grid handle exists)

    def ndof(self, sd) -> int:
        return sd.dim * sd.num_cells

    def context(self, sd) -> _lib.Context:
        ent = self._contexts.get(id(sd))
        if ent is None or ent[0] is not sd:
            if sd.dim not in (2, 3):
                raise NotImplementedError("porepy_amd.Mpsa covers 2-D and 3-D grids")
            ctx = _lib.Context(self.device, self._library)
            ctx.set_grid(grid_to_raw(sd))
            self._contexts[id(sd)] = (sd, ctx)
            return ctx
        return ent[1]

    def discretize(self, sd, data: dict) -> None:
        pd = data[PARAMETERS][self.keyword]
        md = data.setdefault(DISCRETIZATION_MATRICES, {}).setdefault(self.keyword, {})
        C = pd["fourth_order_tensor"]
        bnd = pd["bc"]
        if np.asarray(bnd.is_dir).ndim != 2:
            # same failure mode as the reference (mpsa.py:658-659)
            raise AttributeError("MPSA must be given a vectorial boundary condition")
        if hasattr(sd, "periodic_face_map"):
            # as the reference (mpsa.py:661-664)
            raise NotImplementedError("Periodic boundary conditions are not implemented for Mpsa")
        basis = getattr(bnd, "basis", None)
        if basis is not None and np.asarray(basis).ndim != 3:
            basis = None
        spec = [pd.get(k) for k in ("specified_cells", "specified_faces", "specified_nodes")]
        partial = any(v is not None for v in spec)
        update = bool(pd.get("update_discretization", False))
        nsub = sps_nnz(sd.face_nodes)
        eta = pd.get("mpsa_eta", None)
        eta_sub = None
        if eta is None:
            eta = determine_eta(sd)
        elif np.asarray(eta).size != 1:
            # one continuity point per sub-face, used as given also on the boundary (mpsa.py:293-303, 647-652;
            # _fvutils.py:222-277); the values follow the storage order of the caller's face_nodes
            eta_sub = np.asarray(eta, dtype=float).ravel()
            if eta_sub.size != nsub:
                raise ValueError("size of eta must either be 1 or number of subfaces")
            eta_sub = eta_sub[subface_order(sd.face_nodes)]
            eta = 0.0
        hf_eta = pd.get("reconstruction_eta", None)
        if hf_eta is not None and np.asarray(hf_eta).size != 1:
            # one reconstruction point per sub-face (compute_dist_face_cell with an array, _fvutils.py:222-277), in the
            # storage order of the caller's face_nodes
            hf_eta = np.asarray(hf_eta, dtype=float).ravel()
            if hf_eta.size != nsub:
                raise ValueError("size of eta must either be 1 or number of subfaces")
            hf_eta = hf_eta[subface_order(sd.face_nodes)]
        elif hf_eta is not None and eta_sub is None and float(hf_eta) == float(eta):
            hf_eta = None  # the continuity points themselves
        note_ignored_parameters(pd, self.keyword, {
            "inverter": "the local systems are inverted by the device kernel (register Gauss-Jordan); the reference's "
                        "numba / python choice does not apply"})
        is_rob = getattr(bnd, "is_rob", None)
        subface = np.asarray(bnd.is_dir).shape[1] == nsub and nsub != sd.num_faces
        self._split.pop(id(sd), None)
        ent = self._contexts.get(id(sd))
        if not (ent is not None and ent[0] is sd and ent[1].has_mpsa_discretization):
            # the expanded rows of an interaction region are nd x as wide as the MPFA tables, the matrices nd^2 x
            nparts = plan_subproblems(sd, pd.get("partition_arguments"), _lib.free_device_bytes(self.device, self._library),
                                      need=sd.dim * estimate_device_bytes(sd), what="MPSA")
            if nparts > 1:
                if not (partial or update or subface or eta_sub is not None or hf_eta is not None):
                    return self._discretize_in_pieces(sd, data, nparts, float(eta), basis)
                import logging

                logging.getLogger("porepy_amd").warning(
                    "partition_arguments: partial updates and conditions per sub-face are discretized in one piece")
        ctx = self.context(sd)
        order = None
        if subface:
            # conditions per sub-face (mpsa.py:712-720): they follow the storage order of the caller's face_nodes,
            # the device numbers sub-faces by the sorted CSC arrays
            if partial or update:
                raise NotImplementedError("partial discretization with conditions per sub-face is not covered")
            order = subface_order(sd.face_nodes)
            if basis is not None and np.asarray(basis).shape != (sd.dim, sd.dim, nsub):
                raise ValueError("the basis of conditions per sub-face must have one entry per sub-face")
            nd, nf = sd.dim, sd.num_faces
            ctx.mpsa_set_params(np.asarray(C.values), sd.cell_volumes, np.zeros((nd, nf), bool), np.ones((nd, nf), bool),
                                float(eta))  # per-face placeholders; the sub-face arrays take over
            rob_sub = None if is_rob is None else np.asarray(is_rob, bool)[:, order]
            rw = getattr(bnd, "robin_weight", None)
            ctx.mpsa_set_subface_bc(np.asarray(bnd.is_dir, bool)[:, order], np.asarray(bnd.is_neu, bool)[:, order],
                                    rob_sub, None if (rob_sub is None or rw is None) else np.asarray(rw, float)[:, :, order],
                                    basis_sub=None if basis is None else np.asarray(basis, float)[:, :, order])
        else:
            ctx.mpsa_set_params(np.asarray(C.values), sd.cell_volumes, bnd.is_dir, bnd.is_neu, float(eta),
                                is_rob=is_rob,
                                robin_weight=getattr(bnd, "robin_weight", None) if is_rob is not None else None,
                                basis=basis)
        ctx.mpsa_set_subface_eta(eta_sub)  # (None: the scalar eta of mpsa_set_params)
        if hf_eta is not None and (partial or update):
            raise NotImplementedError("reconstruction_eta with partial updates is not covered")
        # displacement traces reconstructed at x_f + hf_eta (x_v - x_f) (mpsa.py:185, 757-761, 1187-1266)
        ctx.mpsa_set_reconstruction_eta(hf_eta)
        rows = None
        try:
            if partial:
                # node-list launch around the active faces (mpsa.py:209-216, 383-416)
                active_cells, active_faces = active_indices(sd, *spec)
                keep = update and ctx.has_mpsa_discretization
                ctx.mpsa_discretize_faces(active_faces, keep_other_rows=keep)
                if not keep:
                    rows = (sd.dim * active_faces[:, None] + np.arange(sd.dim)[None, :]).ravel()
            else:
                ctx.mpsa_discretize()
                active_cells, active_faces = np.arange(sd.num_cells), np.arange(sd.num_faces)
        except _lib.PorefvError as e:
            if e.status == 1:
                raise ValueError("Error in inversion of local linear systems") from e
            if e.status == 2:
                raise AssertionError(e.message) from e
            raise
        _note_contrast_regions(ctx)
        for name, which in _KEYS:
            new = ctx.matrix(which, rows=rows)
            if order is not None and not np.array_equal(order, np.arange(order.size)):
                # sub-face blocks from the device's numbering back to the caller's (rows of stress / bound_stress,
                # columns of the two boundary matrices)
                import scipy.sparse as sps

                blk = (sd.dim * order[:, None] + np.arange(sd.dim)[None, :]).ravel()  # device -> caller
                coo = sps.coo_matrix(new)
                r = blk[coo.row] if name in ("stress", "bound_stress") else coo.row
                c = blk[coo.col] if name in ("bound_stress", "bound_displacement_face") else coo.col
                new = sps.csr_matrix((coo.data, (r, c)), shape=new.shape)
                new.sort_indices()
            if partial and update and rows is not None and name in md:
                old = md[name].tolil()
                old[rows] = new[rows]
                new = old.tocsr()
            md[name] = new
        pd["active_cells"] = active_cells
        pd["active_faces"] = active_faces

    def _discretize_in_pieces(self, sd, data: dict, nparts: int, eta: float, basis) -> None:
        """Memory-bounded discretization (mpsa.py:201-207, 245-380 with _fvutils.subproblems, _fvutils.py:414-539):
        as ``Mpfa._discretize_in_pieces`` -- cell partition, one node-ring of overlap, every piece discretized on
        the device on its own, rows of the faces of its own cells merged on the host (faces between two pieces
        are computed by both and averaged), one piece resident at a time; the system matrix div_nd @ stress comes
        from the pieces too."""
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
        ncols = {"stress": nd * nc, "bound_stress": nd * nf, "bound_displacement_cell": nd * nc,
                 "bound_displacement_face": nd * nf}
        acc = {name: ([], [], []) for name, _ in _KEYS}
        sysacc = ([], [], [])
        count = np.zeros(nf, dtype=np.int64)
        comp = np.arange(nd)[None, :]
        for r in range(nparts):
            if not np.any(owner == r):
                continue
            lp = extract_subdomain(raw, owner, r)
            ctx = _lib.Context(self.device, self._library)
            try:
                ctx.set_grid(lp.raw)
                art = lp.artificial_boundary
                ldir, lneu = is_dir[:, lp.face_gid].copy(), is_neu[:, lp.face_gid].copy()
                ldir[:, art], lneu[:, art] = False, True  # never touches a node of an own cell
                lrob = None
                if is_rob is not None:
                    lrob = is_rob[:, lp.face_gid].copy()
                    lrob[:, art] = False
                ctx.mpsa_set_params(np.ascontiguousarray(Cv[:, :, lp.cell_gid]), lp.raw["cell_volumes"], ldir, lneu, eta,
                                    is_rob=lrob,
                                    robin_weight=None if (lrob is None or robw is None) else
                                    np.ascontiguousarray(np.asarray(robw, float)[:, :, lp.face_gid]),
                                    basis=None if basis is None else np.ascontiguousarray(np.asarray(basis, float)[:, :, lp.face_gid]))
                try:
                    ctx.mpsa_discretize(rebuild_topology=True)
                except _lib.PorefvError as e:
                    if e.status == 1:
                        raise ValueError("Error in inversion of local linear systems") from e
                    if e.status == 2:
                        raise AssertionError(e.message) from e
                    raise
                cfp = lp.raw["cf_indptr"]
                own_faces = np.unique(lp.raw["cf_indices"][: cfp[lp.n_own]])
                count[lp.face_gid[own_faces]] += 1
                lrows = (nd * own_faces[:, None] + comp).ravel()
                grows = (nd * lp.face_gid[own_faces][:, None] + comp).ravel()
                ccol = (nd * lp.cell_gid[:, None] + comp).ravel()
                fcol = (nd * lp.face_gid[:, None] + comp).ravel()
                for name, which in _KEYS:
                    M = ctx.matrix_rows(which, lrows).tocoo()
                    cmap = fcol if name in ("bound_stress", "bound_displacement_face") else ccol
                    rr, cc, vv = acc[name]
                    rr.append(grows[M.row])
                    cc.append(cmap[M.col])
                    vv.append(M.data)
                ctx.mpsa_assemble(np.zeros(nd * lp.face_gid.size), None)
                S = ctx.matrix_rows(_lib.MAT_MECH_SYSTEM, np.arange(nd * lp.n_own)).tocoo()
                sysacc[0].append(ccol[S.row])
                sysacc[1].append(ccol[S.col])
                sysacc[2].append(S.data)
            finally:
                ctx.close()
        scale = np.repeat(1.0 / np.maximum(count, 1), nd)
        for name, _ in _KEYS:
            rr, cc, vv = (np.concatenate(x) if x else np.zeros(0) for x in acc[name])
            M = sps.coo_matrix((vv * scale[rr.astype(np.int64)], (rr, cc)), shape=(nd * nf, ncols[name])).tocsr()
            M.sum_duplicates()
            M.sort_indices()
            md[name] = M
        A = sps.coo_matrix((np.concatenate(sysacc[2]), (np.concatenate(sysacc[0]), np.concatenate(sysacc[1]))),
                           shape=(nd * nc, nd * nc)).tocsr()
        A.sum_duplicates()
        A.sort_indices()
        self._split[id(sd)] = (sd, A)
        self._contexts.pop(id(sd), None)
        pd["active_cells"] = np.arange(nc)
        pd["active_faces"] = np.arange(nf)

    def _split_system(self, sd, data: dict):
        """(A, b) of a grid discretized in pieces: A from the pieces' device-side div_nd @ stress, b from the merged
        bound_stress (one host SpMV, mpsa.py:486-529)."""
        import scipy.sparse as sps

        pd = data[PARAMETERS][self.keyword]
        md = data[DISCRETIZATION_MATRICES][self.keyword]
        nd = sd.dim
        div = sps.kron(sd.cell_faces.T.tocsr(), sps.identity(nd), format="csr")
        b = -(div @ (md["bound_stress"] @ np.asarray(pd["bc_values"], dtype=float)))
        src = pd.get("source", None)
        if src is not None:
            b = b + np.asarray(src, dtype=float)
        return self._split[id(sd)][1], b

    def update_discretization(self, sd, data: dict) -> None:
        """Rediscretize around ``data["update_discretization"]["modified_cells" / "modified_faces"]``,
        keep every other row (mpsa.py:418-487 via _fvutils.partial_update_discretization)."""
        info = data.get("update_discretization", {})
        pd = data[PARAMETERS][self.keyword]
        cells = np.asarray(info.get("modified_cells", []), dtype=int)
        faces = np.asarray(info.get("modified_faces", []), dtype=int)
        ent = self._contexts.get(id(sd))
        remapped = "map_cells" in info or "map_faces" in info
        if remapped or ent is None or ent[0] is not sd or not ent[1].has_mpsa_discretization:
            saved = {k: pd.pop(k, None) for k in ("specified_cells", "specified_faces", "specified_nodes")}
            if ent is not None and remapped:
                self._contexts.pop(id(sd), None)  # the grid itself changed: upload it again
            try:
                self.discretize(sd, data)
            finally:
                pd.update({k: v for k, v in saved.items() if v is not None})
            return
        if cells.size == 0 and faces.size == 0:
            return
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

    def _assemble(self, sd, data):
        pd = data[PARAMETERS][self.keyword]
        ent = self._contexts.get(id(sd))
        if ent is None or ent[0] is not sd:
            raise RuntimeError("discretize(sd, data) must run on this object first")
        ctx = ent[1]
        src = pd.get("source", None)
        ctx.mpsa_assemble(np.asarray(pd["bc_values"], dtype=float), src)
        return ctx

    def assemble_matrix_rhs(self, sd, data: dict):
        """A = div_nd @ stress, b = -div_nd @ bound_stress @ bc_values + source (mpsa.py:486-529)."""
        sp = self._split.get(id(sd))
        if sp is not None and sp[0] is sd:
            return self._split_system(sd, data)
        ctx = self._assemble(sd, data)
        n = sd.dim * sd.num_cells
        return ctx.matrix(_lib.MAT_MECH_SYSTEM), ctx.active_rhs(n)

    def solve(self, sd, data: dict, method: str = "bicgstab", rtol: float = 1e-12, maxit: int = 50000, x0=None,
              restart: int = 0, precond: str = "jacobi"):
        sp = self._split.get(id(sd))
        if sp is not None and sp[0] is sd:
            from .solvers import solve_csr

            A, b = self._split_system(sd, data)
            if getattr(self, "_split_ctx", None) is None:
                self._split_ctx = _lib.Context(self.device, self._library)  # kept between solves
            return solve_csr(A, b, method=method, rtol=rtol, maxit=maxit, restart=restart, precond=precond, x0=x0,
                             context=self._split_ctx)
        if getattr(self, "_pieces_without_system", {}).get(id(sd)) is sd:
            raise NotImplementedError(
                "this grid was discretized in pieces with the Biot coupling terms (partition_arguments / memory "
                "bound): the mechanics system is not kept on the device -- assemble the coupled system from "
                "data[DISCRETIZATION_MATRICES] and hand it to porepy_amd.solve_csr")
        ctx = self._assemble(sd, data)
        return ctx.solve(method=method, rtol=rtol, maxit=maxit, x0=x0, n=sd.dim * sd.num_cells, restart=restart,
                         precond=precond)


def as_porepy_mpsa(device: int = 0, library=None):
    """Subclass of the reference's ``pp.Mpsa`` whose hot path runs on the MI355X; rebind with
    ``pp.Mpsa = porepy_amd.as_porepy_mpsa()`` before the model is built (``pp.ad.MpsaAd`` resolves
    ``pp.Mpsa`` at call time, numerics/ad/discretizations.py:134-150)."""
    import porepy as pp  # the reference; absent on the GPU box

    _device, _library = device, library
    _Ref = pp.Mpsa

    class HipMpsa(_Ref):  # type: ignore[misc]
        def __init__(self, keyword: str):
            _Ref.__init__(self, keyword)
            self._hip = Mpsa(keyword, _device, _library)

        def discretize(self, sd, data):
            if sd.dim < 2:
                return _Ref.discretize(self, sd, data)
            return self._hip.discretize(sd, data)

        def update_discretization(self, sd, data):
            if sd.dim < 2:
                return _Ref.update_discretization(self, sd, data)
            return self._hip.update_discretization(sd, data)

        def assemble_matrix_rhs(self, sd, data):
            if sd.dim < 2:
                return _Ref.assemble_matrix_rhs(self, sd, data)
            return self._hip.assemble_matrix_rhs(sd, data)

    return HipMpsa
