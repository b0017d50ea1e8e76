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
from .mpfa import determine_eta
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
        if eta is None:
            eta = determine_eta(sd)
        ctx = self.context(sd)
        is_rob = getattr(bnd, "is_rob", None)
        ctx.mpsa_set_params(np.asarray(C.values), sd.cell_volumes, bnd.is_dir, bnd.is_neu, float(eta), is_rob=is_rob,
                            robin_weight=getattr(bnd, "robin_weight", None) if is_rob is not None else None,
                            basis=basis)
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
        for name, which in _MECH_KEYS:
            md[name] = ctx.matrix(which)
        for name, term in _TERMS:
            md[name] = {k: ctx.biot_matrix(term, i) for i, k in enumerate(keys)}
        pd["active_cells"] = active_cells
        pd["active_faces"] = active_faces

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
