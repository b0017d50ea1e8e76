"""``Tpfa`` — two-point flux approximation on the device, with the operator API of the
reference's ``pp.Tpfa`` (numerics/fv/tpfa.py:24-279): same keys as ``Mpfa``, same stored
patterns.  It is what ``Mpfa`` hands 1-D grids to (mpfa.py:690-712)."""
from __future__ import annotations

import numpy as np
import scipy.sparse as sps

from . import _lib
from .grid import grid_to_raw
from .periodic import merge_periodic
from .params import DISCRETIZATION_MATRICES, PARAMETERS, bc_flags

_KEYS = (
    ("flux", _lib.MAT_FLUX),
    ("bound_flux", _lib.MAT_BOUND_FLUX),
    ("bound_pressure_cell", _lib.MAT_BOUND_PRESSURE_CELL),
    ("bound_pressure_face", _lib.MAT_BOUND_PRESSURE_FACE),
    ("vector_source", _lib.MAT_VECTOR_SOURCE),
    ("bound_pressure_vector_source", _lib.MAT_BOUND_PRESSURE_VECTOR_SOURCE),
)


def empty_matrices(sd, vector_source_dim: int) -> dict:
    """What the reference stores for a 0-D grid (tpfa.py:106-124, mpfa.py:129-149)."""
    nc, vd = sd.num_cells, max(int(vector_source_dim), 1)
    return {
        "flux": sps.csr_matrix((0, nc)),
        "bound_flux": sps.csr_matrix((0, 0)),
        "bound_pressure_cell": sps.csr_matrix((0, nc)),
        "bound_pressure_face": sps.csr_matrix((0, 0)),
        "vector_source": sps.csr_matrix((0, nc * vd)),
        "bound_pressure_vector_source": sps.csr_matrix((0, nc * vd)),
    }


class Tpfa:
    """TPFA flux discretization for ``keyword`` on the device."""

    def __init__(self, keyword: str, device: int = 0, library=None):
        self.keyword = keyword
        self.device = device
        self._library = library
        self.flux_matrix_key = "flux"
        self.bound_flux_matrix_key = "bound_flux"
        self.bound_pressure_cell_matrix_key = "bound_pressure_cell"
        self.bound_pressure_face_matrix_key = "bound_pressure_face"
        self.vector_source_matrix_key = "vector_source"
        self.bound_pressure_vector_source_matrix_key = "bound_pressure_vector_source"
        self._contexts: dict = {}
        self._periodic: dict = {}  # id(sd) -> PeriodicMerge or None

    def ndof(self, sd) -> int:
        return sd.num_cells

    def context(self, sd) -> _lib.Context:
        ent = self._contexts.get(id(sd))
        if ent is None or ent[0] is not sd:
            ctx = _lib.Context(self.device, self._library)
            raw = grid_to_raw(sd)
            merge = None
            if hasattr(sd, "periodic_face_map"):
                # the cells across a periodic pair become neighbours over the left face
                # (tpfa.py:114-262); see periodic.py
                merge = merge_periodic(raw, sd.periodic_face_map)
                raw = merge.raw
            self._periodic[id(sd)] = merge
            ctx.set_grid(raw)
            if merge is not None:
                ctx.set_periodic(merge.native, merge.shift)
            self._contexts[id(sd)] = (sd, ctx)
            return ctx
        return ent[1]

    def discretize(self, sd, data: dict) -> None:
        pd = data[PARAMETERS][self.keyword]
        md = data.setdefault(DISCRETIZATION_MATRICES, {}).setdefault(self.keyword, {})
        vdim = int(pd.get("ambient_dimension", sd.dim))
        if sd.dim == 0:
            md.update(empty_matrices(sd, vdim))
            return
        if data.get("Aavatsmark_transmissibilities", False):
            raise NotImplementedError("Aavatsmark_transmissibilities is not covered")
        if not 1 <= vdim <= 3:
            raise ValueError("ambient_dimension must be 1, 2 or 3")
        ctx = self.context(sd)
        bnd = pd["bc"]
        ctx.set_params(np.asarray(pd["second_order_tensor"].values, dtype=float), bc_flags(bnd), None, 0.0, None)
        ctx.tpfa_discretize(vdim)
        merge = self._periodic.get(id(sd))
        for name, which in _KEYS:
            md[name] = ctx.matrix(which) if merge is None else merge.copy_rows(ctx.matrix(which))

    def update_discretization(self, sd, data: dict) -> None:
        self.discretize(sd, data)

    def _ctx(self, sd):
        ent = self._contexts.get(id(sd))
        if ent is None or ent[0] is not sd:
            raise RuntimeError("discretize(sd, data) must run on this object first")
        return ent[1]

    def assemble_matrix_rhs(self, sd, data: dict):
        """fv_elliptic.py:67-112 on the device."""
        pd = data[PARAMETERS][self.keyword]
        if sd.dim == 0:
            return sps.csr_matrix((sd.num_cells, sd.num_cells)), np.zeros(sd.num_cells)
        ctx = self._ctx(sd)
        ctx.assemble(np.asarray(pd["bc_values"], dtype=float), pd.get("vector_source", None), None)
        return ctx.matrix(_lib.MAT_SYSTEM), ctx.rhs()

    def solve(self, sd, data: dict, source=None, method: str = "cg", rtol: float = 1e-12, maxit: int = 20000,
              x0=None, restart: int = 0, precond: str = "jacobi"):
        pd = data[PARAMETERS][self.keyword]
        ctx = self._ctx(sd)
        ctx.assemble(np.asarray(pd["bc_values"], dtype=float), pd.get("vector_source", None), source)
        return ctx.solve(method=method, rtol=rtol, maxit=maxit, x0=x0, restart=restart, precond=precond)
