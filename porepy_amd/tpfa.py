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


class DifferentiableTpfa:
    """Differentiable two-point transmissibilities on the device: the numerical core of the reference's
    ``AdTpfaFlux.__transmissibility_matrix`` (models/constitutive_laws.py:1504-1578), which composes
    ``t_f_full = 1 / (hf_to_f @ (1 / (d_n_by_dist @ k_c)))`` from the half-face matrices of
    ``DifferentiableTpfa`` (numerics/fv/tpfa.py:546-620) and lets the forward AD carry the Jacobian through
    two sparse products.  Here value and Jacobian come out of one pass over the half-faces
    (``pfv_tpfa_transmissibility_ad``).

    ``k_c`` is the reference's cell-wise tensor vector: 9 entries per cell, cell-major, ``K[r][s]`` at
    ``3 r + s``.  ``transmissibility(sd, k_c)`` returns ``(t_f, dt_f/dk_c)`` with the Jacobian as a
    ``(num_faces, 9 num_cells)`` CSR matrix; a permeability that depends on the primary variables is
    chained on the caller's side, ``jac_u = dt_dk @ k_c.jac`` - what the reference's AdArray arithmetic
    does (``mpfa``-based fluxes use the same matrix for their ``p_diff * d(T_TPFA)`` term,
    constitutive_laws.py:1580-1625)."""

    def __init__(self, device: int = 0, library=None):
        self.device = device
        self._library = library
        self._contexts: dict = {}
        self.calls = 0  # evaluations through the device kernel

    def context(self, sd) -> _lib.Context:
        ent = self._contexts.get(id(sd))
        if ent is None or ent[0] is not sd:
            ctx = _lib.Context(self.device, self._library)
            raw = grid_to_raw(sd)
            ctx.set_grid(raw)
            self._contexts[id(sd)] = (sd, ctx, raw)
            return ctx
        return ent[1]

    def half_face_cells(self, sd) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
        """(face, cell, sign) of every half-face in the library's order (the entries of ``cell_faces``, cell
        by cell).  The reference numbers its half-faces face by face (``sps.find(sd.cell_faces)``):
        ``np.lexsort((cell, face))`` of these arrays is the permutation into its order."""
        self.context(sd)
        raw = self._contexts[id(sd)][2]
        cells = np.repeat(np.arange(raw["cf_indptr"].size - 1), np.diff(raw["cf_indptr"]))
        return raw["cf_indices"].astype(np.int64), cells, raw["cf_sign"].astype(np.int64)

    def transmissibility(self, sd, k_c) -> tuple[np.ndarray, sps.csr_matrix]:
        nc = sd.num_cells
        k = np.asarray(k_c, dtype=np.float64)
        if k.shape == (3, 3, nc):
            K = k
        elif k.shape == (9 * nc,):
            K = np.ascontiguousarray(k.reshape(nc, 9).T).reshape(3, 3, nc)
        else:
            raise ValueError("k_c must be the 9 * num_cells vector of the reference or a (3, 3, num_cells) array")
        ctx = self.context(sd)
        t, d = ctx.tpfa_transmissibility_ad(K)
        self.calls += 1
        fi, ci, _ = self.half_face_cells(sd)
        rows = np.repeat(fi, 9)
        cols = (9 * ci[:, None] + np.arange(9)[None, :]).ravel()
        jac = sps.csr_matrix((d.ravel(), (rows, cols)), shape=(sd.num_faces, 9 * nc))
        return t, jac


def as_porepy_ad_tpfa_flux(device: int = 0, library=None):
    """Mixin for models of the reference that use its differentiable two-point flux (``pp.constitutive_laws.
    AdTpfaFlux`` through ``DarcysLawAd`` / ``FouriersLawAd``): put the returned class BEFORE those in the
    bases and ``AdTpfaFlux.__transmissibility_matrix`` (models/constitutive_laws.py:1504-1578) evaluates the
    face transmissibilities and their Jacobian with the device kernel instead of two sparse products through
    the forward AD; everything built on ``t_f_full`` upstream (``diffusive_flux``, ``potential_trace``, the
    MPFA / TPFA mixture) is untouched.

        class Model(porepy_amd.as_porepy_ad_tpfa_flux(), pp.constitutive_laws.DarcysLawAd, pp.SinglePhaseFlow): ...
    """
    import porepy as pp  # the reference; absent on the GPU box

    hip = DifferentiableTpfa(device, library)

    class HipAdTpfaFlux:
        hip_differentiable_tpfa = hip

        # the name the reference's ``self.__transmissibility_matrix(...)`` resolves to inside AdTpfaFlux
        def _AdTpfaFlux__transmissibility_matrix(self, subdomains, diffusivity_tensor):
            basis = self.basis(subdomains, dim=9)
            volumes = pp.ad.sum_operator_list([e @ self.specific_volume(subdomains) for e in basis])
            k_c = volumes * diffusivity_tensor(subdomains)
            diff_discr = pp.numerics.fv.tpfa.DifferentiableTpfa()
            _, d_vec, _ = diff_discr.half_face_geometry_matrices(subdomains)
            hf_to_f = diff_discr.half_face_map(subdomains, to_entity="faces", with_sign=True)
            sds = list(subdomains)

            def transmissibility(k):
                kv = np.asarray(getattr(k, "val", k), dtype=np.float64)
                vals, jacs, o = [], [], 0
                for sd in sds:
                    n9 = 9 * sd.num_cells
                    if sd.num_cells == 0 or sd.num_faces == 0:
                        vals.append(np.zeros(sd.num_faces))
                        jacs.append(sps.csr_matrix((sd.num_faces, n9)))
                    else:
                        v, j = hip.transmissibility(sd, kv[o:o + n9])
                        vals.append(v)
                        jacs.append(j)
                    o += n9
                val = np.concatenate(vals) if vals else np.zeros(0)
                if not hasattr(k, "jac"):
                    return val
                dt_dk = sps.block_diag(jacs, format="csr") if jacs else sps.csr_matrix((0, 0))
                return pp.ad.AdArray(val, dt_dk @ k.jac)

            t_f_full = pp.ad.Function(transmissibility, "hip_tpfa_transmissibility")(k_c)
            t_f_full.set_name("transmissibility matrix")
            return t_f_full, diff_discr, hf_to_f, d_vec

    return HipAdTpfaFlux
