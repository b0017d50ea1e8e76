"""Golden fixtures for the differentiable MPFA flux: value and Jacobian of ``darcy_flux`` and of the mass
balance residual, evaluated by the REFERENCE's own operator tree and forward AD (AdTpfaFlux with an Mpfa base
discretization, models/constitutive_laws.py:1195-1336, 1580-1721) on
  * the reference's own unit-test model (tests/numerics/fv/test_tpfa.py: UnitTestAdTpfaFlux -- two cells,
    full-tensor permeability that depends on the pressure, one Neumann and one non-zero Dirichlet face,
    vector source), and
  * a subclass of it on a perturbed 3-D tetrahedral grid with K(p) = K0 (1 + p^2), mixed conditions.

TEST INFRASTRUCTURE; build container only:

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 \
      PYTHONPATH=/root/repo/oracle/shim:/root/reference/src:/root/repo \
      python /root/repo/oracle/gen_golden_ad_flux.py
"""
from __future__ import annotations

import importlib.util
import os
import sys

import numpy as np
import scipy.sparse as sps

import porepy as pp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.gen_golden import OUT, pack_csr, perturb_interior  # noqa: E402
from oracle.ref_bridge import grid_to_raw  # noqa: E402

_spec = importlib.util.spec_from_file_location("reference_test_tpfa", "/root/reference/tests/numerics/fv/test_tpfa.py")
_ref = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_ref)
UnitTestAdTpfaFlux = _ref.UnitTestAdTpfaFlux


class TetAdFlux(UnitTestAdTpfaFlux):
    """The same model on a perturbed tetrahedral grid: K(p) = K0 (1 + p^2) cell-wise, Dirichlet on the x-sides
    (value 1 + y), Neumann elsewhere (small fluxes), a vector source."""

    def set_domain(self):
        self._domain = pp.Domain({"xmin": 0, "xmax": 1, "ymin": 0, "ymax": 1, "zmin": 0, "zmax": 1})

    def set_geometry(self):
        self.set_domain()
        self.set_fractures()
        self.fracture_network = pp.create_fracture_network(self.fractures, self.domain)
        rng = np.random.default_rng(77)
        g = perturb_interior(pp.StructuredTetrahedralGrid([2, 2, 2], [1, 1, 1]), rng, 0.08)
        mdg = pp.MixedDimensionalGrid()
        mdg.add_subdomains([g])
        mdg.set_boundary_grid_projections()
        self.mdg = mdg
        self.nd = 3
        self.set_well_network()
        nc = g.num_cells
        B = rng.random((3, 3, nc)) - 0.5
        self._K0 = np.einsum("ikn,jkn->ijn", B, B) + 0.5 * np.eye(3)[:, :, None]
        self._p0 = 0.5 + rng.random(nc)
        self._neu_vals = 0.01 * rng.random(g.num_faces)

    def ic_values_pressure(self, sd):
        return self._p0.copy()

    def permeability(self, subdomains):
        if len(subdomains) == 0:
            return pp.wrap_as_dense_ad_array(0, size=0)
        sd = subdomains[0]
        nc = sd.num_cells
        k0 = np.ascontiguousarray(self._K0.reshape(9, nc).T).ravel()     # cell-major, 3 r + s
        rep = sps.csr_matrix((np.ones(9 * nc), (np.arange(9 * nc), np.repeat(np.arange(nc), 9))), shape=(9 * nc, nc))
        p = self.pressure(subdomains)
        one_plus_p2 = pp.ad.SparseArray(rep) @ (p ** 2) + pp.wrap_as_dense_ad_array(np.ones(9 * nc))
        return pp.wrap_as_dense_ad_array(k0, name="K0") * one_plus_p2

    def _dir_faces(self, sd):
        bf = sd.get_all_boundary_faces()
        x = sd.face_centers[0, bf]
        return bf, (x < 1e-9) | (x > 1 - 1e-9)

    def bc_type_darcy_flux(self, sd):
        bf, isdir = self._dir_faces(sd)
        return pp.BoundaryCondition(sd, bf, ["dir" if d else "neu" for d in isdir])

    def bc_values_darcy_flux(self, bg):
        sd = bg.parent
        bf, isdir = self._dir_faces(sd)
        vals = np.zeros(sd.num_faces)
        vals[bf[~isdir]] = self._neu_vals[bf[~isdir]]
        return bg.projection() @ vals

    def bc_values_pressure(self, bg):
        sd = bg.parent
        bf, isdir = self._dir_faces(sd)
        vals = np.zeros(sd.num_faces)
        vals[bf[isdir]] = 1.0 + sd.face_centers[1, bf[isdir]]
        return bg.projection() @ vals


def save(name, model_cls, vs):
    model = model_cls({"darcy_flux_discretization": "mpfa", "vector_source": vs, "times_to_export": []})
    model.prepare_simulation()
    sds = model.mdg.subdomains()
    sd = sds[0]
    model.discretize()
    es = model.equation_system
    flux = model.darcy_flux(sds).value_and_jacobian(es)
    p = model.pressure(sds).value(es)
    kc = model.permeability(sds).value_and_jacobian(es)
    nc = sd.num_cells
    K = np.ascontiguousarray(kc.val.reshape(nc, 9).T).reshape(3, 3, nc)
    dk = np.zeros((9, nc))
    J = sps.csr_matrix(kc.jac)
    for c in range(nc):
        for r in range(9):
            dk[r, c] = J[9 * c + r, c]
    assert abs(J).sum() - np.abs(dk).sum() < 1e-12 * max(1.0, abs(J).sum())   # K_c depends on p_c only
    data = model.mdg.subdomain_data(sd)
    bc = data[pp.PARAMETERS][model.darcy_keyword]["bc"]
    # boundary values the flux sees: Dirichlet pressures and Neumann fluxes, face-wise
    bvals = model.combine_boundary_operators_darcy_flux(sds).value(es)
    div = sd.cell_faces.T
    store = {}
    for k, v in grid_to_raw(sd).items():
        store["grid_" + k] = np.asarray(v)
    store["perm"], store["dk_dp"], store["p"] = K, dk.reshape(3, 3, nc), p
    store["bc_flags"] = (np.asarray(bc.is_dir) * 1 + np.asarray(bc.is_neu) * 2).astype(np.uint8)
    store["bc_values"] = np.asarray(bvals)
    store["vector_source"] = np.asarray(vs, dtype=float)
    store["ref_flux"] = np.asarray(flux.val)
    pack_csr("ref_flux_jac", sps.csr_matrix(flux.jac), store)
    pack_csr("ref_div_flux_jac", sps.csr_matrix(div @ flux.jac), store)
    store["ref_div_flux"] = np.asarray(div @ flux.val)
    md = data[pp.DISCRETIZATION_MATRICES][model.darcy_keyword]
    pack_csr("ref_mpfa_flux", sps.csr_matrix(md["flux"]), store)
    pack_csr("ref_mpfa_vector_source", sps.csr_matrix(md["vector_source"]), store)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **store)
    print(f"{name:28s} cells={nc:4d} faces={sd.num_faces:4d} |flux|max={np.abs(flux.val).max():.3e} "
          f"jac nnz={flux.jac.nnz} {os.path.getsize(path)/1024:.0f} KiB")


def main():
    save("adflux_unit_2cells", UnitTestAdTpfaFlux, np.array([1.0, 2.0, 3.0, 5.0]))
    save("adflux_unit_2cells_novs", UnitTestAdTpfaFlux, np.zeros(4))
    rng = np.random.default_rng(5)
    save("adflux_tet3d_2x2x2", TetAdFlux, rng.random(3 * 48))


if __name__ == "__main__":
    main()
