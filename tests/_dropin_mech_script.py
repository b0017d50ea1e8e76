"""Run inside a subprocess with the REFERENCE PorePy importable: stock MomentumBalance and
Poromechanics models (Cartesian 2-D grid) run untouched and with ``pp.Mpsa`` / ``pp.Biot`` /
``pp.Mpfa`` rebound to the porepy_amd operators (host-emulation library: no GPU in this container);
displacement, pressure and Jacobian must agree."""
import json

import numpy as np

import porepy as pp
from porepy.applications.md_grids.domains import nd_cube_domain
from porepy.models.momentum_balance import MomentumBalance
from porepy.models.poromechanics import Poromechanics

import porepy_amd as pa
from tests import _parity as P


class Geometry:
    def set_domain(self):
        self._domain = nd_cube_domain(2, 1.0)

    def grid_type(self):
        return "cartesian"

    def meshing_arguments(self):
        return {"cell_size": 1.0 / 6}


class MechBCs:
    def bc_type_mechanics(self, sd):
        sides = self.domain_boundary_sides(sd)
        bc = pp.BoundaryConditionVectorial(sd, sides.south + sides.north, "dir")
        bc.internal_to_dirichlet(sd)
        return bc

    def bc_values_displacement(self, bg):
        sides = self.domain_boundary_sides(bg)
        v = np.zeros((self.nd, bg.num_cells))
        v[1, sides.north] = -0.01
        v[0, sides.north] = 0.002
        return v.ravel("F")


class FlowBCs:
    def bc_type_darcy_flux(self, sd):
        sides = self.domain_boundary_sides(sd)
        return pp.BoundaryCondition(sd, sides.west + sides.east, "dir")

    def bc_values_pressure(self, bg):
        sides = self.domain_boundary_sides(bg)
        v = np.zeros(bg.num_cells)
        v[sides.west] = 1.0
        return v


class Mech(Geometry, MechBCs, MomentumBalance):
    pass


class Poro(Geometry, MechBCs, FlowBCs, Poromechanics):
    pass


def run(cls):
    params = {"times_to_export": [], "linear_solver": "scipy_sparse", "darcy_flux_discretization": "mpfa"}
    m = cls(params)
    pp.run_time_dependent_model(m, params)
    x = m.equation_system.get_variable_values(time_step_index=0)
    A, b = m.linear_system
    return np.asarray(x), A.copy()


ref_mech = run(Mech)
ref_poro = run(Poro)
lib = P.dropin_library()
calls = {"mpsa": 0, "biot": 0}
HipMpsa, HipBiot, HipMpfa = pa.as_porepy_mpsa(library=lib), pa.as_porepy_biot(library=lib), pa.as_porepy_discretization(library=lib)
for cls_, key in ((HipMpsa, "mpsa"), (HipBiot, "biot")):
    orig = cls_.discretize

    def counting(self, sd, data, _o=orig, _k=key):
        calls[_k] += 1
        return _o(self, sd, data)

    cls_.discretize = counting
pp.Mpsa, pp.Biot, pp.Mpfa = HipMpsa, HipBiot, HipMpfa
our_mech = run(Mech)
our_poro = run(Poro)
out = {
    "calls": calls,
    "mech_dofs": int(ref_mech[0].size),
    "mech_x_rel_err": float(np.linalg.norm(our_mech[0] - ref_mech[0]) / np.linalg.norm(ref_mech[0])),
    "mech_A_rel_err": float(abs(our_mech[1] - ref_mech[1]).max() / abs(ref_mech[1]).max()),
    "poro_dofs": int(ref_poro[0].size),
    "poro_x_rel_err": float(np.linalg.norm(our_poro[0] - ref_poro[0]) / np.linalg.norm(ref_poro[0])),
    "poro_A_rel_err": float(abs(our_poro[1] - ref_poro[1]).max() / abs(ref_poro[1]).max()),
}
out["library"] = str(P.dropin_library()._name)
print("RESULT " + json.dumps(out))
