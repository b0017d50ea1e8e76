"""Run inside a subprocess with the REFERENCE PorePy importable (oracle/shim + /root/reference/src): a
MIXED-DIMENSIONAL single-phase flow model of the reference -- a 3-D Cartesian box cut by two intersecting
planar fractures (2-D subdomains, their 1-D intersection line, mortar grids between them: the structured
stand-in for BASELINE configs[4], whose 52-fracture geometry needs gmsh) -- run twice: untouched, and with
``pp.Mpfa`` rebound to the porepy_amd operator (host-emulation library: no GPU in this container) and the
coupled Jacobian solved by the HIP Krylov solver.  The reference discretizes one subdomain at a time
(numerics/ad/ad_utils.py:288-308) and couples through mortar fluxes (models/constitutive_laws.py:987-1000):
every subdomain of dimension >= 2 goes through the device path (fracture faces are internal Neumann faces
there, mpfa.py:1452-1454; the 2-D fracture grids are embedded in 3-D), the 1-D line through Tpfa."""
import json

import numpy as np

import porepy as pp
from porepy.applications.md_grids.domains import nd_cube_domain
from porepy.models.fluid_mass_balance import SinglePhaseFlow

import porepy_amd as pa
from tests import _parity as P


class Geometry:
    def set_domain(self):
        self._domain = nd_cube_domain(3, 1.0)

    def set_fractures(self):
        f1 = pp.PlaneFracture(np.array([[0.5, 0.5, 0.5, 0.5], [0.0, 1.0, 1.0, 0.0], [0.0, 0.0, 1.0, 1.0]]))
        f2 = pp.PlaneFracture(np.array([[0.0, 1.0, 1.0, 0.0], [0.5, 0.5, 0.5, 0.5], [0.0, 0.0, 1.0, 1.0]]))
        self._fractures = [f1, f2]

    def grid_type(self):
        return "cartesian"

    def meshing_arguments(self):
        return {"cell_size": float(__import__("os").environ.get("MD_CELL_SIZE", "0.25"))}


class BCs:
    def bc_type_darcy_flux(self, sd):
        sides = self.domain_boundary_sides(sd)
        return pp.BoundaryCondition(sd, sides.west + sides.east, "dir")

    def bc_values_pressure(self, bg):
        sides = self.domain_boundary_sides(bg)
        v = np.zeros(bg.num_cells)
        v[sides.west] = 3.0
        v[sides.east] = 1.0
        return v


class Permeability:
    def permeability(self, subdomains):
        # fractures 100 x more permeable than the matrix (isotropic)
        vals = np.concatenate([np.full(sd.num_cells, 1.0 if sd.dim == 3 else 100.0) for sd in subdomains]) if subdomains else np.zeros(0)
        return self.isotropic_second_order_tensor(subdomains, pp.wrap_as_dense_ad_array(vals, name="k"))


class Model(Geometry, BCs, Permeability, SinglePhaseFlow):
    pass


class HipSolveModel(pa.HipLinearSolver, Model):
    hip_library = P.dropin_library()


def run(cls=Model, linear_solver="scipy_sparse", opts=None):
    params = {"times_to_export": [], "linear_solver": linear_solver, "darcy_flux_discretization": "mpfa"}
    if opts:
        params["hip_solver_options"] = opts
    m = cls(params)
    pp.run_time_dependent_model(m, params)
    x = m.equation_system.get_variable_values(time_step_index=0)
    A, b = m.linear_system
    dims = sorted({sd.dim for sd in m.mdg.subdomains()}, reverse=True)
    return {"x": np.asarray(x), "A": A.copy(), "b": np.asarray(b).copy(), "dims": dims,
            "n_sub": len(m.mdg.subdomains()), "n_intf": len(m.mdg.interfaces()),
            "cells": {int(d): int(sum(sd.num_cells for sd in m.mdg.subdomains(dim=d))) for d in dims},
            "mortar_cells": int(sum(i.num_cells for i in m.mdg.interfaces())),
            "info": getattr(m, "hip_solver_info", None)}


ref = run()
calls = {}
HipMpfa = pa.as_porepy_discretization(library=P.dropin_library())
orig = HipMpfa.discretize


def counting(self, sd, data):
    calls[sd.dim] = calls.get(sd.dim, 0) + 1
    return orig(self, sd, data)


HipMpfa.discretize = counting
pp.Mpfa = HipMpfa
ours = run()
both = run(HipSolveModel, "hip_bicgstab", {"rtol": 1e-13, "precond": "jacobi", "maxit": 20000})
nrm = np.linalg.norm(ref["x"])
out = {
    "dims": ref["dims"], "subdomains": ref["n_sub"], "interfaces": ref["n_intf"], "cells": ref["cells"],
    "mortar_cells": ref["mortar_cells"], "dofs": int(ref["x"].size),
    "device_calls_by_dim": {str(k): v for k, v in calls.items()},
    "x_rel_err": float(np.linalg.norm(ours["x"] - ref["x"]) / nrm),
    "A_rel_err": float(abs(ours["A"] - ref["A"]).max() / abs(ref["A"]).max()),
    "x_rel_err_hip_solver": float(np.linalg.norm(both["x"] - ref["x"]) / nrm),
    "hip_solver_iterations": int(both["info"]["iterations"]),
    "x_norm": float(nrm),
}
# the coupled Jacobian of the reference as a fixture for the sharded-solve test (owner = subdomain blocks)
if "--save" in __import__("sys").argv:
    import os
    import scipy.sparse as sps

    A = sps.csr_matrix(ref["A"])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "md_jacobian_box_2fractures.npz")
    np.savez_compressed(path, data=A.data, indices=A.indices, indptr=A.indptr, shape=np.array(A.shape), b=ref["b"], x=ref["x"])
out["library"] = str(P.dropin_library()._name)
print("RESULT " + json.dumps(out))
