"""Run inside a subprocess with the REFERENCE PorePy importable (oracle/shim + /root/reference/src):
a stock SinglePhaseFlow model (BASELINE config C1: 50x50 Cartesian, Mpfa) run twice — untouched,
and with ``pp.Mpfa`` rebound to the porepy_amd operator (host-emulation library, because this
container has no GPU) — must produce the same pressure field and the same Jacobian."""
import json
import sys

import numpy as np

import porepy as pp
from porepy.applications.md_grids.domains import nd_cube_domain
from porepy.models.fluid_mass_balance import SinglePhaseFlow

import porepy_amd as pa
from tests import _parity as P


class Geometry:
    def set_domain(self):
        self._domain = nd_cube_domain(2, 1.0)

    def grid_type(self):
        return "cartesian"

    def meshing_arguments(self):
        return {"cell_size": 1.0 / 50}


class BCs:
    def bc_type_darcy_flux(self, sd):
        sides = self.domain_boundary_sides(sd)
        return pp.BoundaryCondition(sd, sides.west + sides.east, "dir")

    def bc_values_pressure(self, bg):
        sides = self.domain_boundary_sides(bg)
        v = np.zeros(bg.num_cells)
        v[sides.west] = 5.0
        v[sides.east] = 2.0
        return v


class Model(Geometry, BCs, SinglePhaseFlow):
    pass


class HipSolveModel(pa.HipLinearSolver, Model):
    hip_library = P.dropin_library()


def run(cls=Model, linear_solver="scipy_sparse"):
    params = {"times_to_export": [], "linear_solver": linear_solver,
              "darcy_flux_discretization": "mpfa"}
    m = cls(params)
    pp.run_time_dependent_model(m, params)
    sd = m.mdg.subdomains()[0]
    p = m.equation_system.get_variable_values([m.pressure_variable], time_step_index=0)
    A, b = m.linear_system
    return sd.num_cells, np.asarray(p), A.copy(), np.asarray(b).copy(), getattr(m, "hip_solver_info", None)


ref = run()
calls = {"n": 0}
HipMpfa = pa.as_porepy_discretization(library=P.dropin_library())
orig = HipMpfa.discretize


def counting(self, sd, data):
    calls["n"] += 1
    return orig(self, sd, data)


HipMpfa.discretize = counting
pp.Mpfa = HipMpfa
ours = run()
# ... and with the linear solve routed to the device Krylov solver as well
both = run(HipSolveModel, "hip_bicgstab")
# ... and the whole Newton step in HBM: matrices kept on the device (lazy proxies), operator trees walked with device
# Jacobians and device matrix leaves (DeviceAssembly), the device Jacobian handed to the device solver
class AllOnDevice(pa.DeviceAssembly, pa.HipLinearSolver, Model):
    hip_library = P.dropin_library()


pp.Mpfa = pa.as_porepy_discretization(library=P.dropin_library(), lazy=True)
alldev = run_alldev = None
try:
    params = {"times_to_export": [], "linear_solver": "hip_bicgstab", "darcy_flux_discretization": "mpfa",
              "hip_solver_options": {"rtol": 1e-13}}
    m = AllOnDevice(params)
    # (the leaves stay device matrices for everything the model evaluates -- the flux post-processing, the upwind
    # directions --, not only inside the assembly)
    with pa.ad.device_matrix_leaves(pa.Context(0, P.dropin_library())):
        pp.run_time_dependent_model(m, params)
    p_dev = np.asarray(m.equation_system.get_variable_values([m.pressure_variable], time_step_index=0))
    J_dev = m.linear_system[0]
    sd = m.mdg.subdomains()[0]
    md = m.mdg.subdomain_data(sd)[pp.DISCRETIZATION_MATRICES]
    kw = [k for k in md if "flux" in md[k]][0]
    alldev = {"p_rel_err": float(np.linalg.norm(p_dev - ref[1]) / np.linalg.norm(ref[1])),
              "jacobian_on_device": isinstance(J_dev, pa.DeviceCsr),
              "solved_from_device_jacobian": bool(m.hip_solver_info.get("device_jacobian")),
              "iterations": int(m.hip_solver_info["iterations"]),
              "flux_proxy": type(md[kw]["flux"]).__name__,
              "flux_fetched_to_host": bool(getattr(md[kw]["flux"], "materialized", True)),
              "A_rel_err": float(abs(J_dev.to_scipy() - ref[2]).max() / abs(ref[2]).max())}
except Exception as e:  # noqa: BLE001
    import traceback

    alldev = {"error": repr(e), "trace": traceback.format_exc()[-1500:]}
out = {
    "all_on_device": alldev,
    "p_rel_err_hip_solver": float(np.linalg.norm(both[1] - ref[1]) / np.linalg.norm(ref[1])),
    "hip_solver_iterations": int(both[4]["iterations"]),
    "cells": int(ref[0]),
    "calls_into_device_path": calls["n"],
    "p_rel_err": float(np.linalg.norm(ours[1] - ref[1]) / np.linalg.norm(ref[1])),
    "p_sum_ref": float(ref[1].sum()),
    "A_rel_err": float(abs(ours[2] - ref[2]).max() / abs(ref[2]).max()),
}
out["library"] = str(P.dropin_library()._name)
print("RESULT " + json.dumps(out))
