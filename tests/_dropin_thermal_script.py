"""Run inside a subprocess with the REFERENCE PorePy importable: the reference's coupled THERMO-HYDRO model
(models/mass_and_energy_balance.py:83 ``MassAndEnergyBalance``: Darcy flux and Fourier flux both discretized by the
same operator class, models/constitutive_laws.py:1078-1088 and :2425-2437, plus upwinded enthalpy flux) on the
MIXED-DIMENSIONAL stand-in for BASELINE configs[4] -- a 3-D box cut by two intersecting fractures (2-D subdomains,
1-D intersection, mortar grids) -- run twice: untouched, and with ``pp.Mpfa`` rebound to the porepy_amd operator, so
that both the hydraulic ("mobility" keyword) and the thermal ("fourier" keyword) discretizations of every subdomain
of dimension >= 2 go through the device path.  Two implicit time steps with Newton iterations; all unknowns
(pressures, temperatures, mortar Darcy / Fourier / enthalpy fluxes) and the last Jacobian must reproduce."""
import json
import os

import numpy as np

import porepy as pp
from porepy.applications.md_grids.domains import nd_cube_domain
from porepy.models.mass_and_energy_balance import MassAndEnergyBalance

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
        return {"cell_size": float(os.environ.get("MD_CELL_SIZE", "0.25"))}


class BCs:
    def bc_type_darcy_flux(self, sd):
        s = self.domain_boundary_sides(sd)
        return pp.BoundaryCondition(sd, s.west + s.east, "dir")

    def bc_values_pressure(self, bg):
        s = self.domain_boundary_sides(bg)
        v = np.zeros(bg.num_cells)
        v[s.west] = 2.0
        v[s.east] = 1.0
        return v

    def bc_type_fourier_flux(self, sd):
        s = self.domain_boundary_sides(sd)
        return pp.BoundaryCondition(sd, s.west + s.east, "dir")

    def bc_type_enthalpy_flux(self, sd):
        s = self.domain_boundary_sides(sd)
        return pp.BoundaryCondition(sd, s.west + s.east, "dir")

    def bc_values_temperature(self, bg):
        s = self.domain_boundary_sides(bg)
        v = np.full(bg.num_cells, 1.0)
        v[s.west] = 3.0
        return v


class Model(Geometry, BCs, MassAndEnergyBalance):
    pass


class HipSolveModel(pa.HipLinearSolver, Model):
    hip_library = P.dropin_library()


def run(cls=Model, linear_solver="scipy_sparse", opts=None):
    solid = pp.SolidConstants(permeability=0.5, thermal_conductivity=2.0, porosity=0.2, specific_heat_capacity=1.5,
                              normal_permeability=5.0, residual_aperture=1e-1)
    fluid = pp.FluidComponent(thermal_conductivity=0.6, specific_heat_capacity=2.0, compressibility=1e-2,
                              thermal_expansion=1e-3, viscosity=1.0)
    params = {"times_to_export": [], "linear_solver": linear_solver, "darcy_flux_discretization": "mpfa",
              "fourier_flux_discretization": "mpfa",
              "material_constants": {"solid": solid, "fluid": fluid},
              "time_manager": pp.TimeManager(schedule=[0.0, 0.2], dt_init=0.1, constant_dt=True),
              "max_iterations": 20, "nl_convergence_tol": 1e-10, "nl_convergence_tol_res": 1e-10}
    if opts is not None:
        params["hip_solver_options"] = opts
    m = cls(params)
    solves = []
    if cls is not Model:
        inner = m.solve_linear_system

        def recording():
            x = inner()
            solves.append(dict(m.hip_solver_info))
            return x

        m.solve_linear_system = recording
    pp.run_time_dependent_model(m, params)
    x = m.equation_system.get_variable_values(time_step_index=0)
    A, b = m.linear_system
    names = sorted({v.name for v in m.equation_system.variables})
    return {"x": np.asarray(x), "A": A.copy(), "names": names, "dims": sorted({sd.dim for sd in m.mdg.subdomains()}, reverse=True),
            "n_sub": len(m.mdg.subdomains()), "n_intf": len(m.mdg.interfaces()), "solves": solves,
            "blocks": m._hip_blocks({}) if cls is not Model else None,
            "T": np.asarray(m.equation_system.get_variable_values([m.temperature_variable], time_step_index=0)),
            "p": np.asarray(m.equation_system.get_variable_values([m.pressure_variable], time_step_index=0))}


ref = run()
calls = {}
HipMpfa = pa.as_porepy_discretization(library=P.dropin_library())
orig = HipMpfa.discretize


def counting(self, sd, data):
    key = f"{self.keyword}:{sd.dim}"
    calls[key] = calls.get(key, 0) + 1
    return orig(self, sd, data)


HipMpfa.discretize = counting
pp.Mpfa = HipMpfa
ours = run()
# ... and with the linear systems of every Newton iteration solved on the device as well: GMRES with the block
# lower-triangular preconditioner (rows matched to unknowns, one block per variable and subdomain / interface)
both = run(HipSolveModel, "hip_gmres", {"precond": "block", "rtol": 1e-13, "restart": 80})
nrm = np.linalg.norm(ref["x"])
out = {
    "x_rel_err_hip_solver": float(np.linalg.norm(both["x"] - ref["x"]) / nrm),
    "hip_linear_solves": len(both["solves"]),
    "hip_solver_max_iterations": int(max(s_["iterations"] for s_ in both["solves"])),
    "hip_solver_blocks": int(both["solves"][0]["blocks"]), "hip_rows_matched": bool(both["solves"][0]["rows_matched"]),
    "hip_solver_worst_true_residual": float(max(s_["true_rel_residual"] for s_ in both["solves"])),
    "variables": ref["names"], "dims": ref["dims"], "subdomains": ref["n_sub"], "interfaces": ref["n_intf"],
    "dofs": int(ref["x"].size), "device_calls": calls,
    "x_rel_err": float(np.linalg.norm(ours["x"] - ref["x"]) / nrm),
    "T_rel_err": float(np.linalg.norm(ours["T"] - ref["T"]) / np.linalg.norm(ref["T"])),
    "p_rel_err": float(np.linalg.norm(ours["p"] - ref["p"]) / np.linalg.norm(ref["p"])),
    "A_rel_err": float(abs(ours["A"] - ref["A"]).max() / abs(ref["A"]).max()),
    "T_range": [float(ref["T"].min()), float(ref["T"].max())], "p_range": [float(ref["p"].min()), float(ref["p"].max())],
}
if "--save" in __import__("sys").argv:
    import scipy.sparse as sps

    A = sps.csr_matrix(ref["A"])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "md_thermal_jacobian_box_2fractures.npz")
    # right-hand side of a known answer (the converged state itself: the last Newton residual is round-off)
    np.savez_compressed(path, data=A.data, indices=A.indices, indptr=A.indptr, shape=np.array(A.shape),
                        b=A @ ref["x"], x=ref["x"], block_of=both["blocks"][0], row_perm=both["blocks"][1])
# ... and with the Jacobian itself assembled on the device (DeviceAssembly: device Jacobians, device matrix leaves, lazy
# matrices); the block solver reads its block description from the device matrix
class AllOnDevice(pa.DeviceAssembly, pa.HipLinearSolver, Model):
    hip_library = P.dropin_library()


pp.Mpfa = pa.as_porepy_discretization(library=P.dropin_library(), lazy=True)
try:
    with pa.ad.device_matrix_leaves(pa.Context(0, P.dropin_library())):
        alld = run(AllOnDevice, "hip_gmres", {"precond": "block", "rtol": 1e-13, "restart": 80})
    out["all_on_device"] = {"x_rel_err": float(np.linalg.norm(alld["x"] - ref["x"]) / nrm),
                            "jacobian_on_device": isinstance(alld["A"], pa.DeviceCsr),
                            "max_iterations": int(max(s_["iterations"] for s_ in alld["solves"]))}
except Exception as e:  # noqa: BLE001
    out["all_on_device"] = {"error": repr(e)}
out["library"] = str(P.dropin_library()._name)
print("RESULT " + json.dumps(out))
