"""Run inside a subprocess with the REFERENCE PorePy importable: SURVEY §8 row N4, the operator tree itself.
The reference's mixed-dimensional single-phase flow model (compressible fluid: the accumulation term, the upwinded
mobility and the density function make the residual nonlinear) is advanced a few Newton iterations; at that state
``EquationSystem.assemble()`` (numerics/ad/equation_system.py:1579) is compared with
``porepy_amd.ad.assemble_on_device``: the same operator trees walked by the reference's own parser, every Jacobian
formed on the device as a ``DeviceCsr``.  Also the thermo-hydro model (two coupled fields, three mortar variables)."""
import json

import numpy as np
import scipy.sparse as sps

import porepy as pp

import porepy_amd as pa
from tests import _parity as P
from tests._dropin_md_script_model import Model


def compare(m, ctx):
    es = m.equation_system
    A, b = es.assemble()
    A = sps.csr_matrix(A)
    A.sort_indices()
    J, bd = pa.ad.assemble_on_device(es, ctx)
    assert isinstance(J, pa.DeviceCsr)
    Jh = J.to_scipy()
    Jh.sort_indices()
    # scipy keeps explicit zeros in places; compare the matrices as operators and their stored structure
    same_bits = (Jh.shape == A.shape and np.array_equal(Jh.indptr, A.indptr) and np.array_equal(Jh.indices, A.indices)
                 and np.array_equal(Jh.data, A.data))
    diff = abs(Jh - A)
    scale = abs(A).max()
    Az = A.copy()
    Az.eliminate_zeros()
    Jz = Jh.copy()
    Jz.eliminate_zeros()
    return {"dofs": int(A.shape[0]), "nnz_reference": int(A.nnz), "nnz_device": int(Jh.nnz),
            "bit_identical": bool(same_bits),
            "same_nonzero_pattern": bool(np.array_equal(Az.indptr, Jz.indptr) and np.array_equal(Az.indices, Jz.indices)),
            "jac_rel_err": float(diff.max() / scale) if diff.nnz else 0.0,
            "rhs_identical": bool(np.array_equal(np.asarray(b), bd)),
            "equations": len(es.equations)}


out = {}
lib = P.dropin_library()
ctx = pa.Context(0, lib)

# --- mixed-dimensional compressible single-phase flow, two time steps in
fluid = pp.FluidComponent(compressibility=0.05, viscosity=1.0, density=1.0)
solid = pp.SolidConstants(permeability=1.0, porosity=0.2, normal_permeability=2.0, residual_aperture=0.1)
params = {"times_to_export": [], "linear_solver": "scipy_sparse", "darcy_flux_discretization": "mpfa",
          "material_constants": {"fluid": fluid, "solid": solid},
          "time_manager": pp.TimeManager(schedule=[0.0, 0.4], dt_init=0.2, constant_dt=True)}
m = Model(params)
pp.run_time_dependent_model(m, params)
# a state off the converged one, so that the residual and every nonlinear term are non-trivial
x = m.equation_system.get_variable_values(iterate_index=0)
rng = np.random.default_rng(0)
m.equation_system.set_variable_values(x * (1.0 + 0.05 * rng.random(x.size)) + 0.01 * rng.random(x.size), iterate_index=0)
out["md_flow"] = compare(m, ctx)
out["md_flow"]["dims"] = sorted({sd.dim for sd in m.mdg.subdomains()}, reverse=True)
# ... and with the discretization matrices entering the trees as DeviceCsr as well (no host copy of a leaf in any product)
A0, b0 = m.equation_system.assemble()
J1, b1 = pa.ad.assemble_on_device(m.equation_system, ctx, device_leaves=True)
A0 = sps.csr_matrix(A0)
J1h = J1.to_scipy()
out["md_flow_device_leaves"] = {
    "jac_rel_err": float(abs(J1h - A0).max() / abs(A0).max()),
    "rhs_rel_err": float(np.linalg.norm(np.asarray(b0) - b1) / np.linalg.norm(b0)),
    "same_shape": bool(J1h.shape == A0.shape)}

# --- the device Jacobian goes to the device solver without a host copy: Newton increment vs scipy's direct solve
A, b = m.equation_system.assemble()
J, bd = pa.ad.assemble_on_device(m.equation_system, ctx)
dx_ref = __import__("scipy.sparse.linalg", fromlist=["spsolve"]).spsolve(sps.csc_matrix(A), b)
sysctx = J.as_system(bd)
dx, info = sysctx.solve(method="bicgstab", rtol=1e-13, maxit=5000, n=J.shape[0], precond="jacobi")
out["newton_increment_rel_err"] = float(np.linalg.norm(dx - dx_ref) / np.linalg.norm(dx_ref))
out["newton_increment_iterations"] = int(info["iterations"])

# --- thermo-hydro on the same grid (tests/_dropin_thermal_script.py's model)
try:
    import importlib.util
    import os
    import sys

    spec = importlib.util.spec_from_file_location("_thermal_model", os.path.join(os.path.dirname(__file__), "_dropin_thermal_script.py"))
    src = open(spec.origin).read().split("ref = run()")[0]
    ns = {"__name__": "_thermal_model", "__file__": spec.origin}
    exec(compile(src, spec.origin, "exec"), ns)
    solid = pp.SolidConstants(permeability=0.5, thermal_conductivity=2.0, porosity=0.2, specific_heat_capacity=1.5,
                              normal_permeability=5.0, residual_aperture=1e-1)
    fluid = pp.FluidComponent(thermal_conductivity=0.6, specific_heat_capacity=2.0, compressibility=1e-2,
                              thermal_expansion=1e-3, viscosity=1.0)
    tparams = {"times_to_export": [], "linear_solver": "scipy_sparse", "darcy_flux_discretization": "mpfa",
               "fourier_flux_discretization": "mpfa", "material_constants": {"solid": solid, "fluid": fluid},
               "time_manager": pp.TimeManager(schedule=[0.0, 0.1], dt_init=0.1, constant_dt=True),
               "max_iterations": 20, "nl_convergence_tol": 1e-10, "nl_convergence_tol_res": 1e-10}
    tm = ns["Model"](tparams)
    pp.run_time_dependent_model(tm, tparams)
    x = tm.equation_system.get_variable_values(iterate_index=0)
    tm.equation_system.set_variable_values(x * (1.0 + 0.02 * rng.random(x.size)), iterate_index=0)
    out["thermo_hydro"] = compare(tm, ctx)
except Exception as e:  # noqa: BLE001
    out["thermo_hydro"] = {"error": repr(e)}
out["library"] = str(lib._name)
print("RESULT " + json.dumps(out))
