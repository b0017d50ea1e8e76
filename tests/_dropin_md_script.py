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

import porepy_amd as pa
from tests import _parity as P


from tests._dropin_md_script_model import Model


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


# the whole Newton step on the device for the mixed-dimensional model as well: device-resident matrices of the 3-D and
# 2-D grids (lazy proxies), host matrices of the 1-D grid and of the interface laws uploaded into the same block-diagonal
# leaves, operator trees with device Jacobians, the coupled device Jacobian solved by the device Krylov solver
class AllOnDevice(pa.DeviceAssembly, pa.HipLinearSolver, Model):
    hip_library = P.dropin_library()


pp.Mpfa = pa.as_porepy_discretization(library=P.dropin_library(), lazy=True)
try:
    with pa.ad.device_matrix_leaves(pa.Context(0, P.dropin_library())):
        alld = run(AllOnDevice, "hip_bicgstab", {"rtol": 1e-13, "precond": "jacobi", "maxit": 20000})
    all_on_device = {"x_rel_err": float(np.linalg.norm(alld["x"] - ref["x"]) / nrm),
                     "jacobian_on_device": isinstance(alld["A"], pa.DeviceCsr),
                     "solved_from_device_jacobian": bool(alld["info"].get("device_jacobian")),
                     "iterations": int(alld["info"]["iterations"]),
                     "A_rel_err": float(abs(alld["A"].to_scipy() - ref["A"]).max() / abs(ref["A"]).max())}
except Exception as e:  # noqa: BLE001
    import traceback

    all_on_device = {"error": repr(e), "trace": traceback.format_exc()[-1500:]}
out = {
    "all_on_device": all_on_device,
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
