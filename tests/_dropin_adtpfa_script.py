"""Run inside a subprocess with the REFERENCE PorePy importable (oracle/shim + /root/reference/src + the
reference's own tests): the reference's unit-test model for its differentiable TPFA flux
(tests/numerics/fv/test_tpfa.py: UnitTestAdTpfaFlux - two cells, full-tensor permeability that depends
on the pressure) evaluates ``AdTpfaFlux.__transmissibility_matrix`` through its operator tree and forward
AD; the porepy_amd kernel (host-emulation library here: no GPU in this container) gets the same cell-wise
tensor vector k_c and must return the same transmissibilities and, chained with k_c's Jacobian, the
same derivatives with respect to the pressure unknowns."""
import json
import sys

import numpy as np

import porepy as pp

import importlib.util

import porepy_amd as pa
from tests import _parity as P

# the reference's test module, loaded by path (its package is called ``tests`` like this repo's)
_REF_TEST = "/root/reference/tests/numerics/fv/test_tpfa.py"
if __import__("os").path.exists(_REF_TEST):
    _spec = importlib.util.spec_from_file_location("reference_test_tpfa", _REF_TEST)
    _ref = importlib.util.module_from_spec(_spec)
    _spec.loader.exec_module(_ref)
else:  # the GPU box: the same module byte-compiled into oracle/_ref/porepy_ref.zip (oracle/make_ref.py)
    _ref = importlib.import_module("reference_test_tpfa")
UnitTestAdTpfaFlux = _ref.UnitTestAdTpfaFlux


def main():
    out = {}
    for base in ("tpfa", "mpfa"):
        model = UnitTestAdTpfaFlux({"darcy_flux_discretization": base, "vector_source": np.zeros(4),
                                    "times_to_export": []})
        model.prepare_simulation()
        sds = model.mdg.subdomains()
        sd = sds[0]
        t_op, *_ = model._AdTpfaFlux__transmissibility_matrix(sds, model.permeability)
        t_ref = t_op.value_and_jacobian(model.equation_system)
        # the argument of the reference's expression (constitutive_laws.py:1537-1541)
        basis = model.basis(sds, dim=9)
        volumes = pp.ad.sum_operator_list([e @ model.specific_volume(sds) for e in basis])
        k_c = (volumes * model.permeability(sds)).value_and_jacobian(model.equation_system)
        val, dt_dk = pa.DifferentiableTpfa(library=P.dropin_library()).transmissibility(sd, k_c.val)
        jac = dt_dk @ k_c.jac
        out[base] = {
            "faces": int(sd.num_faces), "dofs": int(t_ref.jac.shape[1]),
            "t_rel_err": float(np.max(np.abs(val - t_ref.val)) / np.max(np.abs(t_ref.val))),
            "jac_rel_err": float(abs(jac - t_ref.jac).max() / abs(t_ref.jac).max()),
            "jac_nnz_ref": int(t_ref.jac.nnz), "jac_max": float(abs(t_ref.jac).max()),
        }
    # the whole flux with the mixin in front of the reference's classes: value and Jacobian of
    # darcy_flux / potential trace must not change
    Mixin = pa.as_porepy_ad_tpfa_flux(library=P.dropin_library())

    class HipModel(Mixin, UnitTestAdTpfaFlux):
        pass

    for base in ("tpfa", "mpfa"):
        res = {}
        for name, cls in (("ref", UnitTestAdTpfaFlux), ("hip", HipModel)):
            model = cls({"darcy_flux_discretization": base, "vector_source": np.array([1.0, 2.0, 3.0, 5.0]),
                         "times_to_export": []})
            model.prepare_simulation()
            sds = model.mdg.subdomains()
            model.discretize()
            flux = model.darcy_flux(sds).value_and_jacobian(model.equation_system)
            trace = model.potential_trace(sds, model.pressure, model.permeability,
                                          model.combine_boundary_operators_darcy_flux, "darcy_flux"
                                          ).value_and_jacobian(model.equation_system)
            res[name] = (flux, trace)
        fr, tr = res["ref"]
        fh, th = res["hip"]
        out[base + "_model"] = {
            "flux_rel_err": float(np.max(np.abs(fh.val - fr.val)) / np.max(np.abs(fr.val))),
            "flux_jac_rel_err": float(abs(fh.jac - fr.jac).max() / abs(fr.jac).max()),
            "trace_rel_err": float(np.max(np.abs(th.val - tr.val)) / np.max(np.abs(tr.val))),
            "trace_jac_rel_err": float(abs(th.jac - tr.jac).max() / abs(tr.jac).max()),
            "flux_jac_nnz": int(fr.jac.nnz),
            "device_path_calls": int(Mixin.hip_differentiable_tpfa.calls),
        }
    out["library"] = str(P.dropin_library()._name)
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    main()
