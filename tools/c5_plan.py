"""Plan of the sharded discretization loop of the 52-fracture thermo-hydro model (BASELINE configs[4] stand-in) at 2 / 4 / 8
ranks, whole subdomains only against subdomains + cell pieces of the 3-D matrix grid (porepy_amd/md_sharding.py): cost model
only, nothing is discretized.  Build container (reference importable):
    cd /tmp && C5_N_SIDE=32 PYTHONPATH=/root/repo/oracle/shim:/root/reference/src:/root/repo python /root/repo/tools/c5_plan.py
-> profiles/r06_c5_plan_32cube_52fractures.json"""
import os, sys, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.environ.setdefault("PFV_DROPIN_LIBRARY", "emulation")
import numpy as np
import porepy as pp
import importlib.util
spec = importlib.util.spec_from_file_location("c5", "/root/repo/tests/_dropin_c5_script.py")
c5 = importlib.util.module_from_spec(spec); spec.loader.exec_module(c5)
from porepy_amd import md_sharding as S
c5.rebind()
captured = {}
orig = pp.ad.discretize_from_list
def cap(discretizations, mdg):
    if "d" not in captured:
        captured["d"] = discretizations
    raise KeyboardInterrupt
pp.ad.discretize_from_list = cap
solid = pp.SolidConstants(permeability=0.5, thermal_conductivity=2.0, porosity=0.2, specific_heat_capacity=1.5, normal_permeability=5.0, residual_aperture=1e-1)
fluid = pp.FluidComponent(thermal_conductivity=0.6, specific_heat_capacity=2.0, compressibility=1e-2, thermal_expansion=1e-3, viscosity=1.0)
params = {"times_to_export": [], "linear_solver": "scipy_sparse", "darcy_flux_discretization": "mpfa", "fourier_flux_discretization": "mpfa",
          "material_constants": {"solid": solid, "fluid": fluid}, "time_manager": pp.TimeManager(schedule=[0.0, 0.1], dt_init=0.1, constant_dt=True)}
m = c5.Model(params)
try:
    m.prepare_simulation()
except KeyboardInterrupt:
    pass
d = captured["d"]
cells = {}
for sd in m.mdg.subdomains():
    cells[sd.dim] = cells.get(sd.dim, 0) + sd.num_cells
out = {"n_side": c5.N_SIDE, "fractures": c5.N_FRAC, "cells_by_dim": cells}
for world in (2, 4, 8):
    a = S.plan(d, world, is_interface=lambda g: isinstance(g, pp.MortarGrid), split=False).summary()
    b = S.plan(d, world, is_interface=lambda g: isinstance(g, pp.MortarGrid), split=True).summary()
    out[f"world_{world}"] = {"whole_subdomains": {k: a[k] for k in ("jobs", "speedup_by_cost_model", "bound_total_over_largest_job")},
                             "with_cell_pieces": {k: b[k] for k in ("jobs", "piece_jobs", "subdomains_cut_into_pieces", "speedup_by_cost_model", "speedup_vs_the_serial_loop_by_cost_model", "bound_total_over_largest_job")}}
print(json.dumps(out, indent=1))
