"""Run inside a subprocess with the REFERENCE PorePy importable: BASELINE configs[4] at its stated SIZE OF NETWORK
-- the reference's coupled thermo-hydro model (models/mass_and_energy_balance.py:83) on a mixed-dimensional grid
with **52 fractures** in a 3-D box (one 3-D grid, 52 fracture planes, ~110 intersection lines, ~20 intersection
points, ~390 mortar grids) -- with ``pp.Mpfa`` rebound to the porepy_amd operator and every Newton system solved by
the library's GMRES + block preconditioner.  gmsh is not in this image: the network is made of axis-aligned
rectangles on the planes of a Cartesian grid (deterministic generator below) and meshed by the reference's own
structured mixed-dimensional mesher (``pp.create_mdg("cartesian", ...)``, fracs/structured.py).

Modes:
  (default)       one process: reference untouched vs rebound + device solves; unknowns and last Jacobian compared.
  --save          writes tests/golden/md_thermal_jacobian_box_52fractures.npz (third Newton system + block description).
  --sharded       under ``python -m torch.distributed.run`` (gloo): every rank runs the rebound model twice, with the
                  reference's serial discretization loop and with ``porepy_amd.md_sharding.sharded_discretization``
                  (the (discretization, grid) pairs of ad_utils.py:281-308 dealt out to the ranks, one exchange of
                  the stored matrices); the two runs must agree BITWISE on every rank.
"""
import json
import os
import sys

import numpy as np

import porepy as pp
from porepy.applications.md_grids.domains import nd_cube_domain
from porepy.models.mass_and_energy_balance import MassAndEnergyBalance

import porepy_amd as pa
from porepy_amd import md_sharding
from tests import _parity as P

N_SIDE = int(os.environ.get("C5_N_SIDE", "16"))
N_FRAC = int(os.environ.get("C5_FRACTURES", "52"))
L_MAX = int(os.environ.get("C5_MAX_EXTENT", "7"))


def fracture_rectangles(n: int = N_SIDE, count: int = N_FRAC, lmax: int = L_MAX, seed: int = 3):
    """``count`` rectangles on grid planes of an n^3 Cartesian box: random axis, plane and extent (3..lmax cells a
    side); two rectangles of the same plane keep at least one cell between them (no overlapping co-planar
    fractures); crossing rectangles of different planes make the intersection lines and points."""
    rng = np.random.default_rng(seed)
    placed, out = [], []
    for _ in range(100000):
        if len(out) == count:
            break
        ax, k = int(rng.integers(0, 3)), int(rng.integers(1, n))
        la, lb = int(rng.integers(3, lmax + 1)), int(rng.integers(3, lmax + 1))
        a0, b0 = int(rng.integers(0, n - la + 1)), int(rng.integers(0, n - lb + 1))
        a1, b1 = a0 + la, b0 + lb
        if any(ax2 == ax and k2 == k and not (a1 + 1 < c0 or c1 + 1 < a0 or b1 + 1 < d0 or d1 + 1 < b0)
               for (ax2, k2, c0, c1, d0, d1) in placed):
            continue
        placed.append((ax, k, a0, a1, b0, b1))
        h = 1.0 / n
        other = [d for d in range(3) if d != ax]
        pts = np.zeros((3, 4))
        pts[ax] = k * h
        pts[other[0]] = np.array([a0, a1, a1, a0]) * h
        pts[other[1]] = np.array([b0, b0, b1, b1]) * h
        out.append(pp.PlaneFracture(pts))
    assert len(out) == count
    return out


class Geometry:
    def set_domain(self):
        self._domain = nd_cube_domain(3, 1.0)

    def set_fractures(self):
        self._fractures = fracture_rectangles()

    def grid_type(self):
        return "cartesian"

    def meshing_arguments(self):
        return {"cell_size": 1.0 / N_SIDE}


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
              "time_manager": pp.TimeManager(schedule=[0.0, 0.1], dt_init=0.1, constant_dt=True),
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
    A, _ = m.linear_system
    dims = {}
    for sd in m.mdg.subdomains():
        dims[sd.dim] = dims.get(sd.dim, 0) + 1
    return {"x": np.asarray(x), "A": A.copy(), "dims": dims, "n_intf": len(m.mdg.interfaces()), "solves": solves,
            "cells": int(sum(sd.num_cells for sd in m.mdg.subdomains())),
            "T": np.asarray(m.equation_system.get_variable_values([m.temperature_variable], time_step_index=0)),
            "p": np.asarray(m.equation_system.get_variable_values([m.pressure_variable], time_step_index=0))}


def rebind():
    calls = {}
    HipMpfa = pa.as_porepy_discretization(library=P.dropin_library())
    orig = HipMpfa.discretize

    def counting(self, sd, data):
        key = f"{self.keyword}:{sd.dim}"
        calls[key] = calls.get(key, 0) + 1
        return orig(self, sd, data)

    HipMpfa.discretize = counting
    orig_batch = HipMpfa.discretize_batch

    def counting_batch(self, items):
        items = list(items)
        st = orig_batch(self, items)
        calls["batch_calls"] = calls.get("batch_calls", 0) + 1
        calls["grids_in_batch_calls"] = calls.get("grids_in_batch_calls", 0) + len(items)
        calls["device_unions"] = calls.get("device_unions", 0) + st["unions"]
        calls["grids_in_unions"] = calls.get("grids_in_unions", 0) + st["batched"]
        calls["grids_alone"] = calls.get("grids_alone", 0) + st["single"]
        return st

    HipMpfa.discretize_batch = counting_batch
    pp.Mpfa = HipMpfa
    return calls


def save_first_jacobian():
    """tests/golden/md_thermal_jacobian_box_52fractures.npz: the third Newton system of the rebound model with the
    solver's own block description (variable-wide blocks, equation pairing, mask of the interface unknowns)."""
    import scipy.sparse as sps

    class Saved(Exception):
        pass

    class M(HipSolveModel):
        def solve_linear_system(self):
            self._solves = getattr(self, "_solves", 0) + 1
            if self._solves < 3:  # (the third Newton system: advective coupling of T to the fluxes is in)
                return super().solve_linear_system()
            A, b = self.linear_system
            block_of, row_perm = self._hip_blocks({})
            A = sps.csr_matrix(A)
            path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "md_thermal_jacobian_box_52fractures.npz")
            np.savez_compressed(path, data=A.data, indices=A.indices, indptr=A.indptr, shape=np.array(A.shape), b=b,
                                block_of=block_of, row_perm=row_perm, interface=self._hip_interface_mask)
            raise Saved()

    rebind()
    try:
        run(M, "hip_gmres", {"precond": "block"})
    except Saved:
        print("RESULT " + json.dumps({"saved": True}))


def main_serial():
    ref = run()
    calls = rebind()
    # the subdomain loop hands all grids of a discretization object to its discretize_batch: the 52 fracture planes
    # are ONE disjoint union on the device per keyword (the 3-D grid is alone in its dimension: single-grid path)
    loop_stats = {}
    with md_sharding.batched_discretization(pp, stats=loop_stats):
        both = run(HipSolveModel, "hip_gmres", {"precond": "block", "rtol": 1e-13, "restart": 80})
    nrm = np.linalg.norm(ref["x"])
    out = {
        "fractures": ref["dims"].get(2, 0), "lines": ref["dims"].get(1, 0), "points": ref["dims"].get(0, 0),
        "interfaces": ref["n_intf"], "cells": ref["cells"], "dofs": int(ref["x"].size), "device_calls": calls,
        "loop": {k: loop_stats.get(k) for k in ("calls", "batch_calls", "jobs_in_batches")},
        "x_rel_err": float(np.linalg.norm(both["x"] - ref["x"]) / nrm),
        "T_rel_err": float(np.linalg.norm(both["T"] - ref["T"]) / np.linalg.norm(ref["T"])),
        "p_rel_err": float(np.linalg.norm(both["p"] - ref["p"]) / np.linalg.norm(ref["p"])),
        "A_rel_err": float(abs(both["A"] - ref["A"]).max() / abs(ref["A"]).max()),
        "hip_linear_solves": len(both["solves"]),
        "hip_solver_max_iterations": int(max(s_["iterations"] for s_ in both["solves"])),
        "hip_solver_blocks": int(both["solves"][0]["blocks"]),
        "hip_solver_worst_true_residual": float(max(s_["true_rel_residual"] for s_ in both["solves"])),
        "T_range": [float(ref["T"].min()), float(ref["T"].max())],
        "library": str(P.dropin_library()._name),
    }
    print("RESULT " + json.dumps(out))


def main_sharded():
    import torch.distributed as dist

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    calls = rebind()
    serial = run()
    n_serial = sum(calls.values())
    calls.clear()
    stats = {}
    # C5_SPLIT=1 (round 6): the jobs that bound the loop -- the 3-D matrix grid under Darcy and Fourier -- are cut into
    # cell pieces (node ring of overlap) that are dealt out with the fracture jobs; rows merged on every rank.  The
    # default (0) hands out whole subdomains only, which reproduces the serial loop bit for bit.
    split = os.environ.get("C5_SPLIT", "0") == "1"
    with md_sharding.sharded_discretization(pp, stats=stats, split=split):
        if split:
            # ... and every Newton system solved SHARDED over the same ranks (solve_block_system_sharded through the
            # HipLinearSolver mixin: interface fluxes condensed, unknowns dealt out by position inside every variable)
            sharded = run(HipSolveModel, "hip_bicgstab", {"precond": "block", "rtol": 1e-13,
                                                          "sharded": {"dist": dist, "device": "cpu"}})
        else:
            sharded = run()
    n_here = sum(v for k, v in calls.items() if ":" in k) + calls.get("grids_in_batch_calls", 0)
    same = bool(np.array_equal(serial["x"], sharded["x"]) and (serial["A"] != sharded["A"]).nnz == 0)
    x_rel = float(np.linalg.norm(serial["x"] - sharded["x"]) / np.linalg.norm(serial["x"]))
    a_rel = float(abs(serial["A"] - sharded["A"]).max() / abs(serial["A"]).max())
    # what the same plan would be on 8 ranks (cost model only: nobody runs it here)
    gathered = [None] * world
    dist.all_gather_object(gathered, {"rank": rank, "same": same, "x_rel_err": x_rel, "A_rel_err": a_rel,
                                      "device_calls_here": n_here,
                                      "sharded_solves": [dict(iterations=int(q.get("iterations", -1)),
                                                              world=int(q.get("sharded_world", 0)))
                                                         for q in sharded.get("solves", [])],
                                      "device_calls_serial": n_serial, "stats": stats,
                                      "device_unions": calls.get("device_unions", 0)})
    if rank == 0:
        out = {"world": world, "ranks": gathered, "dofs": int(serial["x"].size),
               "fractures": serial["dims"].get(2, 0), "library": str(P.dropin_library()._name)}
        print("RESULT " + json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    if "--sharded" in sys.argv:
        main_sharded()
    elif "--save" in sys.argv:
        save_first_jacobian()
    else:
        main_serial()
