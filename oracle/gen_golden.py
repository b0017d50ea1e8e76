"""Generate golden fixtures under tests/golden/ by running the REFERENCE PorePy.

TEST INFRASTRUCTURE.  Runs only in the build container (needs /root/reference):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 \
      PYTHONPATH=/root/repo/oracle/shim:/root/reference/src:/root/repo \
      python /root/repo/oracle/gen_golden.py

Each fixture is an .npz holding the raw grid arrays, the parameters, and the
reference's outputs (six discretization matrices in CSR form, A, b, and the direct
solution).  The known-answer cases of the reference's own test-suite
(tests/numerics/fv/test_mpfa.py:140-250, golden arrays in
applications/test_utils/reference_dense_arrays.py:807-1138) are stored with the
hard-coded expected vectors next to what the reference computes today.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import scipy.sparse as sps
import scipy.sparse.linalg as spla

import porepy as pp
from porepy.applications.test_utils import common_xpfa_tests as xpfa
from porepy.applications.test_utils import reference_dense_arrays as rda

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.ref_bridge import bc_to_raw, grid_to_raw  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
KEYS = (
    "flux",
    "bound_flux",
    "bound_pressure_cell",
    "bound_pressure_face",
    "vector_source",
    "bound_pressure_vector_source",
)


def pack_csr(prefix: str, m, store: dict):
    m = sps.csr_matrix(m)
    m.sort_indices()
    store[prefix + "_indptr"] = m.indptr.astype(np.int32)
    store[prefix + "_indices"] = m.indices.astype(np.int32)
    store[prefix + "_data"] = m.data.astype(np.float64)
    store[prefix + "_shape"] = np.array(m.shape, dtype=np.int64)


def run_reference(g, K, bc, bc_values, source, eta=None, vector_source=None):
    params = {"second_order_tensor": K, "bc": bc, "bc_values": bc_values, "mpfa_inverter": "python"}
    if eta is not None:
        params["mpfa_eta"] = eta
    if vector_source is not None:
        params["vector_source"] = vector_source
    data = pp.initialize_data({}, "flow", params)
    discr = pp.Mpfa("flow")
    discr.discretize(g, data)
    mats = data[pp.DISCRETIZATION_MATRICES]["flow"]
    A, b = discr.assemble_matrix_rhs(g, data)
    return mats, sps.csr_matrix(A), b + source


def save_case(name, g, K, bc, bc_values, source, eta=None, vector_source=None, extra=None,
              keys=KEYS):
    mats, A, rhs = run_reference(g, K, bc, bc_values, source, eta, vector_source)
    x = spla.spsolve(A.tocsc(), rhs)
    store = {}
    for k, v in grid_to_raw(g).items():
        store["grid_" + k] = np.asarray(v)
    for k, v in bc_to_raw(bc).items():
        store["bc_" + k] = v
    store["perm"] = np.ascontiguousarray(K.values)
    store["bc_values"] = bc_values
    store["source"] = source
    store["eta"] = np.array(np.nan if eta is None else eta)
    if vector_source is not None:
        store["vector_source_values"] = vector_source
    for k in keys:
        pack_csr("ref_" + k, mats[k], store)
    pack_csr("ref_A", A, store)
    store["ref_rhs"] = rhs
    store["ref_x"] = x
    if extra:
        store.update(extra)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **store)
    print(f"{name:32s} cells={g.num_cells:5d} faces={g.num_faces:5d}  {os.path.getsize(path)/1024:.0f} KiB")


def perturb_interior(g, rng, rate):
    x = g.nodes.copy()
    lo, hi = x.min(axis=1, keepdims=True), x.max(axis=1, keepdims=True)
    interior = np.all((x[: g.dim] > lo[: g.dim] + 1e-9) & (x[: g.dim] < hi[: g.dim] - 1e-9), axis=0)
    x[: g.dim, interior] += (rng.random((g.dim, interior.sum())) - 0.5) * rate
    g.nodes = x
    g.compute_geometry()
    return g


def mixed_bc(g, kinds, robin_weight=1.5):
    bf = g.get_all_boundary_faces()
    lab = np.array(kinds)[np.arange(bf.size) % len(kinds)]
    bc = pp.BoundaryCondition(g, bf, list(lab))
    bc.robin_weight = robin_weight * np.ones(g.num_faces)
    return bc


def bc_vals(g, bc, rng):
    v = np.zeros(g.num_faces)
    bf = g.get_all_boundary_faces()
    v[bf] = rng.random(bf.size) - 0.3
    return v


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20250925)

    # --- small synthetic cases, every matrix -------------------------------------
    g = pp.CartGrid([4, 3]); g.compute_geometry()
    nc = g.num_cells
    K = pp.SecondOrderTensor(kxx=1 + rng.random(nc), kyy=2 + rng.random(nc), kxy=0.3 * rng.random(nc))
    bc = mixed_bc(g, ["dir", "neu", "rob"])
    save_case("cart2d_4x3_mixed", g, K, bc, bc_vals(g, bc, rng), rng.random(nc) * g.cell_volumes)

    g = perturb_interior(pp.StructuredTriangleGrid([4, 4], [1, 1]), rng, 0.08)
    nc = g.num_cells
    K = pp.SecondOrderTensor(kxx=1 + rng.random(nc), kyy=3 + rng.random(nc), kxy=0.4 * rng.random(nc))
    bc = mixed_bc(g, ["dir", "dir", "neu", "rob"])
    save_case("tri2d_4x4_mixed", g, K, bc, bc_vals(g, bc, rng), rng.random(nc) * g.cell_volumes)

    g = pp.CartGrid([3, 3, 3]); g.compute_geometry()
    nc = g.num_cells
    K = pp.SecondOrderTensor(np.where(g.cell_centers[0] > 1.5, 1e6, 1.0))
    bc = mixed_bc(g, ["dir", "neu"])
    save_case("cart3d_3x3x3_hetero", g, K, bc, bc_vals(g, bc, rng), rng.random(nc) * g.cell_volumes)

    g = perturb_interior(pp.StructuredTetrahedralGrid([3, 3, 3], [1, 1, 1]), rng, 0.2 / 3)
    nc = g.num_cells
    kk = np.where(g.cell_centers[0] > 0.5, 1e-3, 1.0)
    K = pp.SecondOrderTensor(
        kxx=kk * (1 + rng.random(nc)), kyy=kk * (10 + rng.random(nc)), kzz=kk * (0.1 + rng.random(nc)),
        kxy=kk * 0.5 * rng.random(nc), kxz=kk * 0.05 * rng.random(nc), kyz=kk * 0.2 * rng.random(nc),
    )
    bc = mixed_bc(g, ["dir", "neu", "rob"])
    gvec = rng.random(3 * nc) - 0.5
    save_case("tet_3x3x3_mixed_aniso", g, K, bc, bc_vals(g, bc, rng), rng.random(nc) * g.cell_volumes,
              vector_source=gvec)

    # generic anisotropic, all Dirichlet: the reference's stored pattern equals the
    # structural stencil here (SURVEY note N3) -> pattern must match bit-exactly.
    g = perturb_interior(pp.StructuredTetrahedralGrid([2, 2, 2], [1, 1, 1]), rng, 0.1)
    nc = g.num_cells
    K = pp.SecondOrderTensor(kxx=np.ones(nc), kyy=10 * np.ones(nc), kzz=0.1 * np.ones(nc),
                             kxy=0.5 * np.ones(nc), kxz=0.05 * np.ones(nc), kyz=0.2 * np.ones(nc))
    bf = g.get_all_boundary_faces()
    bc = pp.BoundaryCondition(g, bf, ["dir"] * bf.size)
    bv = np.zeros(g.num_faces); bv[bf] = g.face_centers[0, bf]
    save_case("tet_2x2x2_dir_generic", g, K, bc, bv, g.cell_volumes.copy())

    # structured tets, isotropic, Dirichlet p = x (mini version of config C2)
    g = pp.StructuredTetrahedralGrid([4, 4, 4], [1, 1, 1]); g.compute_geometry()
    nc = g.num_cells
    K = pp.SecondOrderTensor(np.ones(nc))
    bf = g.get_all_boundary_faces()
    bc = pp.BoundaryCondition(g, bf, ["dir"] * bf.size)
    bv = np.zeros(g.num_faces); bv[bf] = g.face_centers[0, bf]
    save_case("tet_4x4x4_iso_linear", g, K, bc, bv, np.zeros(nc), keys=("flux", "bound_flux"))

    # --- the reference's own known-answer cases (test_mpfa.py:140-250) -----------
    import sympy

    def chi(xc, yc):
        return np.logical_and(xc > 0.5, yc > 0.5)

    xs, ys = sympy.symbols("x y")
    for grid_type in ("cart", "simplex"):
        g_nolines, g_lines = xpfa.create_grid_mpfa_mpsa_reproduce_known_values(grid_type)
        for hetero in (False, True):
            if hetero:
                g = g_lines
                kappa = 1e-6 if grid_type == "cart" else 1e6
                u = sympy.sin(2 * sympy.pi * xs) * sympy.sin(2 * sympy.pi * ys)
            else:
                g = g_nolines
                kappa = 1.0
                u = sympy.sin(xs) * sympy.cos(ys)
            u_f = sympy.lambdify((xs, ys), u, "numpy")
            rhs_f = sympy.lambdify(
                (xs, ys), -sympy.diff(u, xs, 2) - sympy.diff(u, ys, 2), "numpy"
            )
            cf = chi(g.cell_centers[0], g.cell_centers[1]) * 1.0
            K = pp.SecondOrderTensor((1 - cf) + kappa * cf)
            bf = g.tags["domain_boundary_faces"].nonzero()[0]
            bc = pp.BoundaryCondition(g, bf, ["dir"] * bf.size)
            fb = chi(g.face_centers[0, bf], g.face_centers[1, bf]) * 1
            ub = np.zeros(g.num_faces)
            ub[bf] = u_f(g.face_centers[0, bf], g.face_centers[1, bf]) / ((1 - fb) + kappa * fb)
            src = rhs_f(g.cell_centers[0], g.cell_centers[1]) * g.cell_volumes
            key = grid_type + ("_heterogeneous" if hetero else "_homogeneous")
            known = rda.test_mpfa["TestMpfaReproduceKnownValues"][key]
            save_case(
                "known_" + key, g, K, bc, ub, src, eta=0.0,
                extra={"known_u": np.asarray(known["u"], float), "known_flux": np.asarray(known["flux"], float)},
                keys=("flux", "bound_flux"),
            )

    # --- scalar known answers: tutorial sum and config C1 ------------------------
    # tutorials/flux_discretizations.ipynb cell 30: sum(p) == 14.192684340967542
    g = pp.CartGrid([20, 20], [1, 1]); g.compute_geometry()
    K = pp.SecondOrderTensor(np.ones(g.num_cells))
    bf = g.get_all_boundary_faces()
    bc = pp.BoundaryCondition(g, bf, ["dir"] * bf.size)
    mats, A, rhs = run_reference(g, K, bc, np.zeros(g.num_faces), g.cell_volumes.copy())
    p20 = spla.spsolve(A.tocsc(), rhs)
    # config C1: CartGrid([50,50],[1,1]), K=I, Dirichlet west 5 / east 2, Neumann-0 else
    g = pp.CartGrid([50, 50], [1, 1]); g.compute_geometry()
    K = pp.SecondOrderTensor(np.ones(g.num_cells))
    bf = g.get_all_boundary_faces()
    west = bf[g.face_centers[0, bf] < 1e-10]
    east = bf[g.face_centers[0, bf] > 1 - 1e-10]
    bc = pp.BoundaryCondition(g, np.r_[west, east], ["dir"] * (west.size + east.size))
    bv = np.zeros(g.num_faces); bv[west] = 5.0; bv[east] = 2.0
    mats, A, rhs = run_reference(g, K, bc, bv, np.zeros(g.num_cells))
    p50 = spla.spsolve(A.tocsc(), rhs)
    np.savez_compressed(
        os.path.join(OUT, "scalar_known_answers.npz"),
        tutorial_sum_documented=np.array(14.192684340967542),
        tutorial_sum_reference_today=np.array(p20.sum()),
        c1_sum_reference_today=np.array(p50.sum()),
        c1_min_max=np.array([p50.min(), p50.max()]),
        c1_A_nnz=np.array(A.nnz),
    )
    print("tutorial sum", p20.sum(), " C1 sum", p50.sum(), "C1 A nnz", A.nnz)


if __name__ == "__main__":
    main()
