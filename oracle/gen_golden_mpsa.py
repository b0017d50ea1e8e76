"""Generate MPSA golden fixtures (tests/golden/mpsa_*.npz) by running the REFERENCE PorePy.

TEST INFRASTRUCTURE; build container only:
    cd /tmp && PYTHONDONTWRITEBYTECODE=1 \
      PYTHONPATH=/root/repo/oracle/shim:/root/reference/src:/root/repo \
      python /root/repo/oracle/gen_golden_mpsa.py
Includes the reference's own known-answer cases (tests/numerics/fv/test_mpsa.py:1189-1323,
golden arrays applications/test_utils/reference_dense_arrays.py:53-701).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import scipy.sparse as sps
import scipy.sparse.linalg as spla
import sympy

import porepy as pp
from porepy.applications.test_utils import common_xpfa_tests as xpfa
from porepy.applications.test_utils import reference_dense_arrays as rda

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.gen_golden import OUT, pack_csr  # noqa: E402
from oracle.ref_bridge import grid_to_raw  # noqa: E402

KEYS = ("stress", "bound_stress", "bound_displacement_cell", "bound_displacement_face")


def save_case(name, g, C, bc, bc_values, source, eta=None, extra=None, keys=KEYS, more_params=None):
    params = {"fourth_order_tensor": C, "bc": bc, "inverter": "python", "bc_values": bc_values,
              "source": source}
    if eta is not None:
        params["mpsa_eta"] = eta
    if more_params:
        params.update(more_params)
    data = pp.initialize_data({}, "mechanics", params)
    discr = pp.Mpsa("mechanics")
    discr.discretize(g, data)
    A, b = discr.assemble_matrix_rhs(g, data)
    A = sps.csr_matrix(A)
    x = spla.spsolve(A.tocsc(), b)
    mats = data[pp.DISCRETIZATION_MATRICES]["mechanics"]
    store = {}
    for k, v in grid_to_raw(g).items():
        store["grid_" + k] = np.asarray(v)
    store["bc_is_dir"] = np.asarray(bc.is_dir, bool)
    store["bc_is_neu"] = np.asarray(bc.is_neu, bool)
    store["stiffness"] = np.ascontiguousarray(C.values)
    store["bc_values"] = bc_values
    store["source"] = source
    store["eta"] = np.array(np.nan if eta is None else eta)
    for k in keys:
        pack_csr("ref_" + k, mats[k], store)
    pack_csr("ref_A", A, store)
    store["ref_rhs"] = b
    store["ref_x"] = x
    if extra:
        store.update(extra)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **store)
    print(f"{name:34s} cells={g.num_cells:4d}  {os.path.getsize(path)/1024:.0f} KiB")


def vec_bc(g, mode):
    bf = g.get_all_boundary_faces()
    bc = pp.BoundaryConditionVectorial(g)
    bc.is_neu[:, bf] = True
    bc.is_dir[:, bf] = False
    bot = bf[g.face_centers[g.dim - 1, bf] < 1e-9]
    bc.is_dir[:, bot] = True
    bc.is_neu[:, bot] = False
    if mode == "roller":
        west = bf[g.face_centers[0, bf] < 1e-9]
        bc.is_dir[0, west] = True
        bc.is_neu[0, west] = False
        bc.is_dir[1:, west] = False
        bc.is_neu[1:, west] = True
    if mode == "dir":
        bc.is_dir[:, bf] = True
        bc.is_neu[:, bf] = False
    return bc


def perturb(g, rng, rate):
    x = g.nodes.copy()
    d = g.dim
    inter = np.all((x[:d] > 1e-9) & (x[:d] < x[:d].max(axis=1, keepdims=True) - 1e-9), axis=0)
    x[:d, inter] += (rng.random((d, inter.sum())) - 0.5) * rate
    g.nodes = x
    g.compute_geometry()
    return g


def main():
    rng = np.random.default_rng(20250926)
    g = pp.CartGrid([4, 3]); g.compute_geometry(); nc = g.num_cells
    C = pp.FourthOrderTensor(mu=1 + rng.random(nc), lmbda=1 + rng.random(nc))
    bc = vec_bc(g, "roller")
    bv = (rng.random((2, g.num_faces)) - 0.4) * (bc.is_dir | bc.is_neu)
    save_case("mpsa_cart2d_4x3_roller", g, C, bc, bv.ravel("F"), rng.random(2 * nc) * 0.1)

    g = perturb(pp.StructuredTriangleGrid([3, 3], [1, 1]), rng, 0.08); nc = g.num_cells
    het = np.where(g.cell_centers[0] > 0.5, 1e3, 1.0)
    C = pp.FourthOrderTensor(mu=het * (1 + rng.random(nc)), lmbda=het * (2 + rng.random(nc)))
    bc = vec_bc(g, "neu")
    bv = (rng.random((2, g.num_faces)) - 0.4) * (bc.is_dir | bc.is_neu)
    save_case("mpsa_tri2d_3x3_hetero", g, C, bc, bv.ravel("F"), np.zeros(2 * nc))

    g = perturb(pp.StructuredTetrahedralGrid([2, 2, 2], [1, 1, 1]), rng, 0.1); nc = g.num_cells
    C = pp.FourthOrderTensor(mu=1 + rng.random(nc), lmbda=1 + rng.random(nc))
    bc = vec_bc(g, "roller")
    bv = (rng.random((3, g.num_faces)) - 0.4) * (bc.is_dir | bc.is_neu)
    save_case("mpsa_tet_2x2x2_roller", g, C, bc, bv.ravel("F"), rng.random(3 * nc) * 0.05)

    g = pp.CartGrid([3, 2, 2]); g.compute_geometry(); nc = g.num_cells
    C = pp.FourthOrderTensor(mu=np.ones(nc), lmbda=np.ones(nc))
    bc = vec_bc(g, "dir")
    bv = (rng.random((3, g.num_faces)) - 0.4) * bc.is_dir
    save_case("mpsa_cart3d_3x2x2_dir", g, C, bc, bv.ravel("F"), np.zeros(3 * nc), keys=("stress", "bound_stress"))

    # the reference's own known answers (test_mpsa.py:1189-1323)
    xs, ys = sympy.symbols("x y")

    def chi(xc, yc):
        return np.logical_and(xc > 0.5, yc > 0.5)

    for grid_type in ("cart", "simplex"):
        g_nolines, g_lines = xpfa.create_grid_mpfa_mpsa_reproduce_known_values(grid_type)
        for hetero in (False, True):
            if hetero:
                g, kappa = g_lines, 1e-6
                ux = sympy.sin(2 * sympy.pi * xs) * sympy.sin(2 * sympy.pi * ys)
                uy = sympy.cos(sympy.pi * xs) * (ys - 0.5) ** 2
            else:
                g, kappa = g_nolines, 1.0
                ux = sympy.sin(xs) * sympy.cos(ys)
                uy = sympy.sin(xs) * xs**2
            ux_f, uy_f = sympy.lambdify((xs, ys), ux, "numpy"), sympy.lambdify((xs, ys), uy, "numpy")
            dux_x, dux_y = sympy.diff(ux, xs), sympy.diff(ux, ys)
            duy_x, duy_y = sympy.diff(uy, xs), sympy.diff(uy, ys)
            divu = dux_x + duy_y
            sxx, sxy, syx, syy = 2 * dux_x + divu, dux_y + duy_x, duy_x + dux_y, 2 * duy_y + divu
            rhs_x_f = sympy.lambdify((xs, ys), sympy.diff(sxx, xs) + sympy.diff(syx, ys), "numpy")
            rhs_y_f = sympy.lambdify((xs, ys), sympy.diff(sxy, xs) + sympy.diff(syy, ys), "numpy")
            cfn = chi(g.cell_centers[0], g.cell_centers[1]) * 1.0
            mat = (1 - cfn) + kappa * cfn
            C = pp.FourthOrderTensor(mat, mat)
            bf = g.tags["domain_boundary_faces"].nonzero()[0]
            bc = pp.BoundaryConditionVectorial(g, bf, ["dir"] * bf.size)
            xf = g.face_centers
            cb = chi(xf[0, bf], xf[1, bf]) * 1
            ub = np.zeros((2, g.num_faces))
            ub[0, bf] = ux_f(xf[0, bf], xf[1, bf]) / ((1 - cb) + kappa * cb)
            ub[1, bf] = uy_f(xf[0, bf], xf[1, bf]) / ((1 - cb) + kappa * cb)
            xc = g.cell_centers
            rhs = (np.vstack((rhs_x_f(xc[0], xc[1]), rhs_y_f(xc[0], xc[1]))) * g.cell_volumes).ravel("F")
            key = grid_type + ("_heterogeneous" if hetero else "_homogeneous")
            known = rda.test_mpsa["TestMpsaReproduceKnownValues"][key]
            save_case("mpsa_known_" + key, g, C, bc, ub.ravel("F"), np.zeros(2 * g.num_cells), eta=0.0,
                      extra={"known_u": np.asarray(known["u"], float),
                             "known_stress": np.asarray(known["stress"], float), "known_rhs": rhs},
                      keys=("stress", "bound_stress"))


if __name__ == "__main__":
    main()
