"""Checks of the block-preconditioned device solve (pfv_set_block_preconditioner; porepy_amd.solvers) shared by the CPU
suite (host-emulation build) and the GPU suite (libporefv_hip.so)."""
import os

import numpy as np
import scipy.sparse as sps
import scipy.sparse.linalg as spla

from porepy_amd import solvers

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def thermo_hydro_jacobian(lib):
    """The Jacobian of the reference's thermo-hydro model on the mixed-dimensional stand-in (made by
    tests/_dropin_thermal_script.py --save): 440 unknowns in 20 blocks, equations ordered differently from the unknowns
    (364 structurally zero diagonal entries), mortar coupling.  With the model's block pairing GMRES converges to the
    direct solution; block Gauss-Seidel needs no more iterations than block Jacobi.  (With an entry-wise matching instead
    of the pairing -- returned as "entrywise_converged" -- GMRES stagnates on this system: some pressure unknowns get
    paired with energy equations and their diagonal blocks are no discretization of anything.)"""
    z = np.load(os.path.join(GOLDEN, "md_thermal_jacobian_box_2fractures.npz"))
    A = sps.csr_matrix((z["data"], z["indices"], z["indptr"]), shape=tuple(z["shape"]))
    b, block_of, row_perm = z["b"], z["block_of"], z["row_perm"]
    x_ref = spla.spsolve(A.tocsc(), b)
    assert int((A.diagonal() == 0).sum()) > 300
    out = {}
    for gs in (True, False):
        x, info = solvers.solve_block_system(A, b, block_of, rtol=1e-13, restart=80, library=lib, gauss_seidel=gs,
                                             row_perm=row_perm)
        assert info["converged"] and info["blocks"] == 20 and info["rows_matched"]
        assert info["true_rel_residual"] < 1e-12
        assert np.linalg.norm(x - x_ref) <= 1e-10 * np.linalg.norm(x_ref)
        out["gs" if gs else "jacobi"] = info["iterations"]
    assert out["gs"] <= out["jacobi"]
    # one block = the whole system inverted densely: the preconditioner is A^-1 up to rounding
    x, info = solvers.solve_block_system(A, b, np.zeros(A.shape[0], int), rtol=1e-13, library=lib)
    assert info["converged"] and info["iterations"] <= 12
    assert np.linalg.norm(x - x_ref) <= 1e-10 * np.linalg.norm(x_ref)
    out["one_block"] = info["iterations"]
    _, info = solvers.solve_block_system(A, b, block_of, rtol=1e-13, restart=80, maxit=240, library=lib)
    out["entrywise_converged"] = bool(info["converged"])
    return out


def flow_blocks_with_amg(lib, n=14):
    """Blocks beyond the dense limit take the AMG cycle: a two-field system [[L, eps I], [eps I, L + I]] of two 3-D
    Laplacians (n^3 unknowns each, > 1024) solved with two AMG blocks."""
    import porepy_amd as pa

    g = pa.CartGrid([n, n, n], [1, 1, 1])
    g.compute_geometry()
    N = g.num_cells
    e = np.ones(n)
    T = sps.diags([-e[:-1], 2 * e, -e[:-1]], [-1, 0, 1])
    I1 = sps.identity(n)
    L = sps.kron(sps.kron(T, I1), I1) + sps.kron(sps.kron(I1, T), I1) + sps.kron(sps.kron(I1, I1), T)
    A = sps.bmat([[L, 0.3 * sps.identity(N)], [-0.2 * sps.identity(N), L + sps.identity(N)]], format="csr")
    rng = np.random.default_rng(0)
    b = rng.random(2 * N)
    x, info = solvers.solve_block_system(A, b, np.repeat([0, 1], N), rtol=1e-12, restart=40, library=lib)
    assert info["converged"] and not info["rows_matched"] and info["true_rel_residual"] < 1e-11
    assert N > 1024 and info["iterations"] < 40
    return info["iterations"]


def thermo_hydro_jacobian_52_fractures(lib):
    """The third Newton system of the thermo-hydro model on the 52-fracture network (tests/_dropin_c5_script.py --save:
    21 360 unknowns; 385 interfaces with three flux variables each).  Sweeps over (variable, grid) or variable-wide
    blocks do not converge on it (returned as "without_condensation_converged"); with the interface unknowns condensed
    on the device (solve_block_system: eliminate) GMRES and BiCGStab reach the direct solution."""
    z = np.load(os.path.join(GOLDEN, "md_thermal_jacobian_box_52fractures.npz"))
    A = sps.csr_matrix((z["data"], z["indices"], z["indptr"]), shape=tuple(z["shape"]))
    b, block_of, row_perm, interface = z["b"], z["block_of"], z["row_perm"], z["interface"]
    assert A.shape[0] == 21360 and int(interface.sum()) == 3 * 3392 and np.unique(block_of).size == 5
    x_ref = spla.spsolve(A.tocsc(), b)
    out = {}
    for method in ("gmres", "bicgstab"):
        x, info = solvers.solve_block_system(A, b, block_of, method=method, rtol=1e-13, restart=80, library=lib,
                                             row_perm=row_perm, eliminate=interface)
        assert info["converged"] and info["condensed_unknowns"] == int(interface.sum())
        assert info["true_rel_residual"] < 1e-12
        assert np.linalg.norm(x - x_ref) <= 1e-10 * np.linalg.norm(x_ref)
        out[method] = info["iterations"]
    _, info = solvers.solve_block_system(A, b, block_of, rtol=1e-13, restart=80, maxit=160, library=lib, row_perm=row_perm)
    out["without_condensation_converged"] = bool(info["converged"])
    # a block is condensed as a whole or not at all
    half = interface.copy()
    half[np.where(interface)[0][::2]] = False
    try:
        solvers.solve_block_system(A, b, block_of, library=lib, row_perm=row_perm, eliminate=half)
        raise AssertionError("a partly condensed block must be refused")
    except ValueError:
        pass
    return out
