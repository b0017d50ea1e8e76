"""Cases of the device-resident CSR algebra (DeviceCsr, csrc/csr_algebra.inc), shared by the CPU suite (host-emulation
library) and the GPU suite (gfx950 library): every result is compared with scipy's on the same inputs -- bit for bit
where scipy computes the same thing (products, sums, block diagonals follow its accumulation order)."""
import os

import numpy as np
import pytest
import scipy.sparse as sps
import scipy.sparse.linalg as spla

import porepy_amd as pa
from porepy_amd import _lib


def same(a: sps.csr_matrix, b: sps.spmatrix):
    """pattern and values identical (b put into canonical form first)"""
    b = sps.csr_matrix(b)
    b.sort_indices()
    assert a.shape == b.shape
    assert np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices)
    assert np.array_equal(a.data, b.data)


def close(a: sps.csr_matrix, b: sps.spmatrix, tol=1e-14):
    """same pattern, values equal to rounding"""
    b = sps.csr_matrix(b)
    b.sort_indices()
    assert np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices)
    assert np.allclose(a.data, b.data, rtol=0, atol=tol * max(1.0, abs(b.data).max() if b.nnz else 1.0))


def flow_problem(n=3, seed=0, dim=3):
    if dim == 3:
        g = pa.StructuredTetrahedralGrid([n, n, n], [1.0, 1.0, 1.0])
    else:
        g = pa.StructuredTriangleGrid([2 * n, 2 * n], [1.0, 1.0])
    g.compute_geometry()
    g = pa.perturb_interior_nodes(g, 0.1 / n, seed=seed)
    rng = np.random.default_rng(seed)
    nc = g.num_cells
    K = (pa.SecondOrderTensor(kxx=1 + rng.random(nc), kyy=2 + rng.random(nc), kzz=0.5 + rng.random(nc), kxy=0.2 * rng.random(nc))
         if dim == 3 else pa.SecondOrderTensor(kxx=1 + rng.random(nc), kyy=2 + rng.random(nc), kxy=0.2 * rng.random(nc)))
    bf = g.get_all_boundary_faces()
    bc = pa.BoundaryCondition(g, bf, ["dir"] * bf.size)
    bv = np.zeros(g.num_faces)
    bv[bf] = 1.0 + g.face_centers[0, bf]
    data = pa.initialize_data({}, "flow", {"second_order_tensor": K, "bc": bc, "bc_values": bv})
    return g, data


def random_algebra(lib):
    ctx = pa.Context(0, lib)
    rng = np.random.default_rng(7)
    for trial in range(6):
        m, k, n = rng.integers(1, 120, size=3)
        dens = [0.02, 0.1, 0.4][trial % 3]
        A = sps.random(m, k, dens, random_state=int(rng.integers(1 << 30)), format="csr")
        B = sps.random(k, n, dens, random_state=int(rng.integers(1 << 30)), format="csr")
        C = sps.random(m, k, dens, random_state=int(rng.integers(1 << 30)), format="csr")
        # entries that cancel exactly: scipy drops them from sums and products, so does the device
        C = sps.csr_matrix(C - 0.5 * A.multiply(C != 0))
        C.data[::3] = -A[C.nonzero()].A1[::3] if C.nnz else C.data[::3]
        dA, dB, dC = (pa.DeviceCsr.from_scipy(M, ctx) for M in (A, B, C))
        assert dA.shape == A.shape and dA.nnz == A.nnz
        same((dA @ dB).to_scipy(), A @ B)
        same((dA + dC).to_scipy(), A + C)
        same((dA - dC).to_scipy(), A - C)
        same(dA.axpby(0.3, dC, -1.7).to_scipy(), 0.3 * A + (-1.7) * C)
        same((2.5 * dA).to_scipy(), 2.5 * A)
        same((-dA).to_scipy(), -A)
        r, c = rng.random(m) + 0.5, rng.random(k) + 0.5
        same(dA.scaled(r, c).to_scipy(), sps.diags(r) @ A @ sps.diags(c))
        same(pa.block_diag([dA, dB, dC]).to_scipy(), sps.block_diag([A, B, C], format="csr"))
        same(dA.T.to_scipy(), A.T.tocsr())
        same((dA.T @ dC).to_scipy(), A.T.tocsr() @ C)
        same(pa.bmat([[dA, None, dC], [None, dB.T, None]]).to_scipy(),
             sps.bmat([[A, None, C], [None, B.T, None]], format="csr"))
        same(pa.bmat([[dA], [dC]]).to_scipy(), sps.vstack([A, C], format="csr"))
        same(pa.bmat([[dA, dC]]).to_scipy(), sps.hstack([A, C], format="csr"))
        x = rng.random(k)
        assert np.allclose(dA @ x, A @ x, rtol=0, atol=1e-14 * max(1.0, abs(A).sum(axis=1).max()))
        # a chain as the operator tree builds it: (A B) D + E
        D = sps.random(n, m, dens, random_state=int(rng.integers(1 << 30)), format="csr")
        E = sps.random(m, m, dens, random_state=int(rng.integers(1 << 30)), format="csr")
        dD, dE = pa.DeviceCsr.from_scipy(D, ctx), pa.DeviceCsr.from_scipy(E, ctx)
        # (scipy's own product leaves its rows unsorted, so the SECOND product of a chain accumulates in another order
        # there: same pattern, values to rounding; with the intermediate put into canonical form, same bits again)
        chain = (((dA @ dB) @ dD) + dE).to_scipy()
        close(chain, ((A @ B) @ D) + E)
        AB = A @ B
        AB.sort_indices()
        same(chain, (AB @ D) + E)
    # empty operands
    Z = sps.csr_matrix((5, 7))
    dZ = pa.DeviceCsr.from_scipy(Z, ctx)
    same((dZ @ pa.DeviceCsr.from_scipy(sps.random(7, 3, 0.5, random_state=1, format="csr"), ctx)).to_scipy(), sps.csr_matrix((5, 3)))
    same(pa.block_diag([dZ], ctx).to_scipy(), Z)
    ctx.close()


def input_checks(lib):
    ctx = pa.Context(0, lib)
    A = sps.random(20, 20, 0.3, random_state=3, format="csr")
    # from_scipy puts its argument into scipy's canonical form (sorted, duplicates summed) ...
    coo = A.tocoo()
    dup = sps.csr_matrix((np.r_[coo.data, coo.data], (np.r_[coo.row, coo.row], np.r_[coo.col, coo.col])), shape=A.shape)
    dup.has_canonical_format = False
    same(pa.DeviceCsr.from_scipy(dup, ctx).to_scipy(), 2 * A)
    # ... the C ABI itself refuses rows that are not
    ip = np.array([0, 2], dtype=np.int32)
    ix = np.array([3, 1], dtype=np.int32)
    dv = np.ones(2)
    out = _lib._h()
    st = ctx.lib.pfv_csr_from_host(ctx._h, 1, 5, _lib._ptr(ip, _lib._ip), _lib._ptr(ix, _lib._ip), _lib._ptr(dv, _lib._dp),
                                   __import__("ctypes").byref(out))
    assert st == 4  # PFV_ERR_ARGUMENT
    dA = pa.DeviceCsr.from_scipy(A, ctx)
    with pytest.raises(pa.PorefvError, match="inner dimensions"):
        dA @ pa.DeviceCsr.from_scipy(sps.random(19, 4, 0.3, random_state=1, format="csr"), ctx)
    with pytest.raises(pa.PorefvError, match="shapes differ"):
        dA + pa.DeviceCsr.from_scipy(sps.random(20, 19, 0.3, random_state=1, format="csr"), ctx)
    # more products per row than the LDS sort holds: the path through global sorts, same bits
    rng = np.random.default_rng(3)
    big = sps.csr_matrix(rng.random((3, 70)) - 0.5)
    wide = sps.csr_matrix(rng.random((70, 70)) - 0.5)
    same((pa.DeviceCsr.from_scipy(big, ctx) @ pa.DeviceCsr.from_scipy(wide, ctx)).to_scipy(), big @ wide)
    os.environ["PFV_SPGEMM_GENERIC"] = "1"   # ... which any product can be sent through
    try:
        for seed in range(4):
            M1 = sps.random(40, 30, 0.2, random_state=seed, format="csr")
            M2 = sps.random(30, 50, 0.2, random_state=seed + 10, format="csr")
            M2 = sps.csr_matrix(M2 - M2.multiply(M2 > 0.8))  # (some exact cancellations)
            same((pa.DeviceCsr.from_scipy(M1, ctx) @ pa.DeviceCsr.from_scipy(M2, ctx)).to_scipy(), M1 @ M2)
    finally:
        del os.environ["PFV_SPGEMM_GENERIC"]
    with pytest.raises(pa.PorefvError, match="does not fit"):
        pa.bmat([[dA, dA], [pa.DeviceCsr.from_scipy(sps.random(19, 20, 0.3, random_state=1, format="csr"), ctx), dA]])
    with pytest.raises(ValueError, match="size is unknown"):
        pa.bmat([[dA, None], [dA, None]])
    with pytest.raises(pa.PorefvError, match="zero diagonal"):
        pa.DeviceCsr.from_scipy(sps.csr_matrix(np.array([[0.0, 1.0], [1.0, 1.0]])), ctx).as_system(np.ones(2))
    ctx.close()
    # a matrix and its handle dying in one garbage cycle: whichever finalizer runs first, nothing is freed twice and
    # nothing is freed through a dead handle (weak references are cleared before finalizers run)
    import gc

    for _ in range(3):
        c2 = pa.Context(0, lib)
        m2 = pa.DeviceCsr.from_scipy(A, c2)
        c2.keeps = m2  # cycle: matrix -> handle -> matrix
        del c2, m2
        gc.collect()


def discretization_to_system(lib):
    """The flow equation of a subdomain assembled ON THE DEVICE from the discretization matrices: div @ flux is the
    library's own system matrix, (div @ flux, -div @ bound_flux @ bc) solved by the device solver equals scipy's
    solution -- no discretization matrix was copied to the host to get there."""
    g, data = flow_problem(4)
    d = pa.Mpfa("flow", library=lib, lazy=True)
    d.discretize(g, data)
    mats = data[pa.DISCRETIZATION_MATRICES]["flow"]
    ctx = d.context(g)
    flux = pa.DeviceCsr.from_any(mats["flux"], ctx)
    bound_flux = pa.DeviceCsr.from_any(mats["bound_flux"], ctx)
    assert not mats["flux"].materialized and not mats["bound_flux"].materialized  # went device-to-device
    div_h = sps.csr_matrix(g.cell_faces.T)
    div = pa.DeviceCsr.from_scipy(div_h, ctx)
    J = div @ flux
    A_ref, b_ref = d.assemble_matrix_rhs(g, data)
    Jh = J.to_scipy()
    # same values as scipy on the fetched matrices, bit for bit; same matrix as the library's own assembly
    same(Jh, div_h @ mats["flux"].tocsr())
    assert abs(Jh - A_ref).max() <= 1e-13 * abs(A_ref).max()
    bv = data[pa.PARAMETERS]["flow"]["bc_values"]
    rhs = -(div @ (bound_flux @ bv))
    assert np.allclose(rhs, b_ref, rtol=0, atol=1e-12 * abs(b_ref).max())
    solver = J.as_system(rhs)
    x, info = solver.solve("bicgstab", rtol=1e-12, maxit=2000, n=g.num_cells, precond="amg")
    x_ref = spla.spsolve(A_ref.tocsc(), b_ref)
    assert np.linalg.norm(x - x_ref) <= 1e-9 * np.linalg.norm(x_ref)
    # the same through the solver entry point of the host mirror
    x2, info2 = pa.solve_csr(J, rhs, rtol=1e-12, precond="amg")
    assert info2["converged"] and np.linalg.norm(x2 - x_ref) <= 1e-9 * np.linalg.norm(x_ref)


def merged_subdomains(lib):
    """MergedOperator.parse (numerics/ad/ad_utils.py:597-663) on the device: the matrices of a 3-D and a 2-D subdomain,
    each resident on its own handle, concatenated block-diagonally and pushed through the products of the flux
    expression without a host copy."""
    (g3, d3), (g2, d2) = flow_problem(3, 1, dim=3), flow_problem(3, 2, dim=2)
    discr = pa.Mpfa("flow", library=lib, lazy=True)
    discr.discretize(g3, d3)
    discr.discretize(g2, d2)
    ctx = discr.context(g3)
    for key in ("flux", "bound_flux", "vector_source", "bound_pressure_cell", "bound_pressure_face"):
        M = pa.merged_matrix([d3, d2], "flow", key, ctx)
        assert not d3[pa.DISCRETIZATION_MATRICES]["flow"][key].materialized
        ref = sps.block_diag([d3[pa.DISCRETIZATION_MATRICES]["flow"][key].tocsr(), d2[pa.DISCRETIZATION_MATRICES]["flow"][key].tocsr()],
                             format="csr")
        same(M.to_scipy(), ref)
    flux = pa.merged_matrix([d3, d2], "flow", "flux", ctx)
    bound_flux = pa.merged_matrix([d3, d2], "flow", "bound_flux", ctx)
    div_h = sps.block_diag([g3.cell_faces.T, g2.cell_faces.T], format="csr")
    div = pa.DeviceCsr.from_scipy(div_h, ctx)
    # a stand-in for the mortar projection: boundary faces of the 2-D grid fed by a few "mortar" unknowns
    rng = np.random.default_rng(0)
    nf = g3.num_faces + g2.num_faces
    proj_h = sps.random(nf, 11, 0.02, random_state=5, format="csr")
    proj = pa.DeviceCsr.from_scipy(proj_h, ctx)
    fl, bf = flux.to_scipy(), bound_flux.to_scipy()
    same((div @ flux).to_scipy(), div_h @ fl)
    inner = bf @ proj_h
    inner.sort_indices()
    same((div @ (bound_flux @ proj)).to_scipy(), div_h @ inner)
    # ... and stacked with a second equation into one block system, as EquationSystem.assemble stacks the equations of
    # its variables: [[Div Flux, Div BoundFlux P], [P^T BoundPressureCell, P^T BoundPressureFace P + I]]
    bpc = pa.merged_matrix([d3, d2], "flow", "bound_pressure_cell", ctx)
    bpf = pa.merged_matrix([d3, d2], "flow", "bound_pressure_face", ctx)
    eye = pa.DeviceCsr.from_scipy(sps.identity(11, format="csr"), ctx)
    J = pa.bmat([[div @ flux, div @ (bound_flux @ proj)], [proj.T @ bpc, (proj.T @ (bpf @ proj)) + eye]])

    def canon(m):
        m = sps.csr_matrix(m)
        m.sort_indices()
        return m

    projT = canon(proj_h.T)
    J_h = sps.bmat([[div_h @ fl, div_h @ inner],
                    [projT @ bpc.to_scipy(), canon(projT @ canon(bpf.to_scipy() @ proj_h)) + sps.identity(11, format="csr")]],
                   format="csr")
    same(J.to_scipy(), J_h)
    assert J.shape == (g3.num_cells + g2.num_cells + 11,) * 2
    p = rng.random(g3.num_cells + g2.num_cells)
    lam = rng.random(11)
    q = flux @ p + bound_flux @ (proj @ lam)
    assert np.allclose(q, fl @ p + bf @ (proj_h @ lam), rtol=0, atol=1e-12 * abs(q).max())


def forward_mode_array_operand(lib):
    """``DeviceCsr @ a`` for a forward-mode AD array ``a`` (anything with ``val`` and ``jac``; the reference's
    ``AdArray.__rmatmul__`` only takes scipy matrices, numerics/ad/forward_mode.py:565-592): value by the device SpMV,
    Jacobian by the device product, result of the operand's own type."""
    import porepy_amd as pa

    class Ad:
        def __init__(self, val, jac):
            self.val, self.jac = val, jac

    rng = np.random.default_rng(4)
    ctx = pa.Context(0, lib)
    M = sps.random(7, 5, density=0.5, random_state=1, format="csr")
    J = sps.random(5, 9, density=0.4, random_state=2, format="csr")
    v = rng.random(5)
    Md = pa.DeviceCsr.from_scipy(M, ctx)
    for jac in (J, pa.DeviceCsr.from_scipy(J, ctx)):
        r = Md @ Ad(v, jac)
        assert isinstance(r, Ad) and isinstance(r.jac, pa.DeviceCsr)
        assert np.allclose(r.val, M @ v, rtol=0, atol=1e-15)
        close(r.jac.to_scipy(), M @ J)
