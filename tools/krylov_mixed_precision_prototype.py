"""Prototype (scipy, CPU): f32 matrix values in the Krylov products inside an iterative-refinement loop with f64 residuals,
on the library's MPFA flow system (host-emulation build) with the scipy V-cycle of tools/amg_sa_prototype.py -- how many
more iterations the restarts cost against the bytes the f32 values save.   python tools/krylov_mixed_precision_prototype.py [n_side]"""
import sys, time, numpy as np, scipy.sparse as sps, scipy.sparse.linalg as spla
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import amg_sa_prototype as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
A, b = S.flow_system(n)
lv = S.hierarchy(A, False)
M = spla.LinearOperator(A.shape, lambda r: S.vcycle(lv, 0, r))
def solve(Aop, rhs, rtol, x0=None):
    its = [0]
    x, info = spla.bicgstab(Aop, rhs, rtol=rtol, atol=0.0, maxiter=500, M=M, x0=x0, callback=lambda _x: its.__setitem__(0, its[0] + 1))
    return x, its[0]
x, i0 = solve(A, b, 1e-13)
print("f64 Krylov product, rtol 1e-13:", i0, "iterations, true residual", np.linalg.norm(b - A @ x) / np.linalg.norm(b))
A32 = sps.csr_matrix((A.data.astype(np.float32).astype(np.float64), A.indices, A.indptr), shape=A.shape)
for inner in (1e-4, 1e-6, 1e-8):
    x = np.zeros_like(b); tot = 0; outer = 0
    while True:
        r = b - A @ x
        rel = np.linalg.norm(r) / np.linalg.norm(b)
        if rel <= 1e-13 or outer >= 12: break
        dx, it = solve(A32, r, max(inner, 1e-13 / rel * 0.5))
        x += dx; tot += it; outer += 1
    print(f"f32 values in the inner product, inner rtol {inner:.0e}: {outer} outer rounds, {tot} inner iterations (+ {outer} f64 residuals), true residual {rel:.1e}")
