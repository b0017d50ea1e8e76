"""Prototype (scipy, CPU): the whole AMG cycle in f32 -- matrices AND vectors -- as preconditioner of an f64 BiCGStab, on the
library's MPFA flow system (host-emulation build) with the scipy V-cycle of tools/amg_sa_prototype.py.  The product keeps
the cycle's matrices in f32 and its vectors in f64; the vectors are 25-50 % of the bytes of the cycle's products.
    python tools/amg_f32_cycle_prototype.py [n_side]"""
import os
import sys

import numpy as np
import scipy.sparse as sps
import scipy.sparse.linalg as spla

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import amg_sa_prototype as S  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
A, b = S.flow_system(n)
lv = S.hierarchy(A, False)
lv32 = []
for L in lv:
    M = {"A": L["A"].astype(np.float32), "dinv": L["dinv"].astype(np.float32)}
    if "lu" in L:
        M["lu"] = L["lu"]
    else:
        M["P"], M["R"] = L["P"].astype(np.float32), L["R"].astype(np.float32)
    lv32.append(M)


def vcycle32(l, r, omega=np.float32(0.7)):
    L = lv32[l]
    if "lu" in L:
        return L["lu"].solve(r.astype(np.float64)).astype(np.float32)
    x = omega * L["dinv"] * r
    x = x + L["P"] @ vcycle32(l + 1, L["R"] @ (r - L["A"] @ x))
    return x + omega * L["dinv"] * (r - L["A"] @ x)


def solve(M, label):
    its = [0]
    x, info = spla.bicgstab(A, b, rtol=1e-13, atol=0.0, maxiter=500, M=M, callback=lambda _x: its.__setitem__(0, its[0] + 1))
    print(f"{label}: {its[0]} iterations, true residual {np.linalg.norm(b - A @ x) / np.linalg.norm(b):.1e}, flag {info}", flush=True)


print(f"{A.shape[0]} cells", flush=True)
solve(spla.LinearOperator(A.shape, lambda r: S.vcycle(lv, 0, r)), "cycle in f64                     ")
solve(spla.LinearOperator(A.shape, lambda r: vcycle32(0, r.astype(np.float32)).astype(np.float64)), "cycle in f32 (matrices and vectors)")
