"""numpy model of the UNPIVOTED lane-grid Gauss-Jordan (mpfa_numeric.inc: node_gj_reg2d_np<B>): lane (gi, gj) = 8 gi + gj
holds W[gi + 8 bi][gj + 8 s'], slots rotating once per 8 steps; the pivot of step k = 8 cb + jk is the diagonal entry:
row k lives in register row bi = cb of the lanes with gi == jk, column k in slot 0 of the lanes with gj == jk -- every
index is static.  Pivot rows stay unscaled until the end (as in the pivoted form)."""
import numpy as np


def gj2d_nopivot(A):
    n = A.shape[0]
    B = (n + 7) // 8
    a = np.zeros((64, B, B))
    for l in range(64):
        i, j = l >> 3, l & 7
        for bi in range(B):
            for s in range(B):
                r, c = i + 8 * bi, j + 8 * s
                a[l, bi, s] = A[r, c] if (r < n and c < n) else (1.0 if r == c else 0.0)
    mypivinv = np.ones((64, B))
    for cb in range(B):
        for jk in range(8):
            k = 8 * cb + jk
            if k >= n:
                continue
            piv = a[8 * jk + jk, cb, 0]          # lane (jk, jk), register row cb, slot 0
            pivinv = 1.0 / piv
            prow = np.array([[a[8 * jk + (l & 7), cb, s] for s in range(B)] for l in range(64)])   # from lane (jk, gj)
            gcol = np.array([[a[8 * (l >> 3) + jk, bi, 0] for bi in range(B)] for l in range(64)])  # from lane (gi, jk)
            for l in range(64):
                gi, gj = l >> 3, l & 7
                for bi in range(B):
                    isp = (gi == jk and bi == cb)
                    g = 0.0 if isp else gcol[l, bi] * pivinv
                    for s in range(B):
                        if gj == jk and s == 0:
                            a[l, bi, 0] = 1.0 if isp else -g
                        else:
                            a[l, bi, s] -= g * prow[l, s]
                    if isp:
                        mypivinv[l, bi] = pivinv
        a = np.roll(a, -1, axis=2)
    inv = np.zeros((n, n))
    for l in range(64):
        gi, gj = l >> 3, l & 7
        for bi in range(B):
            for s in range(B):
                r, c = gi + 8 * bi, gj + 8 * s
                if r < n and c < n:
                    inv[r, c] = a[l, bi, s] * mypivinv[l, bi]
    return inv


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for n in (36, 33, 40, 21, 12, 39):
        A = rng.standard_normal((n, n)) * 0.05 + np.eye(n)
        A /= abs(A).sum(1, keepdims=True)
        inv = gj2d_nopivot(A)
        print(n, abs(inv @ A - np.eye(n)).max())
