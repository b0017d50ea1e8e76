import numpy as np
def gj2d(A):
    n = A.shape[0]; B = (n + 7)//8; 
    a = np.zeros((64, B, B)); 
    for l in range(64):
        i, j = l >> 3, l & 7
        for bi in range(B):
            for s in range(B):
                r, c = i + 8*bi, j + 8*s
                a[l, bi, s] = A[r, c] if (r < n and c < n) else (1.0 if r == c else 0.0)
    used = np.zeros((64, B), bool); mystep = np.zeros((64, B), int); mypivinv = np.ones((64, B))
    for l in range(64):
        for bi in range(B): used[l, bi] = (l >> 3) + 8*bi >= n
    ipiv = np.zeros(8*B, int)
    for cb in range(B):
        for jk in range(8):
            k = 8*cb + jk
            if k < n:
                # pivot search
                best, p = -1.0, -1
                for l in range(64):
                    i, j = l >> 3, l & 7
                    if j != jk: continue
                    for bi in range(B):
                        r = i + 8*bi
                        if not used[l, bi] and abs(a[l, bi, 0]) > best:
                            best, p = abs(a[l, bi, 0]), r
                ip, bp = p & 7, p >> 3
                piv = a[8*ip + jk, bp, 0]; pivinv = 1.0/piv
                ipiv[k] = p
                prow = np.array([[a[8*ip + (l & 7), bp, s] for s in range(B)] for l in range(64)])
                gcol = np.array([[a[8*(l >> 3) + jk, bi, 0] for bi in range(B)] for l in range(64)])
                for l in range(64):
                    i, j = l >> 3, l & 7
                    for bi in range(B):
                        r = i + 8*bi
                        isp = (r == p)
                        g = 0.0 if isp else gcol[l, bi]*pivinv
                        for s in range(B):
                            if j == jk and s == 0:
                                a[l, bi, 0] = 1.0 if isp else -g
                            else:
                                a[l, bi, s] -= g*prow[l, s]
                        if isp:
                            used[l, bi] = True; mystep[l, bi] = k; mypivinv[l, bi] = pivinv
        a = np.roll(a, -1, axis=2)   # slot s <- s+1, slot B-1 <- old slot 0
    inv = np.zeros((n, n))
    for l in range(64):
        i, j = l >> 3, l & 7
        for bi in range(B):
            for s in range(B):
                r, c = i + 8*bi, j + 8*s
                if r < n and c < n:
                    inv[mystep[l, bi], ipiv[c]] = a[l, bi, s]*mypivinv[l, bi]
    return inv
rng = np.random.default_rng(0)
for n in (36, 33, 40, 41, 48, 57, 64, 12):
    A = rng.standard_normal((n, n)); 
    A /= abs(A).sum(1, keepdims=True)
    inv = gj2d(A)
    print(n, abs(inv @ A - np.eye(n)).max())
