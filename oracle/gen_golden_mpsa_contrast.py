"""Generate the MPSA high-contrast fixtures (tests/golden/mpsacontrast_*.npz): Lame parameters of neighbouring cells
apart by 1e8 / 1e10 / 1e12, run through the REFERENCE's pp.Mpsa -- and, beside it, through the numpy oracle with its
local gradient systems inverted by mpmath in 60-digit arithmetic ("exact": the result the reference's own systems have
when nothing is lost to FP64; beyond ~1e10 the reference's LAPACK inverse itself is off it by 1e-9 ... 1e-4, see
tools/fuzz_vs_reference.py ... contrast).  A fixture stores both and the reference's distance to the exact result; the
tests hold the device (double-double assembly of the flagged interaction regions, csrc/mpsa.inc + dd.h) to 1e-10 of the
exact matrices always, and of the reference's wherever the reference itself is that close.

TEST INFRASTRUCTURE; build container only:
    cd /tmp && PYTHONDONTWRITEBYTECODE=1 \
      PYTHONPATH=/root/repo/oracle/shim:/root/reference/src:/root/repo \
      python /root/repo/oracle/gen_golden_mpsa_contrast.py
Ref: numerics/fv/mpsa.py:784-930 (the gradient system whose columns keep stiff and soft sub-cells apart).
"""
from __future__ import annotations

import os
import sys

import numpy as np

import porepy as pp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mpsa_oracle as so  # noqa: E402
from oracle.gen_golden import OUT, pack_csr  # noqa: E402
from oracle.gen_golden_mpsa import KEYS, perturb  # noqa: E402
from oracle.ref_bridge import grid_to_raw  # noqa: E402


def exact_matrices(g, C, bc):
    """2-D: every step of the node-local computation in 60-digit arithmetic from the FP64 inputs on (the exact answer to
    the problem the inputs pose); 3-D: the FP64-assembled local systems inverted in 60 digits (object arrays of 216 x 216
    per node are too slow) -- biased towards the reference's own assembly, see tools/fuzz_vs_reference.py: _exact_mechanics."""
    import mpmath as mp

    mp.mp.dps = 60
    if g.dim == 2:
        return so.discretize(grid_to_raw(g), C.values, {"is_dir": bc.is_dir, "is_neu": bc.is_neu}, real=mp.mpf)
    inv0, cond0 = np.linalg.inv, np.linalg.cond
    np.linalg.inv = lambda M: np.array((mp.matrix(M.tolist()) ** -1).tolist(), dtype=float)
    np.linalg.cond = lambda M: 1.0
    try:
        return so.discretize(grid_to_raw(g), C.values, {"is_dir": bc.is_dir, "is_neu": bc.is_neu})
    finally:
        np.linalg.inv, np.linalg.cond = inv0, cond0


def rel(a, b):
    return abs(a - b).max() / max(abs(b).max(), 1e-300)


def save_case(name, g, decades, rng, two_valued=True):
    """Fields drawn exactly as tools/fuzz_vs_reference.py: case_contrast draws them (mechanics leg)."""
    nc, nd = g.num_cells, g.dim
    bf = g.get_all_boundary_faces()
    if two_valued is None:  # exactly the driver's sequence of draws (case_contrast): field, coin, flow tensors, flow conditions
        s = 10.0 ** (decades * (rng.random(nc) - 0.5))
        if rng.random() < 0.5:
            s = np.where(rng.random(nc) < 0.5, 10.0 ** (-decades / 2), 10.0 ** (decades / 2))
        for _ in range(3 if nd == 2 else 6):
            rng.random(nc)
        rng.choice(["dir", "neu"], size=bf.size, p=[0.6, 0.4])
        rng.integers(0, bf.size)
    elif two_valued:
        s = np.where(rng.random(nc) < 0.5, 10.0 ** (-decades / 2), 10.0 ** (decades / 2))
    else:
        s = 10.0 ** (decades * (rng.random(nc) - 0.5))
    bc = pp.BoundaryConditionVectorial(g)
    for a in range(nd):
        tdir = rng.random(bf.size) < 0.6
        bc.is_dir[a, bf[tdir]], bc.is_neu[a, bf[tdir]] = True, False
    bc.is_dir[:, bf[:2]], bc.is_neu[:, bf[:2]] = True, False
    mu, lam = s * (0.5 + rng.random(nc)), s * (0.5 + rng.random(nc))
    C = pp.FourthOrderTensor(mu, lam)
    data = pp.initialize_data({}, "mechanics", {"fourth_order_tensor": C, "bc": bc, "inverter": "python"})
    pp.Mpsa("mechanics").discretize(g, data)
    mats = data[pp.DISCRETIZATION_MATRICES]["mechanics"]
    ex = exact_matrices(g, C, bc)
    store = {}
    for k, v in grid_to_raw(g).items():
        store["grid_" + k] = np.asarray(v)
    store["bc_is_dir"] = np.asarray(bc.is_dir, bool)
    store["bc_is_neu"] = np.asarray(bc.is_neu, bool)
    store["stiffness"] = np.ascontiguousarray(C.values)
    store["decades"] = np.array(float(decades))
    off = {}
    for k in KEYS:
        pack_csr("ref_" + k, mats[k], store)
        pack_csr("exact_" + k, ex[k], store)
        off[k] = rel(mats[k], ex[k])
        store["ref_off_exact_" + k] = np.array(off[k])
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **store)
    print(f"{name:34s} cells={nc:4d} contrast 1e{decades:.0f}  reference vs exact: "
          + ", ".join(f"{k} {v:.1e}" for k, v in off.items()) + f"  {os.path.getsize(path) / 1024:.0f} KiB")


def main():
    from tools.fuzz_vs_reference import random_ref_grid

    # A 5 x 5 Cartesian grid with a two-valued field (seed 104 of a scan with the generators of the differential driver):
    # the reference's result is exact to 2e-16 at every contrast, the FP64 condensed system is off by 4.5e-9 / 1.7e-7 /
    # 7.5e-5 at 1e8 / 1e10 / 1e12 (a stiff sub-cell whose rotation only the soft neighbours fix) -- three reference-made
    # fixtures that the FP64 body fails and the double-double body must meet
    for decades in (8.0, 10.0, 12.0):
        rng = np.random.default_rng(104)
        g, kind = random_ref_grid(rng)
        while kind >= 4:
            g, kind = random_ref_grid(rng)
        save_case(f"mpsacontrast_cart2d_5x5_1e{decades:.0f}", g, decades, rng)
    # A perturbed triangle grid with a CONTINUOUS random field over 9 decades (seed 400023 of the driver's contrast mode):
    # here the reference's FP64 ASSEMBLY of its gradient systems is what is off -- 4e-9 from the all-mpmath answer, as is the
    # exact inverse of those FP64 systems -- while the double-double regions of the device are exact to 2e-16
    rng = np.random.default_rng(400023)
    g, kind = random_ref_grid(rng)
    while kind >= 4:
        g, kind = random_ref_grid(rng)
    decades = rng.uniform(6.0, 10.0)
    save_case("mpsacontrast_tri2d_seed400023_1e9", g, decades, rng, two_valued=None)
    # ... and a tetrahedral one on which it is the REFERENCE's FP64 inverse that is off (7e-7 at 1e12)
    rng = np.random.default_rng(20261001)
    g = perturb(pp.StructuredTetrahedralGrid([2, 1, 2], [1, 1, 1]), rng, 0.06)
    save_case("mpsacontrast_tet_2x1x2_1e12", g, 12.0, rng)


if __name__ == "__main__":
    main()
