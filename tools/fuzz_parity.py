"""Randomized differential test of the kernel sources (host-emulation build) against the oracles: random small
grids, random per-face condition types (MPFA: Dirichlet / Neumann / Robin; MPSA: per component), random tensors.
TEST INFRASTRUCTURE (imports oracle/):  python tools/fuzz_parity.py [mode] [n_cases] [first_seed]
modes: core (kernels vs oracles), pieces (partition_arguments vs one piece), subface (conditions per sub-face vs
oracles), update (update_discretization vs a fresh discretization), solve (Krylov + Jacobi / AMG vs a direct solve),
biot (coupling terms vs oracle, pieces vs one piece); all = every mode.  PFV_FUZZ_DEVICE=1 runs the product library on
the GPU instead of the host-emulation build."""
import os
import sys

import numpy as np
import scipy.sparse.linalg as spla

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import porepy_amd as pa  # noqa: E402
from oracle import mpfa_oracle as mo  # noqa: E402
from oracle import mpsa_oracle as so  # noqa: E402
from tests import _parity as P  # noqa: E402


def rel(a, b):
    return abs(a - b).max() / max(abs(b).max(), 1e-300)


def random_grid(rng):
    kind = rng.integers(0, 5)
    if kind == 0:
        g = pa.CartGrid([int(rng.integers(2, 6)), int(rng.integers(2, 6))], [1.0, 1.0])
    elif kind == 1:
        g = pa.StructuredTriangleGrid([int(rng.integers(2, 5)), int(rng.integers(2, 5))], [1.0, 1.0])
    elif kind == 2:
        g = pa.CartGrid([int(rng.integers(2, 4)), int(rng.integers(2, 4)), int(rng.integers(2, 4))], [1.0, 1.0, 1.0])
    elif kind == 3:
        g = pa.StructuredTetrahedralGrid([int(rng.integers(1, 3)), int(rng.integers(1, 3)), int(rng.integers(2, 3))], [1.0, 1.0, 1.0])
    else:
        pts = rng.random((3, int(rng.integers(12, 30))))
        g = pa.TetrahedralGrid(pts)
    g.compute_geometry()
    if kind in (1, 3):
        g = pa.perturb_interior_nodes(g, 0.05 * rng.random(), seed=int(rng.integers(1, 1000)))
    return g, kind


def fuzz_mpfa(lib, rng):
    g, kind = random_grid(rng)
    nc, nf = g.num_cells, g.num_faces
    s = np.exp(rng.standard_normal(nc) * rng.choice([0.0, 0.5, 2.0]))
    kw = dict(kxx=s * (1 + rng.random(nc)), kyy=s * (1 + rng.random(nc)), kxy=s * 0.4 * (rng.random(nc) - 0.5))
    if g.dim == 3:
        kw.update(kzz=s * (1 + rng.random(nc)), kxz=s * 0.3 * (rng.random(nc) - 0.5), kyz=s * 0.3 * (rng.random(nc) - 0.5))
    K = pa.SecondOrderTensor(**kw)
    bf = g.get_all_boundary_faces()
    types = rng.choice(["dir", "neu", "rob"], size=bf.size, p=rng.dirichlet(np.ones(3)))
    types[rng.integers(0, bf.size)] = "dir"
    bc = pa.BoundaryCondition(g, bf, list(types))
    bc.robin_weight = 0.2 + 2 * rng.random(nf)
    raw = pa.grid_to_raw(g)
    eta = float(rng.choice([0.0, 1.0 / 3.0, 0.2])) if kind != 0 and kind != 2 else 0.0
    ctx = pa.Context(0, lib)
    ctx.set_grid(raw)
    ctx.set_params(K.values, pa.bc_flags(bc), np.asarray(bc.robin_weight, float), eta)
    ctx.discretize()
    ora = mo.discretize(raw, K.values, pa.bc_to_raw(bc), eta=eta)
    worst = 0.0
    for i, k in enumerate(mo.MATRIX_KEYS):
        M = ctx.matrix(i)
        assert np.array_equal(M.indptr, ora[k].indptr) and np.array_equal(M.indices, ora[k].indices), (k, "pattern")
        worst = max(worst, rel(M, ora[k]))
    ctx.close()
    return worst, f"mpfa kind {kind} cells {nc} eta {eta:.2f}"


def fuzz_mpsa(lib, rng):
    g, kind = random_grid(rng)
    nd, nc, nf = g.dim, g.num_cells, g.num_faces
    C = pa.FourthOrderTensor(np.exp(rng.standard_normal(nc)) , np.exp(rng.standard_normal(nc)))
    bf = g.get_all_boundary_faces()
    is_dir = np.zeros((nd, nf), bool)
    is_neu = np.zeros((nd, nf), bool)
    is_rob = np.zeros((nd, nf), bool)
    p = rng.dirichlet(np.ones(3))
    for a in range(nd):
        t = rng.choice(3, size=bf.size, p=p)
        is_dir[a, bf[t == 0]] = True
        is_neu[a, bf[t == 1]] = True
        is_rob[a, bf[t == 2]] = True
    f0 = bf[rng.integers(0, bf.size)]
    is_dir[:, f0], is_neu[:, f0], is_rob[:, f0] = True, False, False
    w = 0.3 + rng.random(nf)
    robw = np.einsum("ij,k->ijk", np.eye(nd), w)
    raw = pa.grid_to_raw(g)
    eta = float(rng.choice([0.0, 1.0 / 3.0])) if kind in (1, 3, 4) else 0.0
    ctx = pa.Context(0, lib)
    ctx.set_grid(raw)
    ctx.mpsa_set_params(C.values, g.cell_volumes, is_dir, is_neu, eta, is_rob=is_rob, robin_weight=robw)
    ctx.mpsa_discretize()
    ora = so.discretize(raw, C.values, {"is_dir": is_dir, "is_neu": is_neu, "is_rob": is_rob, "robin_weight": robw},
                        eta=eta)
    worst = 0.0
    for k in P.MPSA_KEYS:
        M = ctx.matrix(P.MPSA_WHICH[k])
        assert np.array_equal(M.indptr, ora[k].indptr) and np.array_equal(M.indices, ora[k].indices), (k, "pattern")
        worst = max(worst, rel(M, ora[k]))
    ctx.close()
    return worst, f"mpsa kind {kind} cells {nc} eta {eta:.2f}"


def _flow_problem(g, rng, p=(0.5, 0.3, 0.2), sigma=0.5):
    nc, nf = g.num_cells, g.num_faces
    s = np.exp(sigma * rng.standard_normal(nc))
    kw = dict(kxx=s * (1 + rng.random(nc)), kyy=s * (1 + rng.random(nc)), kxy=s * 0.3 * (rng.random(nc) - 0.5))
    if g.dim == 3:
        kw.update(kzz=s * (1 + rng.random(nc)), kxz=s * 0.2 * (rng.random(nc) - 0.5), kyz=s * 0.2 * (rng.random(nc) - 0.5))
    K = pa.SecondOrderTensor(**kw)
    bf = g.get_all_boundary_faces()
    types = rng.choice(["dir", "neu", "rob"], size=bf.size, p=list(p))
    types[:2] = "dir"
    bc = pa.BoundaryCondition(g, bf, list(types))
    bc.robin_weight = 0.2 + 2 * rng.random(nf)
    bv = np.zeros(nf)
    bv[bf] = rng.random(bf.size)
    return K, bc, bv


def _mech_bc(g, rng):
    nd = g.dim
    bc = pa.BoundaryConditionVectorial(g)
    bf = g.get_all_boundary_faces()
    for a in range(nd):
        t = rng.random(bf.size) < 0.5
        bc.is_dir[a, bf[t]] = True
        bc.is_neu[a, bf[t]] = False
    bc.is_dir[:, bf[:2]] = True
    bc.is_neu[:, bf[:2]] = False
    return bc


def fuzz_pieces(lib, rng):
    g, kind = random_grid(rng)
    if kind == 4 or g.num_cells < 8:
        return 0.0, "skipped"
    nd, nc, nf = g.dim, g.num_cells, g.num_faces
    K, bc, bv = _flow_problem(g, rng)
    k = int(rng.integers(2, 5))
    P.split_matches_one_piece(lib, g, K, bc, bv, dict(partition_arguments={"num_subproblems": k}))
    C = pa.FourthOrderTensor(np.exp(0.5 * rng.standard_normal(nc)), np.exp(0.5 * rng.standard_normal(nc)))
    bcv = _mech_bc(g, rng)
    bvv = (rng.random((nd, nf)) - 0.5) * (bcv.is_dir | bcv.is_neu)
    P.mpsa_split_matches_one_piece(lib, g, C, bcv, bvv.ravel("F"), dict(partition_arguments={"num_subproblems": k}),
                                   source=0.01 * rng.standard_normal(nd * nc))
    return 0.0, f"pieces kind {kind} cells {nc} parts {k}"


def fuzz_subface(lib, rng):
    g, kind = random_grid(rng)
    if kind == 4:
        return 0.0, "skipped"
    raw = pa.grid_to_raw(g)
    nd, nc, nf = g.dim, g.num_cells, g.num_faces
    nsub = raw["fn_indices"].size
    face_of_sub = np.repeat(np.arange(nf), np.diff(raw["fn_indptr"]))
    isb = np.isin(face_of_sub, g.get_all_boundary_faces())
    first = np.flatnonzero(isb)[0]
    eta = 0.0 if kind in (0, 2) else 1.0 / 3.0
    worst = 0.0
    # MPFA
    K, _, _ = _flow_problem(g, rng)
    t = rng.choice(3, size=nsub, p=[0.5, 0.3, 0.2])
    is_dir, is_neu, is_rob = isb & (t == 0), isb & (t == 1), isb & (t == 2)
    is_dir[first], is_neu[first], is_rob[first] = True, False, False
    rw = 0.2 + rng.random(nsub)
    ora = mo.discretize(raw, K.values, {"is_dir": is_dir, "is_neu": is_neu, "is_rob": is_rob,
                                        "is_internal": np.zeros(nsub, bool), "robin_weight": rw}, eta=eta)
    ctx = pa.Context(0, lib)
    ctx.set_grid(raw)
    ctx.set_params(K.values, np.full(nf, pa._lib.BC_NEU, np.uint8), np.ones(nf), eta)
    ctx.set_subface_bc((is_dir * pa._lib.BC_DIR + is_neu * pa._lib.BC_NEU + is_rob * pa._lib.BC_ROB).astype(np.uint8), rw)
    ctx.discretize()
    for i, k in enumerate(mo.MATRIX_KEYS):
        M = ctx.matrix(i)
        assert M.shape == ora[k].shape, (k, M.shape, ora[k].shape)
        worst = max(worst, rel(M, ora[k]))
    ctx.close()
    # MPSA
    sd_, sn_, sr_ = np.zeros((nd, nsub), bool), np.zeros((nd, nsub), bool), np.zeros((nd, nsub), bool)
    for a in range(nd):
        t = rng.choice(3, size=nsub, p=[0.5, 0.35, 0.15])
        sd_[a], sn_[a], sr_[a] = isb & (t == 0), isb & (t == 1), isb & (t == 2)
    sd_[:, first], sn_[:, first], sr_[:, first] = True, False, False
    robw = np.einsum("ij,k->ijk", np.eye(nd), 0.3 + rng.random(nsub))
    C = pa.FourthOrderTensor(np.exp(0.4 * rng.standard_normal(nc)), np.exp(0.4 * rng.standard_normal(nc)))
    oras = so.discretize(raw, C.values, {"is_dir": sd_, "is_neu": sn_, "is_rob": sr_, "robin_weight": robw}, eta=eta)
    ctx = pa.Context(0, lib)
    ctx.set_grid(raw)
    ctx.mpsa_set_params(C.values, g.cell_volumes, np.zeros((nd, nf), bool), np.ones((nd, nf), bool), eta)
    ctx.mpsa_set_subface_bc(sd_, sn_, sr_, robw)
    ctx.mpsa_discretize()
    for k in P.MPSA_KEYS:
        M = ctx.matrix(P.MPSA_WHICH[k])
        assert M.shape == oras[k].shape and np.array_equal(M.indices, oras[k].indices), (k, "shape / pattern")
        worst = max(worst, rel(M, oras[k]))
    ctx.close()
    return worst, f"subface kind {kind} cells {nc}"


def fuzz_update(lib, rng):
    g, kind = random_grid(rng)
    if kind == 4 or g.num_cells < 8:
        return 0.0, "skipped"
    nd, nc, nf = g.dim, g.num_cells, g.num_faces
    bf = g.get_all_boundary_faces()
    mod = np.unique(rng.integers(0, nc, size=int(rng.integers(1, max(2, nc // 4)))))

    def Kof(s):
        kw = dict(kxx=s * 1.5, kyy=s * 2.0, kxy=s * 0.2)
        if nd == 3:
            kw.update(kzz=s * 0.7, kxz=s * 0.1, kyz=s * 0.05)
        return pa.SecondOrderTensor(**kw)

    s0 = np.exp(0.3 * rng.standard_normal(nc))
    s1 = s0.copy()
    s1[mod] *= 3.0
    _, bc, bv = _flow_problem(g, rng)
    data = pa.initialize_data({}, "flow", {"second_order_tensor": Kof(s0), "bc": bc, "bc_values": bv})
    d = pa.Mpfa("flow", library=lib)
    d.discretize(g, data)
    data[pa.PARAMETERS]["flow"]["second_order_tensor"] = Kof(s1)
    data["update_discretization"] = {"modified_cells": mod}
    d.update_discretization(g, data)
    fresh = pa.initialize_data({}, "flow", {"second_order_tensor": Kof(s1), "bc": bc, "bc_values": bv})
    pa.Mpfa("flow", library=lib).discretize(g, fresh)
    m1, m2 = data[pa.DISCRETIZATION_MATRICES]["flow"], fresh[pa.DISCRETIZATION_MATRICES]["flow"]
    worst = max(rel(m1[k], m2[k]) for k in m2)
    mu0 = np.exp(0.3 * rng.standard_normal(nc))
    mu1 = mu0.copy()
    mu1[mod] *= 2.5
    bcv = _mech_bc(g, rng)
    data = pa.initialize_data({}, "mechanics", {"fourth_order_tensor": pa.FourthOrderTensor(mu0, mu0), "bc": bcv})
    d = pa.Mpsa("mechanics", library=lib)
    d.discretize(g, data)
    data[pa.PARAMETERS]["mechanics"]["fourth_order_tensor"] = pa.FourthOrderTensor(mu1, mu1)
    data["update_discretization"] = {"modified_cells": mod}
    d.update_discretization(g, data)
    fresh = pa.initialize_data({}, "mechanics", {"fourth_order_tensor": pa.FourthOrderTensor(mu1, mu1), "bc": bcv})
    pa.Mpsa("mechanics", library=lib).discretize(g, fresh)
    m1, m2 = data[pa.DISCRETIZATION_MATRICES]["mechanics"], fresh[pa.DISCRETIZATION_MATRICES]["mechanics"]
    worst = max(worst, max(rel(m1[k], m2[k]) for k in m2))
    return worst, f"update kind {kind} cells {nc} modified {mod.size}"


def fuzz_solve(lib, rng):
    g, kind = random_grid(rng)
    nc = g.num_cells
    K, bc, bv = _flow_problem(g, rng, p=(0.4, 0.4, 0.2), sigma=float(rng.choice([0.0, 0.5, 1.5])))
    data = pa.initialize_data({}, "flow", {"second_order_tensor": K, "bc": bc, "bc_values": bv})
    d = pa.Mpfa("flow", library=lib)
    d.discretize(g, data)
    A, b = d.assemble_matrix_rhs(g, data)
    src = rng.random(nc) * g.cell_volumes
    xo = spla.spsolve(A.tocsc(), b + src)
    worst = 0.0
    for pre in ("jacobi", "amg"):
        for method in ("bicgstab", "gmres"):
            x, info = d.solve(g, data, source=src, method=method, rtol=1e-11, precond=pre, maxit=5000,
                              restart=30 if method == "gmres" else 0)
            err = float(np.linalg.norm(x - xo) / np.linalg.norm(xo))
            assert info["converged"] and err <= 1e-7, (pre, method, info, err)  # rtol 1e-11 on the residual
            worst = max(worst, err)
    return 0.0, f"solve kind {kind} cells {nc} (worst error {worst:.1e})"


def fuzz_biot(lib, rng):
    g, kind = random_grid(rng)
    if kind == 4 or g.num_cells < 6:
        return 0.0, "skipped"
    raw = pa.grid_to_raw(g)
    nc = g.num_cells
    bc = _mech_bc(g, rng)
    C = pa.FourthOrderTensor(np.exp(0.4 * rng.standard_normal(nc)), np.exp(0.4 * rng.standard_normal(nc)))
    a1 = np.zeros((3, 3, nc))
    for i in range(3):
        a1[i, i] = 0.5 + rng.random(nc)
    a2 = a1.copy()
    a2[0, 1] = a2[1, 0] = 0.2 * rng.random(nc)
    alphas = {"p": a1, "T": a2}
    maps = {k: type("A", (), {"values": v})() for k, v in alphas.items()}

    def run(**extra):
        data = pa.initialize_data({}, "mechanics", {"fourth_order_tensor": C, "bc": bc, "scalar_vector_mappings": maps, **extra})
        pa.Biot("mechanics", library=lib).discretize(g, data)
        return data[pa.DISCRETIZATION_MATRICES]["mechanics"]

    one = run()
    ora = so.discretize(raw, C.values, {"is_dir": bc.is_dir, "is_neu": bc.is_neu}, alphas=alphas)
    worst = max(rel(one[k][key], ora[k][key]) for k in so.BIOT_KEYS for key in alphas)
    worst = max(worst, max(rel(one[k], ora[k]) for k in P.MPSA_KEYS))
    many = run(partition_arguments={"num_subproblems": int(rng.integers(2, 4))})
    worst = max(worst, max(rel(many[k][key], one[k][key]) for k in so.BIOT_KEYS for key in alphas))
    worst = max(worst, max(rel(many[k], one[k]) for k in P.MPSA_KEYS))
    return worst, f"biot kind {kind} cells {nc}"


MODES = {"core": (fuzz_mpfa, fuzz_mpsa), "pieces": (fuzz_pieces,), "subface": (fuzz_subface,), "update": (fuzz_update,),
         "solve": (fuzz_solve,), "biot": (fuzz_biot,)}


def run_mode(lib, mode: str, n: int, seed0: int, verbose: bool = True) -> int:
    """Number of suspicious cases (error above 1e-8, or an exception other than a singular random input)."""
    fns = sum((MODES[m] for m in (MODES if mode == "all" else [mode])), ())
    bad = 0
    for i in range(n):
        for fn in fns:
            rng = np.random.default_rng([seed0 + i, sum(map(ord, fn.__name__))])
            try:
                err, what = fn(lib, rng)
                bad += err >= 1e-8
                if verbose:
                    print(f"seed {seed0 + i:4d} {what:44s} max rel err {err:.2e}{'   <-- LARGE' if err >= 1e-8 else ''}", flush=True)
            except ValueError as e:  # singular local systems are a property of the random input (both sides raise)
                if verbose:
                    print(f"seed {seed0 + i:4d} {fn.__name__}: {type(e).__name__} {str(e)[:80]}", flush=True)
            except pa._lib.PorefvError as e:
                if e.status != 1:
                    bad += 1
                if verbose:
                    print(f"seed {seed0 + i:4d} {fn.__name__}: status {e.status} {str(e)[:100]}", flush=True)
            except Exception as e:
                bad += 1
                print(f"seed {seed0 + i:4d} {fn.__name__}: FAILED {type(e).__name__} {str(e)[:200]}", flush=True)
    return bad


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "core"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    # PFV_FUZZ_DEVICE=1: the gfx950 product library on the GPU instead of the host-emulation build
    lib = None if os.environ.get("PFV_FUZZ_DEVICE") else P.emulation_library()
    print("suspicious cases:", run_mode(lib, mode, n, seed0))


if __name__ == "__main__":
    main()
