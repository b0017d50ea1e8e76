#!/usr/bin/env python
"""bench.py — cells/sec of MPFA assemble + solve on synthetic 3-D simplex box grids.

One "step" = one pass of the hot path over one grid already resident in HBM:
  sub-cell topology + CSR symbolic phase + interaction-region kernel + face kernel
  (= what one ``Mpfa.discretize`` call of the reference does), ``A = div @ flux`` and the
  right-hand side, then the Jacobi-preconditioned BiCGStab solve to rtol.
Default workload (N = 1): BASELINE.json configs[2] — ~2 M tetrahedra, perturbed nodes,
full-tensor anisotropic permeability — the grid the north-star target is quoted on.

Launch: ``python bench.py --gpus N --steps K --warmup W``; for N > 1 under
``python -m torch.distributed.run --nproc-per-node N``.  The path shards by subdomain: each
rank owns one box grid of the same size (weak scaling); assembly needs no collective.  Until
the halo-exchanging solve lands every rank also solves its own subdomain
system, so N > 1 runs are replicas of the single-GPU step (stated in ``config``).

Prints ONE JSON line on rank 0 (contract in the task statement) with ``roofline`` for the
dominant kernel (CSR SpMV of the solve; live HIP-event timing through pfv_time_kernel) and
``cpu_baseline`` = the CPU oracle (``oracle/mpfa_oracle.py`` + scipy direct solve) timed on a
bounded sample of the same workload on this box's host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def make_problem(n_side: int, seed: int = 1):
    import porepy_amd as pa

    g = pa.StructuredTetrahedralGrid([n_side] * 3, [1.0, 1.0, 1.0])
    g.compute_geometry()
    g = pa.perturb_interior_nodes(g, 0.2 / n_side, seed=seed)
    nc = g.num_cells
    rng = np.random.default_rng(2)
    scale = np.exp(0.5 * rng.standard_normal(nc))  # mild cell-wise heterogeneity
    K = pa.SecondOrderTensor(kxx=1.0 * scale, kyy=10.0 * scale, kzz=0.1 * scale, kxy=0.5 * scale,
                             kxz=0.05 * scale, kyz=0.2 * scale)
    bf = g.get_all_boundary_faces()
    xf = g.face_centers[0, bf]
    dirf = bf[(xf < 1e-9) | (xf > 1 - 1e-9)]
    bc = pa.BoundaryCondition(g, dirf, ["dir"] * dirf.size)
    bv = np.zeros(g.num_faces)
    bv[dirf] = g.face_centers[0, dirf]
    return g, K, bc, bv, g.cell_volumes.copy()


def cpu_baseline(n_side: int):
    """Oracle (numpy node loop restating the reference's algorithm) + scipy direct solve on a
    bounded sample: same grid family / tensor / BCs at n_side^3*6 cells, 1 thread."""
    import scipy.sparse.linalg as spla

    import porepy_amd as pa
    from oracle import mpfa_oracle as mo

    g, K, bc, bv, src = make_problem(n_side)
    raw = pa.grid_to_raw(g)
    t0 = time.perf_counter()
    mats = mo.discretize(raw, K.values, pa.bc_to_raw(bc))
    A, b = mo.assemble_matrix_rhs(raw, mats, bv)
    t1 = time.perf_counter()
    x = spla.spsolve(A.tocsc(), b + src)
    t2 = time.perf_counter()
    return {
        "value": g.num_cells / (t2 - t0),
        "unit": "cells/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{g.num_cells} tetrahedra (n_side={n_side}) of the same perturbed anisotropic box: "
                  f"oracle discretize+assemble {t1 - t0:.1f} s, scipy spsolve {t2 - t1:.1f} s; "
                  f"host has {os.cpu_count()} cores, path is single-threaded",
        "check_norm": float(np.linalg.norm(x)),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n-side", type=int, default=69, help="lattice cells per side (6 tets each)")
    ap.add_argument("--rtol", type=float, default=1e-10)
    ap.add_argument("--cpu-n-side", type=int, default=12)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--phases", action="store_true", help="also print per-phase timings to stderr")
    args = ap.parse_args()

    import torch

    import porepy_amd as pa

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N > 1 with python -m torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    g, K, bc, bv, src = make_problem(args.n_side, seed=1 + rank)
    ctx = pa.Context(local_rank)
    ctx.set_grid(pa.grid_to_raw(g))
    ctx.set_params(K.values, pa.bc_flags(bc), bc.robin_weight, pa.determine_eta(g))
    nc = g.num_cells

    def step():
        ctx.discretize(rebuild_topology=True)
        ctx.assemble(bv, None, src)
        x, info = ctx.solve("bicgstab", rtol=args.rtol, maxit=20000, raise_on_fail=False)
        return x, info

    for _ in range(args.warmup):
        x, info = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x, info = step()
    ctx.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = ctx.stats()
    ms_per_step = 1e3 * elapsed / args.steps
    value = nc * world * args.steps / elapsed

    # ---- roofline of the dominant kernel: CSR SpMV with A (2 per BiCGStab iteration) ----
    _, _, nnzA = ctx.matrix_info(pa._lib.MAT_SYSTEM)
    spmv_ms = ctx.time_kernel(0, reps=50)
    spmv_bytes = 12.0 * nnzA + 4.0 * (nc + 1) + 8.0 * nc + 8.0 * nc  # values+indices, indptr, x, y
    achieved = spmv_bytes / (spmv_ms * 1e-3) / 1e9
    roofline = {"bound": "hbm", "kernel": "k_spmv (CSR SpMV with A, 2 launches per BiCGStab iteration)",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": None, "bytes_per_launch": spmv_bytes, "ms_per_launch": spmv_ms}
    # assembly kernels (HBM-bound on their CSR output): algorithmic bytes = inputs once + outputs once
    nnz = {k: ctx.matrix_info(i)[2] for i, k in enumerate(("flux", "bound_flux", "bpc", "bpf", "vs", "bpvs"))}
    out_bytes = 8.0 * sum(nnz.values()) + 4.0 * (nnz["flux"] + nnz["bound_flux"] + nnz["vs"]) + 12.0 * nnzA
    in_bytes = 8.0 * (3 * g.num_nodes + 3 * nc + 7 * g.num_faces) + 72.0 * nc + 5.0 * 4 * nc + 4.0 * 3 * g.num_faces
    asm_ms = st["topology_ms"] + st["symbolic_ms"] + st["node_ms"] + st["face_ms"] + st["assemble_ms"]
    assembly = {"algorithmic_bytes": out_bytes + in_bytes, "ms": asm_ms,
                "achieved_GBs": (out_bytes + in_bytes) / (asm_ms * 1e-3) / 1e9,
                "frac_of_hbm_peak": (out_bytes + in_bytes) / (asm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "phases_ms": {k: st[k] for k in ("topology_ms", "symbolic_ms", "node_ms", "face_ms",
                                                 "assemble_ms", "solve_ms")},
                "cells_per_s_assembly_only": nc / (asm_ms * 1e-3)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.cpu_n_side)

    if rank == 0:
        res_true = None
        try:
            A = ctx.matrix(pa._lib.MAT_SYSTEM)
            b = ctx.rhs()
            res_true = float(np.linalg.norm(b - A @ x) / np.linalg.norm(b))
        except Exception:
            pass
        line = {
            "metric": "cells/sec MPFA assemble+solve, 3D unstructured grid",
            "value": value, "unit": "cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"3D simplex box, {nc} tetrahedra per GPU (n_side={args.n_side}), perturbed "
                                   "nodes, full-tensor anisotropic heterogeneous K, Dirichlet x-faces; "
                                   "MPFA-O discretize (6 matrices) + div@flux + Jacobi-BiCGStab",
                       "cells_per_gpu": nc, "krylov": "bicgstab+jacobi", "rtol": args.rtol,
                       "iterations": info["iterations"], "converged": info["converged"],
                       "true_rel_residual": res_true,
                       "parallelism": "1 GPU" if world == 1 else f"{world} subdomain replicas, no halo exchange yet"},
            "roofline": roofline, "assembly": assembly, "cpu_baseline": cpu,
        }
        if args.phases:
            print(json.dumps(st, indent=1), file=sys.stderr)
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
