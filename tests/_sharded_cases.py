"""Workers of the sharded-solve comparisons (gloo, host emulation build): shared by tests/test_distributed_gloo.py
and tools/coupled_amg_gloo.py."""
import json
import os
import socket
import time

import numpy as np


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def problem(n_side):
    import porepy_amd as pa

    g = pa.StructuredTetrahedralGrid([n_side, n_side, 2 * n_side], [1, 1, 2.0])
    g.compute_geometry()
    g = pa.perturb_interior_nodes(g, 0.04, seed=3)
    nc = g.num_cells
    rng = np.random.default_rng(5)
    f = np.exp(1.0 * rng.standard_normal(nc))
    K = pa.SecondOrderTensor(kxx=f * (1 + rng.random(nc)), kyy=f * (3 + rng.random(nc)), kzz=f * (0.5 + rng.random(nc)),
                             kxy=f * 0.3 * rng.random(nc), kxz=f * 0.1 * rng.random(nc), kyz=f * 0.1 * rng.random(nc))
    bf = g.get_all_boundary_faces()
    dirf = bf[(g.face_centers[0, bf] < 1e-9) | (g.face_centers[0, bf] > 1 - 1e-9)]
    bc = pa.BoundaryCondition(g, dirf, ["dir"] * dirf.size)
    bv = np.zeros(g.num_faces)
    bv[dirf] = 1 + g.face_centers[1, dirf]
    src = g.cell_volumes * (1 + rng.random(nc))
    return g, K, bc, bv, src


def worker(rank, world, port, n_side, out, env=None):
    import torch
    import torch.distributed as dist

    import porepy_amd as pa
    from porepy_amd import distributed as D
    from tests import _parity as P

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.update(env or {})
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lib = P.emulation_library()
        g, K, bc, bv, src = problem(n_side)
        raw = pa.grid_to_raw(g)
        owner = D.partition_slabs(raw["cell_centers"], world, axis=2)
        lp = D.extract_subdomain(raw, owner, rank)
        sh = D.ShardedMpfa(lp, device="cpu", library=lib, dist=dist)
        flags = sh.local_bc_flags(pa.bc_flags(bc)[lp.face_gid])
        sh.discretize(K.values[:, :, lp.cell_gid], flags, bc.robin_weight[lp.face_gid], pa.determine_eta(g))
        sh.assemble(bv[lp.face_gid], src[lp.cell_gid])
        res = {}
        for precond in ("amg", "amg_block"):
            t0 = time.time()
            x, info = sh.solve(method="bicgstab", rtol=1e-10, maxit=500, precond=precond, check_every=1)
            st = sh.ctx.stats()
            res[precond] = {"iterations": info["iterations"], "converged": info["converged"],
                            "rel_residual": info["rel_residual"], "seconds": time.time() - t0,
                            "levels": int(st["amg_levels"]), "coarsest_rows": int(st["amg_coarsest_rows"])}
            np.save(os.path.join(out, f"x_{precond}_{rank}.npy"), np.stack([lp.cell_gid[: lp.n_own], x.numpy()]))
        if rank == 0:
            json.dump(res, open(os.path.join(out, "res.json"), "w"))
    finally:
        dist.destroy_process_group()




def run(world, n_side, out, env=None):
    """Spawn `world` ranks; returns {precond: {..., "x": global solution}}."""
    import torch.multiprocessing as mp

    from tests import _parity as P

    P.emulation_library()  # build once here, not concurrently in the workers
    mp.spawn(worker, args=(world, _free_port(), n_side, out, env), nprocs=world, join=True)
    res = json.load(open(os.path.join(out, "res.json")))
    for precond in res:
        parts = [np.load(os.path.join(out, f"x_{precond}_{r}.npy")) for r in range(world)]
        gid = np.concatenate([p[0] for p in parts]).astype(int)
        x = np.empty(gid.size)
        x[gid] = np.concatenate([p[1] for p in parts])
        res[precond]["x"] = x
    return res
