"""Subdomain sharding of the MPFA assembly + Krylov solve: one process per GPU.

The reference has no distributed path; the template for the split is its memory-bounded
sub-problem machinery (numerics/fv/_fvutils.py:414-539 ``subproblems``: partition the cells,
add the cells within one node-ring as overlap, keep only the rows of owned faces).  Here:

* every rank discretizes its owned cells plus one node-ring of halo cells on its own GPU with
  the unchanged single-GPU kernels — **no collective in assembly**: the interaction regions of
  all nodes of owned cells are complete inside the local grid, so the rows of ``A`` (and of the
  flux matrices) belonging to owned cells are exactly the rows of the global matrices;
* the Krylov solve is the only place with a data exchange: before each SpMV the halo entries
  of the input vector are fetched from their owners (point-to-point over xGMI through
  RCCL; ``gloo`` on CPU in the tests) and each reduction is one fused all-reduce of 1-2 doubles.

The Krylov loop itself is the library's fused one (``pfv_solve_sharded``: windowed SpMV on the owned
rows with the dot products fused in, Krylov scalars resident in HBM); this module only serves its
two exchange hooks with ``torch.distributed`` on torch's current stream, which the handle is told
to run on.  torch owns the device buffers the hooks address and the process group - plumbing.  The
same iteration spelled out in torch ops around ``pfv_spmv_device_rows`` is kept as
``solve(driver="torch")``, the cross-check of the tests.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sps

from . import _lib


# ------------------------------------------------------------------------------------------
@dataclass
class LocalProblem:
    """A rank's share: local grid (owned cells numbered first) + global ids."""

    raw: dict                 # raw grid arrays of the local grid (see grid.grid_to_raw)
    n_own: int                # cells [0, n_own) are owned, the rest are halo
    cell_gid: np.ndarray      # global id of every local cell
    halo_owner: np.ndarray    # owner rank of every halo cell (length n_local - n_own)
    face_gid: np.ndarray | None = None      # global id of every local face (None: not tracked)
    artificial_boundary: np.ndarray = field(default_factory=lambda: np.zeros(0, bool))
    # local faces that are single-sided only because the neighbour cell is not in the local grid


def partition_slabs(cell_centers: np.ndarray, nparts: int, axis: int = 2) -> np.ndarray:
    """Equal-count coordinate slabs (cf. ``partition_coordinates`` of the reference,
    grids/partition.py:152)."""
    x = np.asarray(cell_centers)[axis]
    order = np.argsort(x, kind="stable")
    owner = np.empty(x.size, dtype=np.int32)
    owner[order] = (np.arange(x.size) * nparts) // x.size
    return owner


def morton_order(centers: np.ndarray, dim: int, box=None) -> np.ndarray:
    """Indices that sort points along a Morton curve (10 bits per axis).  The local numbering of a rank
    is free; a space-filling curve keeps the x gathers of the SpMV / AMG kernels local (the grid
    generator's numbering - the six tetrahedra of a lattice cell ncells/6 apart - does not).

    ``box = (lo, hi)``: quantise against this box instead of the points' own; with the bounding box of
    the grid's face centres the keys are the ones the library computes (csrc/reorder.inc), so a grid
    numbered this way is recognised there and solved in place, without a renumbered copy of the system."""
    x = np.asarray(centers, dtype=float)[:dim]
    if x.shape[1] == 0:
        return np.zeros(0, dtype=np.int64)
    if box is None:
        lo = x.min(axis=1, keepdims=True)
        ext = np.maximum(x.max(axis=1, keepdims=True) - lo, 1e-300)
        q = np.minimum((x - lo) / ext * 1024.0, 1023.0).astype(np.int64)
    else:
        lo = np.asarray(box[0], dtype=float)[:dim, None]
        ext = np.asarray(box[1], dtype=float)[:dim, None] - lo
        sc = np.where(ext > 0, 1024.0 / np.where(ext > 0, ext, 1.0), 0.0)
        q = np.clip((x - lo) * sc, 0.0, 1023.0).astype(np.int64)
    key = np.zeros(x.shape[1], dtype=np.int64)
    for b in range(10):
        for a in range(dim):
            key |= ((q[a] >> b) & 1) << (dim * b + a)
    return np.argsort(key, kind="stable")


def extract_subdomain(raw: dict, owner: np.ndarray, rank: int) -> LocalProblem:
    """Owned cells + one node-ring of halo cells, renumbered with owned cells first (each group
    along a Morton curve of the cell centres)."""
    nc = raw["cell_centers"].shape[1]
    nf = raw["face_centers"].shape[1]
    nn = raw["nodes"].shape[1]
    cf = sps.csc_matrix((np.ones(raw["cf_indices"].size, dtype=np.int8), raw["cf_indices"], raw["cf_indptr"]),
                        shape=(nf, nc))
    fn = sps.csc_matrix((np.ones(raw["fn_indices"].size, dtype=np.int8), raw["fn_indices"], raw["fn_indptr"]),
                        shape=(nn, nf))
    own = np.flatnonzero(owner == rank)
    cell_nodes = (fn.astype(np.int32) @ cf.astype(np.int32)).tocsc()  # nn x nc
    own_nodes = np.unique(cell_nodes[:, own].indices)
    ring = np.unique(cell_nodes.tocsr()[own_nodes].indices)
    halo = np.setdiff1d(ring, own)
    dim = int(raw["dim"])
    own = own[morton_order(raw["cell_centers"][:, own], dim)]
    halo = halo[morton_order(raw["cell_centers"][:, halo], dim)]
    cells = np.concatenate([own, halo])
    sub_cf = cf[:, cells]
    faces = np.unique(sub_cf.indices)
    sub_fn = fn[:, faces]
    nodes = np.unique(sub_fn.indices)
    fmap = np.full(nf, -1, dtype=np.int64)
    fmap[faces] = np.arange(faces.size)
    nmap = np.full(nn, -1, dtype=np.int64)
    nmap[nodes] = np.arange(nodes.size)
    # local cell_faces (CSC by local cell): keep signs, renumber faces, sort within column
    gsign = sps.csc_matrix((raw["cf_sign"].astype(np.int8), raw["cf_indices"], raw["cf_indptr"]), shape=(nf, nc))
    loc_cf = gsign[:, cells].tocsc()
    loc_cf = sps.csc_matrix((loc_cf.data, fmap[loc_cf.indices], loc_cf.indptr), shape=(faces.size, cells.size))
    loc_cf.sort_indices()
    loc_fn = fn[:, faces].tocsc()
    loc_fn = sps.csc_matrix((loc_fn.data, nmap[loc_fn.indices], loc_fn.indptr), shape=(nodes.size, faces.size))
    loc_fn.sort_indices()
    sides_loc = np.bincount(loc_cf.indices, minlength=faces.size)
    sides_glob = np.bincount(raw["cf_indices"], minlength=nf)[faces]
    lraw = {
        "dim": raw["dim"], "name": raw.get("name", ""),
        "nodes": np.ascontiguousarray(raw["nodes"][:, nodes]),
        "cf_indptr": loc_cf.indptr.astype(np.int32), "cf_indices": loc_cf.indices.astype(np.int32),
        "cf_sign": loc_cf.data.astype(np.int8),
        "fn_indptr": loc_fn.indptr.astype(np.int32), "fn_indices": loc_fn.indices.astype(np.int32),
        "face_normals": np.ascontiguousarray(raw["face_normals"][:, faces]),
        "face_centers": np.ascontiguousarray(raw["face_centers"][:, faces]),
        "cell_centers": np.ascontiguousarray(raw["cell_centers"][:, cells]),
        "face_areas": np.ascontiguousarray(raw["face_areas"][faces]),
        "cell_volumes": np.ascontiguousarray(raw["cell_volumes"][cells]),
    }
    return LocalProblem(raw=lraw, n_own=own.size, cell_gid=cells.astype(np.int64),
                        halo_owner=owner[halo].astype(np.int32), face_gid=faces.astype(np.int64),
                        artificial_boundary=(sides_loc == 1) & (sides_glob == 2))


def permute_cells(raw: dict, order: np.ndarray) -> dict:
    """Renumber the cells of a raw grid: new cell k = old cell order[k]."""
    nf = raw["face_centers"].shape[1]
    nc = raw["cell_centers"].shape[1]
    cf = sps.csc_matrix((raw["cf_sign"].astype(np.int8), raw["cf_indices"], raw["cf_indptr"]), shape=(nf, nc))
    cf = cf[:, order].tocsc()
    cf.sort_indices()
    out = dict(raw)
    out["cf_indptr"] = cf.indptr.astype(np.int32)
    out["cf_indices"] = cf.indices.astype(np.int32)
    out["cf_sign"] = cf.data.astype(np.int8)
    out["cell_centers"] = np.ascontiguousarray(raw["cell_centers"][:, order])
    out["cell_volumes"] = np.ascontiguousarray(raw["cell_volumes"][order])
    return out


# ------------------------------------------------------------------------------------------
class HaloPlan:
    """Who sends which owned entries to whom; built once with one all_gather of index lists."""

    def __init__(self, lp: LocalProblem, dist=None):
        import torch

        self.dist = dist
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1
        halo_gid = lp.cell_gid[lp.n_own:]
        need = {int(q): halo_gid[lp.halo_owner == q] for q in np.unique(lp.halo_owner)}
        if self.world > 1:
            gathered = [None] * self.world
            dist.all_gather_object(gathered, need)
        else:
            gathered = [need]
        own_gid = lp.cell_gid[: lp.n_own]
        sorter = np.argsort(own_gid)
        self.send = {}  # peer -> LongTensor of local owned indices (in the order the peer expects)
        for p, req in enumerate(gathered):
            if p == self.rank or not req or self.rank not in req:
                continue
            g = np.asarray(req[self.rank])
            if g.size == 0:
                continue
            pos = sorter[np.searchsorted(own_gid, g, sorter=sorter)]
            if not np.array_equal(own_gid[pos], g):
                raise RuntimeError("halo request for cells this rank does not own")
            self.send[p] = torch.from_numpy(pos.astype(np.int64))
        self.recv = {}  # peer -> LongTensor of local halo positions, same order as requested
        for q, g in need.items():
            if g.size:
                self.recv[q] = torch.from_numpy((lp.n_own + np.flatnonzero(lp.halo_owner == q)).astype(np.int64))
        self.bytes_per_exchange = 8 * sum(int(v.numel()) for v in self.send.values())
        # the same lists in cells, on the host: the coupled hierarchy derives the plans of its coarse levels from them
        self.send_cells = {p: v.numpy().astype(np.int32) for p, v in self.send.items()}
        self.recv_cells = {q: v.numpy().astype(np.int32) for q, v in self.recv.items()}

    def to(self, device):
        self.send = {k: v.to(device) for k, v in self.send.items()}
        self.recv = {k: v.to(device) for k, v in self.recv.items()}
        return self

    def expand(self, bs: int):
        """Exchange bs consecutive unknowns per cell (cell-major, component-minor numbering)."""
        if bs > 1:
            import torch

            off = torch.arange(bs, dtype=torch.int64)
            self.send = {k: (v[:, None] * bs + off[None, :]).reshape(-1) for k, v in self.send.items()}
            self.recv = {k: (v[:, None] * bs + off[None, :]).reshape(-1) for k, v in self.recv.items()}
            self.bytes_per_exchange *= bs
        return self

    def exchange(self, x_full):
        """Fill the halo entries of x_full (owned entries must be current)."""
        if self.world == 1 or (not self.send and not self.recv):
            return
        import torch

        dist = self.dist
        # gloo cannot send / receive device tensors: stage through the host (validation runs only;
        # the product backend is RCCL, which moves device buffers directly)
        stage = x_full.is_cuda and dist.get_backend() == "gloo"
        buf_dev = "cpu" if stage else x_full.device
        ops, rbufs = [], {}
        for q, pos in self.recv.items():
            rbufs[q] = torch.empty(pos.numel(), dtype=x_full.dtype, device=buf_dev)
            ops.append(dist.P2POp(dist.irecv, rbufs[q], q))
        sbufs = []
        for p, idx in self.send.items():
            sb = x_full.index_select(0, idx).contiguous()
            if stage:
                sb = sb.cpu()
            sbufs.append(sb)
            ops.append(dist.P2POp(dist.isend, sb, p))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        for q, pos in self.recv.items():
            x_full.index_copy_(0, pos, rbufs[q].to(x_full.device) if stage else rbufs[q])

# ------------------------------------------------------------------------------------------
class ShardedMpfa:
    """MPFA assembly of one rank's subdomain + the distributed Jacobi-BiCGStab / CG solve."""

    def __init__(self, lp: LocalProblem, device: str = "cuda", local_device_index: int = 0, library=None,
                 dist=None):
        import torch

        self.torch = torch
        self.lp = lp
        self.device = torch.device(device)
        self.dist = dist
        self.ctx = _lib.Context(local_device_index, library)
        self.ctx.set_grid(lp.raw)
        self.bs = self._dofs_per_cell(lp)
        self.n_own = lp.n_own * self.bs        # unknowns of owned cells
        self.n_loc = lp.raw["cell_centers"].shape[1] * self.bs
        self.plan = HaloPlan(lp, dist).expand(self.bs).to(self.device)
        self._b = self._diag = None
        self._amg_ready = False

    system_matrix = _lib.MAT_SYSTEM

    def _dofs_per_cell(self, lp) -> int:
        return 1

    # boundary flags of the local grid: true boundary faces keep theirs; faces that are one-sided
    # only because the neighbour is outside the local grid are Neumann (they never touch a node of
    # an owned cell, so the value is irrelevant for owned rows)
    def local_bc_flags(self, flags_of_true_boundary: np.ndarray) -> np.ndarray:
        fl = np.asarray(flags_of_true_boundary, dtype=np.uint8).copy()
        fl[self.lp.artificial_boundary] = _lib.BC_NEU
        return fl

    def discretize(self, perm_local, bc_flags_local, robin_local=None, eta=0.0, skip_vector_source=True,
                   rebuild_topology=False):
        """``perm_local``: (3, 3, n_local) numpy array -- or a torch tensor resident on this rank's device: then only
        the permeability is replaced, device to device (pfv_mpfa_set_permeability), and the conditions / eta of the
        last call with host arrays are kept (no PCIe traffic in the step)."""
        torch = self.torch
        if isinstance(perm_local, torch.Tensor):
            if not getattr(self, "_params_set", False):
                raise ValueError("the first discretize call takes host arrays (conditions, eta); later ones may pass a device tensor")
            if tuple(perm_local.shape) != (3, 3, self.n_loc) or perm_local.dtype != torch.float64 or not perm_local.is_contiguous():
                raise ValueError("permeability tensor: contiguous float64 of shape (3, 3, n_local)")
            self._use_torch_stream()
            self.ctx.set_permeability_device(perm_local.data_ptr())
        else:
            self.ctx.set_params(perm_local, bc_flags_local, robin_local, eta)
            self._params_set = True
        self.ctx.discretize(rebuild_topology=rebuild_topology, skip_vector_source=skip_vector_source)

    def assemble(self, bc_values_local, source_local=None):
        """bc values / sources of the local grid: numpy arrays, or torch tensors resident on this
        rank's device (no PCIe traffic in the step)."""
        torch = self.torch
        if isinstance(bc_values_local, torch.Tensor):
            self._use_torch_stream()
            self.ctx.assemble_device(bc_values_local.data_ptr(), 0,
                                     0 if source_local is None else source_local.data_ptr())
        else:
            self.ctx.assemble(bc_values_local, None, source_local)
        self._system_changed()

    def _system_changed(self):
        self._amg_ready = False  # the matrix may have changed
        self._b = self._diag = None

    def _fetch_system(self):
        """rhs and diagonal as torch tensors (only the torch driver needs them)."""
        torch = self.torch
        self._b = torch.empty(self.n_loc, dtype=torch.float64, device=self.device)
        self._diag = torch.empty(self.n_loc, dtype=torch.float64, device=self.device)
        self._use_torch_stream()
        self.ctx.copy_device_vector(0, self._b.data_ptr(), self.n_loc)
        self.ctx.copy_device_vector(1, self._diag.data_ptr(), self.n_loc)

    def _use_torch_stream(self):
        if self.device.type == "cuda":
            self.ctx.set_stream(self.torch.cuda.current_stream(self.device).cuda_stream)

    def _allreduce(self, t):
        if self.dist is not None and self.dist.get_world_size() > 1:
            self.dist.all_reduce(t)
        return t

    # ---- coupled AMG hierarchy (pfv_amg_setup_sharded): the transport of its packed buffers
    def _view(self, ptr: int, nbytes: int):
        """uint8 tensor over nbytes of library-owned memory at ptr (host memory of the emulation build, device memory
        of the HIP build)."""
        import ctypes as C

        torch = self.torch
        if nbytes == 0:
            return torch.empty(0, dtype=torch.uint8, device=self.device)
        if self.device.type == "cpu":
            return torch.from_numpy(np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(ptr)))

        class _Mem:  # (torch wraps foreign device memory through the array interface)
            __cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}

        return torch.as_tensor(_Mem(), device=self.device)

    def _torch_transport(self):
        """sendrecv / allgather of pfv_shard_hooks served by torch.distributed (the gloo / CPU test path, and any group
        the native RCCL transport cannot serve).  Blocking: the data is in place when the hook returns."""
        torch, dist = self.torch, self.dist
        world = dist.get_world_size() if dist is not None else 1
        stage = self.device.type == "cuda" and dist is not None and dist.get_backend() == "gloo"
        failure = self._hook_failures = []
        import weakref

        me = weakref.ref(self)  # (the callbacks are stored on the handle this object owns: no reference cycle through them)
        device = self.device

        def sendrecv(_user, n_peers, peers, d_send, send_ptr, d_recv, recv_ptr, _stream):
            try:
                self = me()
                if device.type == "cuda":
                    self.ctx.sync()
                ops, landing = [], []
                for i in range(n_peers):
                    ns, nr = send_ptr[i + 1] - send_ptr[i], recv_ptr[i + 1] - recv_ptr[i]
                    if nr:
                        rb = self._view(int(d_recv) + 8 * recv_ptr[i], 8 * nr).view(torch.float64)
                        buf = torch.empty(nr, dtype=torch.float64) if stage else rb
                        landing.append((rb, buf))
                        ops.append(dist.P2POp(dist.irecv, buf, int(peers[i])))
                    if ns:
                        sb = self._view(int(d_send) + 8 * send_ptr[i], 8 * ns).view(torch.float64)
                        ops.append(dist.P2POp(dist.isend, sb.cpu() if stage else sb, int(peers[i])))
                if ops:
                    for req in dist.batch_isend_irecv(ops):
                        req.wait()
                for rb, buf in landing:
                    if buf is not rb:
                        rb.copy_(buf)
                if device.type == "cuda":
                    torch.cuda.synchronize(device)
                return 0
            except BaseException as e:  # must not propagate through the C frames
                failure.append(e)
                return 1

        def allgather(_user, d_send, d_recv, nbytes, _stream):
            try:
                self = me()
                if device.type == "cuda":
                    self.ctx.sync()
                src = self._view(int(d_send), nbytes)
                dst = self._view(int(d_recv), nbytes * world)
                if world == 1:
                    dst.copy_(src)
                elif stage:
                    parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
                    dist.all_gather(parts, src.cpu())
                    dst.copy_(torch.cat(parts))
                else:
                    dist.all_gather(list(dst.chunk(world)), src)
                if device.type == "cuda":
                    torch.cuda.synchronize(device)
                return 0
            except BaseException as e:
                failure.append(e)
                return 1

        hooks = _lib.ShardHooks()
        hooks.sendrecv = _lib.SENDRECV_FN(sendrecv)
        hooks.allgather = _lib.ALLGATHER_FN(allgather)
        self._transport_callbacks = (hooks.sendrecv, hooks.allgather)  # keep the trampolines alive
        return hooks

    def amg_setup(self, coupled: bool = True, native=None):
        """Hierarchy of this rank's preconditioner.  coupled (default): the levels keep the couplings to the unknowns of
        other ranks -- halo exchange before the smoothing / residual products, coarse levels gathered and replicated
        from PFV_AMG_GATHER_ROWS rows down (pfv_amg_setup_sharded): the iteration count of the one-process solve.
        coupled = False: block Jacobi across ranks (pfv_amg_setup(n_own)), no communication in the preconditioner,
        more iterations.  A collective when coupled."""
        if not coupled:
            self.ctx.amg_setup(self.n_own)
            self._amg_ready = "block"
            return
        import ctypes as C

        dist = self.dist
        world = dist.get_world_size() if dist is not None else 1
        rank = dist.get_rank() if dist is not None else 0
        if native is not None:
            hooks = _lib.ShardHooks()
            self.ctx._check(self.ctx.lib.pfv_rccl_hooks(native._c, C.byref(hooks)))
        else:
            hooks = self._torch_transport()
        peers = sorted(set(self.plan.send_cells) | set(self.plan.recv_cells))
        empty = np.zeros(0, dtype=np.int32)
        try:
            self.ctx.amg_setup_sharded(self.n_own, hooks, rank, world, peers,
                                       [self.plan.send_cells.get(p, empty) for p in peers],
                                       [self.plan.recv_cells.get(p, empty) for p in peers])
        except _lib.PorefvError as e:
            if native is None and getattr(self, "_hook_failures", None):
                raise self._hook_failures[0] from e  # the transport callback's own exception says why
            raise
        self._amg_ready = "coupled-native" if native is not None else "coupled"

    def _spmv_owned(self, x_full, out_owned):
        self.plan.exchange(x_full)
        self.ctx.spmv_device_rows(self.system_matrix, self.n_own, x_full.data_ptr(), out_owned.data_ptr())

    def solve(self, method: str = "bicgstab", rtol: float = 1e-10, maxit: int = 20000, check_every: int | None = None,
              precond: str = "jacobi", driver: str = "library"):
        """Returns (x_owned as a torch tensor, info).  precond = "amg": one cycle of the coupled aggregation AMG
        (``amg_setup``: every level exchanges its halo values, the coarse levels are gathered and replicated -- the
        iteration count of the one-process solve); precond = "amg_block": every rank applies one V-cycle of the
        hierarchy of its own diagonal block (owned cells x owned cells) -- block Jacobi across ranks, no
        communication inside the preconditioner, more iterations.

        driver = "library" (default): the fused Krylov loop of the C ABI (``pfv_solve_sharded``: windowed
        SpMV with fused dot products, fused vector updates, scalars resident in HBM) calling back here
        for the two exchanges of the distributed method -- the halo entries of each SpMV input
        (point-to-point) and one all-reduce per fused pair of dot products.  driver = "torch": the same
        iteration spelled out in torch ops around ``pfv_spmv_device_rows`` (kept as the cross-check)."""
        if driver == "library":
            # check_every: iterations between two reads of the (all-reduced) residual by the host; None = the library's
            # default (PFV_SHARD_CHECK_EVERY, 4: up to three iterations past convergence, no stream drain in between)
            import os

            saved = os.environ.get("PFV_SHARD_CHECK_EVERY")
            if check_every is not None:
                os.environ["PFV_SHARD_CHECK_EVERY"] = str(int(check_every))
            try:
                return self._solve_library(method, rtol, maxit, precond)
            finally:
                if check_every is not None:
                    if saved is None:
                        os.environ.pop("PFV_SHARD_CHECK_EVERY", None)
                    else:
                        os.environ["PFV_SHARD_CHECK_EVERY"] = saved
        if driver != "torch":
            raise ValueError("driver must be 'library' or 'torch'")
        check_every = 10 if check_every is None else check_every
        torch = self.torch
        n, dev = self.n_own, self.device
        self._use_torch_stream()
        if self._b is None:
            self._fetch_system()
        b = self._b[:n]
        dinv = 1.0 / self._diag[:n]
        f64 = dict(dtype=torch.float64, device=dev)
        if precond in ("amg", "amg_block"):
            want = "block" if precond == "amg_block" else "coupled"
            if self._amg_ready != want:
                self.amg_setup(coupled=precond == "amg")
            check_every = 1

            def apply_M(vec):
                out = torch.empty(n, **f64)
                vec = vec.contiguous()
                self.ctx.amg_apply_device(vec.data_ptr(), out.data_ptr())
                return out
        elif precond == "jacobi":
            def apply_M(vec):
                return vec * dinv
        else:
            raise ValueError("precond must be 'jacobi', 'amg' or 'amg_block'")
        x = torch.zeros(n, **f64)
        r = b.clone()
        full = torch.zeros(self.n_loc, **f64)   # owned + halo staging vector for the SpMV input
        bb = self._allreduce(torch.dot(b, b).reshape(1))
        bbh = float(bb.item())
        info = {"iterations": 0, "converged": False, "rel_residual": 0.0,
                "halo_bytes_per_exchange": self.plan.bytes_per_exchange}
        if bbh <= 0.0:
            info["converged"] = True
            return x, info
        tol2 = rtol * rtol * bbh
        if method == "cg":
            z = apply_M(r)
            p = z.clone()
            v = torch.empty(n, **f64)
            red = self._allreduce(torch.stack((torch.dot(r, r), torch.dot(r, z))))
            rho = red[1]
            for it in range(1, maxit + 1):
                full[:n] = p
                self._spmv_owned(full, v)
                alpha = rho / self._allreduce(torch.dot(p, v).reshape(1))[0]
                x += alpha * p
                r -= alpha * v
                z = apply_M(r)
                red = self._allreduce(torch.stack((torch.dot(r, r), torch.dot(r, z))))
                beta = red[1] / rho
                rho = red[1]
                p = z + beta * p
                info["iterations"] = it
                if it % check_every == 0 or it == maxit:
                    rr = float(red[0].item())
                    info["rel_residual"] = (rr / bbh) ** 0.5
                    if rr <= tol2:
                        info["converged"] = True
                        break
                    if rr != rr:
                        break
            return x, info
        rhat = r.clone()
        p = torch.zeros(n, **f64)
        v = torch.zeros(n, **f64)
        t = torch.empty(n, **f64)
        one = torch.ones((), **f64)
        rho_old, alpha, omega = one.clone(), one.clone(), one.clone()
        red = self._allreduce(torch.stack((torch.dot(r, r), torch.dot(rhat, r))))
        rho = red[1]
        for it in range(1, maxit + 1):
            beta = (rho / rho_old) * (alpha / omega)
            p = r + beta * (p - omega * v)
            y = apply_M(p)
            full[:n] = y
            self._spmv_owned(full, v)
            alpha = rho / self._allreduce(torch.dot(rhat, v).reshape(1))[0]
            s = r - alpha * v
            z = apply_M(s)
            full[:n] = z
            self._spmv_owned(full, t)
            red2 = self._allreduce(torch.stack((torch.dot(t, s), torch.dot(t, t))))
            omega = red2[0] / red2[1]
            x += alpha * y + omega * z
            r = s - omega * t
            rho_old = rho
            red = self._allreduce(torch.stack((torch.dot(r, r), torch.dot(rhat, r))))
            rho = red[1]
            info["iterations"] = it
            if it % check_every == 0 or it == maxit:
                rr = float(red[0].item())
                info["rel_residual"] = (rr / bbh) ** 0.5
                if rr <= tol2:
                    info["converged"] = True
                    break
                if rr != rr:
                    break
        return x, info

    def rccl_transport(self):
        """Native RCCL transport of this rank (created on first use): the unique id is made by rank 0 and
        broadcast through the process group, the halo plan is the one of ``HaloPlan`` (same index lists, same
        order).  None when the group is not RCCL-backed (gloo tests) or the library cannot load librccl."""
        if getattr(self, "_rccl", None) is not None or getattr(self, "_rccl_failed", False):
            return self._rccl
        self._rccl = None
        dist = self.dist
        world = dist.get_world_size() if dist is not None else 1
        rank = dist.get_rank() if dist is not None else 0
        if self.device.type != "cuda" or (dist is not None and dist.get_backend() != "nccl"):
            self._rccl_failed = True
            return None
        torch = self.torch

        def all_agree(ok: bool) -> bool:
            """MIN over the ranks: every rank takes the same branch, and every rank takes part in every collective
            of this function whatever failed locally (a rank that skipped one would leave its peers hanging)."""
            if world == 1:
                return ok
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return bool(int(flag.item()))

        # 1. the unique id: rank 0 ALWAYS broadcasts a payload (ok, id), its own failure included
        payload = [None]
        if rank == 0:
            try:
                payload = [(True, _lib.rccl_unique_id(self.ctx.lib))]
            except _lib.PorefvError as e:  # librccl not loadable
                payload = [(False, str(e))]
        if world > 1:
            dist.broadcast_object_list(payload, src=0)
        have_id = bool(payload[0] and payload[0][0])
        # 2. can every rank load the library?  Asked BEFORE ncclCommInitRank, which is itself a collective: a
        #    rank that cannot take part must be known while the others can still stay out of it
        try:
            loadable = have_id and bool(_lib.rccl_available(self.ctx.lib))
        except _lib.PorefvError:
            loadable = False
        comm, ok = None, False
        if all_agree(loadable):
            try:
                comm = _lib.RcclComm(self.ctx, payload[0][1], rank, world)
                comm.set_halo_plan({p: v.cpu().numpy() for p, v in self.plan.send.items()},
                                   {q: v.cpu().numpy() for q, v in self.plan.recv.items()})
                ok = True
            except _lib.PorefvError:
                ok = False
            # 3. the self-test (a halo exchange and an all-reduce) only if every communicator exists
            if all_agree(ok):
                try:
                    ok = self._rccl_self_test(comm)
                except _lib.PorefvError:
                    ok = False
                ok = all_agree(ok)
            else:
                ok = False
        if ok:
            self._rccl = comm
        else:
            self._rccl_failed = True
            if comm is not None and world == 1:
                comm.close()
        return self._rccl

    def _rccl_self_test(self, comm) -> bool:
        """One halo exchange and one all-reduce through the native hooks on a vector whose entries are their
        own global ids: every halo entry must arrive as the id of the cell it stands for, the reduced sums must
        be the sums over the ranks."""
        import ctypes as C

        torch = self.torch
        hooks = _lib.ShardHooks()
        if self.ctx.lib.pfv_rccl_hooks(comm._c, C.byref(hooks)) != 0:
            return False
        gid = torch.from_numpy(np.repeat(self.lp.cell_gid.astype(np.float64), self.bs) * self.bs +
                               np.tile(np.arange(self.bs, dtype=np.float64), self.lp.cell_gid.size)).to(self.device)
        x = gid.clone()
        x[self.n_own:] = -1.0
        red = torch.tensor([float(self.n_own), 1.0], dtype=torch.float64, device=self.device)
        stream = torch.cuda.current_stream(self.device)
        stream.synchronize()
        sp = C.c_void_p(stream.cuda_stream)
        if hooks.exchange_halo(hooks.user, C.c_void_p(x.data_ptr()), sp) != 0:
            return False
        if hooks.allreduce_sum(hooks.user, C.c_void_p(red.data_ptr()), 2, sp) != 0:
            return False
        stream.synchronize()
        world = self.dist.get_world_size() if self.dist is not None else 1
        if not bool(torch.equal(x, gid)) or float(red[1].item()) != float(world):
            return False
        if self.dist is not None and world > 1:
            tot = torch.tensor([self.n_own], dtype=torch.int64, device=self.device)
            self.dist.all_reduce(tot)
            return float(red[0].item()) == float(tot.item())
        return True

    def _solve_library(self, method, rtol, maxit, precond):
        torch = self.torch
        n, nloc, dev = self.n_own, self.n_loc, self.device
        if precond not in ("jacobi", "amg", "amg_block", "block"):
            raise ValueError("precond must be 'jacobi', 'amg', 'amg_block' or 'block'")
        if precond == "block" and not getattr(self, "_block_layout_set", False):
            raise ValueError("precond='block': set_block_preconditioner(block_ptr over the owned unknowns) first")
        import os

        native = self.rccl_transport() if os.environ.get("PFV_SHARDED_TRANSPORT", "rccl") == "rccl" else None
        want = {"amg": "coupled-native" if native is not None else "coupled", "amg_block": "block"}.get(precond)
        if native is not None:
            # the handle keeps its own stream: nothing of torch takes part in the iteration
            if want and self._amg_ready != want:
                self.amg_setup(coupled=precond == "amg", native=native)
            precond = "amg" if want else precond
            work = torch.empty(2 * nloc + 8, dtype=torch.float64, device=dev)
            x = torch.empty(n, dtype=torch.float64, device=dev)
            torch.cuda.current_stream(dev).synchronize()  # the buffers exist before the handle's stream uses them
            info = self.ctx.solve_sharded(n, native, None, work.data_ptr(), x.data_ptr(),
                                          method=method, rtol=rtol, maxit=maxit, precond=precond)
            self.ctx.sync()
            info["halo_bytes_per_exchange"] = self.plan.bytes_per_exchange
            info["driver"] = "library"
            info["hierarchy"] = want
            info["transport"] = "rccl (native hooks)"
            return x, info
        self._use_torch_stream()
        if want and self._amg_ready != want:
            self.amg_setup(coupled=precond == "amg")
        precond = "amg" if want else precond
        work = torch.empty(2 * nloc + 8, dtype=torch.float64, device=dev)
        x = torch.empty(n, dtype=torch.float64, device=dev)
        views = {work.data_ptr(): work[:nloc], work.data_ptr() + 8 * nloc: work[nloc:2 * nloc]}
        red = work[2 * nloc:]
        red_ptr = work.data_ptr() + 16 * nloc
        multi = self.dist is not None and self.dist.get_world_size() > 1

        def exchange_halo(ptr):
            self.plan.exchange(views[ptr])

        def allreduce_sum(ptr, count):
            if ptr != red_ptr or count > 8:
                raise RuntimeError("unexpected reduction buffer")
            if multi:
                part = red[:count]
                # gloo cannot reduce device tensors in place on every build: stage through the host there
                if red.is_cuda and self.dist.get_backend() == "gloo":
                    h = part.cpu()
                    self.dist.all_reduce(h)
                    part.copy_(h)
                else:
                    self.dist.all_reduce(part)

        try:
            info = self.ctx.solve_sharded(n, exchange_halo, allreduce_sum, work.data_ptr(), x.data_ptr(),
                                          method=method, rtol=rtol, maxit=maxit, precond=precond)
        except _lib.PorefvError as e:
            # a transport callback of the coupled hierarchy failed inside the cycle: its own exception says why
            if getattr(self, "_hook_failures", None):
                raise self._hook_failures[0] from e
            raise
        info["halo_bytes_per_exchange"] = self.plan.bytes_per_exchange
        info["driver"] = "library"
        info["hierarchy"] = want
        info["transport"] = "torch.distributed hooks"
        return x, info

    def owned_system_rows(self):
        """(A rows of owned cells as scipy csr over local columns, b_owned) — for tests."""
        A = self.ctx.matrix(self.system_matrix)
        return A[: self.n_own], self.ctx.active_rhs(self.n_loc)[: self.n_own]


class ShardedMpsa(ShardedMpfa):
    """MPSA assembly of one rank's subdomain + the distributed solve of the elasticity system
    (nd unknowns per cell, cell-major; same halo plan, nd values per exchanged cell; the block
    preconditioner aggregates cells and keeps the components apart)."""

    system_matrix = _lib.MAT_MECH_SYSTEM

    def _dofs_per_cell(self, lp) -> int:
        return int(lp.raw["dim"])

    def local_bc(self, is_dir_true, is_neu_true):
        """Component-wise flags of the local grid: faces that are one-sided only because the
        neighbour cell lies outside the local grid become Neumann (they never touch a node of an
        owned cell)."""
        d, n = np.array(is_dir_true, dtype=bool), np.array(is_neu_true, dtype=bool)
        d[:, self.lp.artificial_boundary] = False
        n[:, self.lp.artificial_boundary] = True
        return d, n

    def discretize(self, stiffness_local, is_dir_local, is_neu_local, eta=0.0, rebuild_topology=False,
                   is_rob_local=None, robin_weight_local=None):
        self.ctx.mpsa_set_params(stiffness_local, self.lp.raw["cell_volumes"], is_dir_local, is_neu_local, eta,
                                 is_rob=is_rob_local, robin_weight=robin_weight_local)
        self.ctx.mpsa_discretize(rebuild_topology=rebuild_topology)

    def assemble(self, bc_values_local, source_local=None):
        self.ctx.mpsa_assemble(bc_values_local, source_local)
        self._system_changed()


class ShardedCsr(ShardedMpfa):
    """Sharded solve of an ASSEMBLED square system (e.g. the coupled Jacobian of a mixed-dimensional model
    from the reference's ``EquationSystem.assemble``: matrix, fracture, intersection and mortar unknowns)
    over the ranks of a process group: every rank keeps the rows of the unknowns it owns (``owner[i]`` = rank
    of unknown i -- by subdomain for a fracture network: the 3-D matrix in slabs, each lower-dimensional
    subdomain and its mortar unknowns with one rank) plus the halo columns those rows touch, and the fused
    Krylov loop of the library (``pfv_solve_sharded``) exchanges the halo entries -- mortar and fracture
    unknowns included -- per SpMV and all-reduces the fused dot products.  Same drivers, transports and block
    preconditioners as the grid-based classes; the reference has no distributed path."""

    system_matrix = _lib.MAT_USER_SYSTEM

    def __init__(self, A, b, owner, device: str = "cuda", local_device_index: int = 0, library=None, dist=None):
        import torch

        self.torch = torch
        self.device = torch.device(device)
        self.dist = dist
        rank = dist.get_rank() if dist is not None else 0
        A = sps.csr_matrix(A)
        n = A.shape[0]
        owner = np.asarray(owner)
        if A.shape[0] != A.shape[1] or owner.shape != (n,) or np.asarray(b).shape != (n,):
            raise ValueError("square matrix, matching right-hand side and one owner per unknown expected")
        own = np.flatnonzero(owner == rank)
        rows = A[own]
        cols = np.unique(rows.indices)
        halo = np.setdiff1d(cols, own)
        halo = halo[np.lexsort((halo, owner[halo]))]          # grouped by owner, ascending global id
        gid = np.concatenate([own, halo]).astype(np.int64)
        n_own, n_loc = own.size, gid.size
        to_local = np.full(n, -1, dtype=np.int64)
        to_local[gid] = np.arange(n_loc)
        top = sps.csr_matrix((rows.data, to_local[rows.indices], rows.indptr), shape=(n_own, n_loc))
        # halo unknowns: identity rows (their values arrive through the exchange, the rows are never used)
        bottom = sps.csr_matrix((np.ones(n_loc - n_own), (np.arange(n_loc - n_own), np.arange(n_own, n_loc))),
                                shape=(n_loc - n_own, n_loc))
        A_loc = sps.vstack([top, bottom], format="csr")
        A_loc.sort_indices()
        b_loc = np.concatenate([np.asarray(b, dtype=float)[own], np.zeros(n_loc - n_own)])
        self.lp = LocalProblem(raw={}, n_own=n_own, cell_gid=gid, halo_owner=owner[halo].astype(np.int32))
        self.ctx = _lib.Context(local_device_index, library)
        self.ctx.set_system(A_loc, b_loc)
        self.bs = 1
        self.n_own, self.n_loc = n_own, n_loc
        self.plan = HaloPlan(self.lp, dist).to(self.device)
        self._b = self._diag = None
        self._amg_ready = False
        self.owned_gid = own

    def set_block_preconditioner(self, block_ptr, gauss_seidel: bool = True):
        """Blocks [block_ptr[k], block_ptr[k+1]) of the OWNED unknowns (local numbering, covering [0, n_own)): the
        sharded solve with ``precond="block"`` sweeps them lower-triangularly with one AMG hierarchy (or exact dense
        inverse) per block; couplings to the unknowns of other ranks are left to the Krylov loop."""
        bp = np.asarray(block_ptr, dtype=np.int64)
        if bp[0] != 0 or bp[-1] != self.n_own:
            raise ValueError("the block layout must cover exactly the owned unknowns")
        self.ctx.set_block_preconditioner(bp, gauss_seidel)
        self._block_layout_set = True

    def discretize(self, *a, **k):
        raise NotImplementedError("ShardedCsr takes an assembled system")

    assemble = discretize
