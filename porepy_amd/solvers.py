"""Device linear solve for assembled PorePy systems — the caller one level above the
discretization: ``SolutionStrategy.solve_linear_system``
(models/solution_strategy.py:830-884) holds the global Jacobian as a scipy CSR matrix and the
residual as a numpy vector and hands them to a direct solver; :class:`HipLinearSolver` routes
them to the Jacobi-preconditioned Krylov solvers behind ``pfv_set_system`` / ``pfv_solve``.

    class Model(porepy_amd.HipLinearSolver, SinglePhaseFlow): ...
    params = {"linear_solver": "hip_bicgstab", ...}      # or "hip_gmres", "hip_cg"
    params["hip_solver_options"] = {"precond": "amg", "rtol": 1e-12}   # optional

Any other ``linear_solver`` value falls through to the reference implementation.

Coupled Jacobians of mixed-dimensional / multi-physics models (mortar fluxes, several variables per cell: the
systems the reference hands to a direct solver) go through :func:`solve_block_system`: the rows are first paired with
the unknowns by a maximum-product matching (the equations of such a model are ordered differently from its unknowns:
most diagonal entries are structurally zero), the unknowns grouped in one block per (variable, subdomain / interface),
and GMRES runs on the device with the block lower-triangular preconditioner of ``pfv_set_block_preconditioner`` --
every diagonal block solved by its own AMG hierarchy (small blocks exactly).  ``HipLinearSolver`` takes this path
with ``hip_solver_options = {"precond": "block"}``, the blocks read off the model's ``equation_system``.
"""
from __future__ import annotations

import numpy as np

from . import _lib

_METHODS = {"hip_bicgstab": "bicgstab", "hip_gmres": "gmres", "hip_cg": "cg"}


def solve_csr(A, b, method: str = "bicgstab", rtol: float = 1e-12, maxit: int = 50000, restart: int = 0,
              device: int = 0, library=None, context: _lib.Context | None = None, precond: str = "jacobi", x0=None):
    """x with ||b - A x|| <= rtol ||b||, computed on the device; returns (x, info).  ``A``: scipy sparse, or a
    ``DeviceCsr`` (solved where it lives).  ``context``: a handle to reuse (its buffers, streams and memory pool)
    instead of a fresh one per call; ``x0``: initial guess."""
    from .device_csr import DeviceCsr

    if isinstance(A, DeviceCsr):
        # a system assembled on the device (device_csr.py): it becomes the active system of the solving handle without
        # a host copy (pfv_csr_set_system)
        ctx = A.as_system(b, context=context)
    else:
        ctx = context if context is not None else _lib.Context(device, library)
        ctx.set_system(A, b)
    return ctx.solve(method=method, rtol=rtol, maxit=maxit, x0=x0, restart=restart, n=A.shape[0], precond=precond)


def match_rows(A):
    """Row permutation ``perm`` such that ``A[perm]`` has a zero-free, large diagonal: a maximum-product perfect matching
    of rows to columns (minimum-weight bipartite matching on ``-log |a_ij|``, scipy.sparse.csgraph).  The role of MC64
    in direct solvers; host work, O(nnz log n).  Returns None when the diagonal is already structurally full and
    dominant in every row that has one."""
    import scipy.sparse as sps
    from scipy.sparse.csgraph import min_weight_full_bipartite_matching

    A = sps.csr_matrix(A)
    n = A.shape[0]
    d = np.abs(A.diagonal())
    rowmax = np.maximum.reduceat(np.abs(A.data), A.indptr[:-1]) if A.nnz else np.zeros(n)
    if np.all(d > 0) and np.all(d >= 0.1 * rowmax):
        return None
    B = sps.csr_matrix((np.abs(A.data), A.indices, A.indptr), shape=A.shape)
    B.eliminate_zeros()
    # weights > 0 for every stored entry: log(colmax / |a_ij|) + 1
    colmax = np.asarray(abs(B).max(axis=0).todense()).ravel()
    colmax[colmax == 0] = 1.0
    W = B.copy()
    W.data = np.log(colmax[B.indices] / B.data) + 1.0
    rows, cols = min_weight_full_bipartite_matching(W)
    perm = np.empty(n, dtype=np.int64)
    perm[cols] = rows  # row perm[j] is matched to column j
    return perm


def solve_block_system(A, b, block_of, method: str = "gmres", rtol: float = 1e-12, maxit: int = 2000, restart: int = 60,
                       device: int = 0, library=None, context: _lib.Context | None = None, gauss_seidel: bool = True,
                       block_order=None, row_perm=None, eliminate=None):
    """Solve a coupled system whose unknown ``i`` belongs to block ``block_of[i]`` (one block per variable and
    subdomain / interface) on the device: rows matched to unknowns (:func:`match_rows`), blocks made contiguous in
    ``block_order`` (default: ascending block number), GMRES with the block lower-triangular preconditioner.  Returns
    (x, info) with x in the caller's numbering; ``info["true_rel_residual"]`` is evaluated on the caller's system.
    ``row_perm``: the pairing of equations with unknowns if the caller knows it (row ``row_perm[j]`` is the equation of
    unknown ``j``; :func:`pair_equation_blocks` derives it from the block structure of a model) -- an entry-wise
    matching can pair a pressure unknown with an energy equation whose advective term happens to be large, and the
    diagonal block of that variable is then not the discretization of anything.
    ``eliminate``: boolean mask of unknowns E to condense before the Krylov loop -- the interface fluxes of a
    mixed-dimensional model, whose own equations are (nearly) diagonal in them (the interface law
    ``lambda + kappa (tr p_h - p_l) = 0``: models/constitutive_laws.py:987-1000) but which couple the pressures of a
    matrix cell and a fracture cell as strongly as a face of either grid does.  No block triangular sweep over
    (pressure, flux) blocks sees that coupling (52 fractures: GMRES stalls at 1e-6 with 1535 blocks, and with exact
    solves of five variable-wide blocks too).  With K the kept unknowns and D = diag(A_EE), the system is multiplied
    from the left by L = I - A_KE D^-1 (rows K, columns E) ON THE DEVICE (one sparse product + one sum of
    ``DeviceCsr``): the K rows of L A hold the Schur complement S = A_KK - A_KE D^-1 A_EK, a diagonally dominant
    M-matrix for pressures and temperatures, and what is left of A_KE is A_KE (I - D^-1 A_EE): zero where A_EE is
    diagonal.  The blocks of E are moved behind those of K, so that the lower-triangular sweep solves S first and
    recovers E from it.  The solution is that of the caller's system (L is regular); the loop stops on the residual of
    L A x = L b, ``info["true_rel_residual"]`` is that of A x = b."""
    import scipy.sparse as sps

    A = sps.csr_matrix(A)
    b = np.asarray(b, dtype=float)
    n = A.shape[0]
    block_of = np.asarray(block_of)
    perm = match_rows(A) if row_perm is None else np.asarray(row_perm, dtype=np.int64)
    if perm is not None and np.array_equal(perm, np.arange(n)):
        perm = None
    A1, b1 = (A, b) if perm is None else (A[perm], b[perm])
    ids = list(np.unique(block_of)) if block_order is None else list(block_order)
    if eliminate is not None:
        eliminate = np.asarray(eliminate, dtype=bool)
        if eliminate.shape != (n,):
            raise ValueError("eliminate: a boolean mask over the unknowns")
        gone = {int(k) for k in np.unique(block_of[eliminate])}
        if any(not np.all(eliminate[block_of == k]) for k in gone):
            raise ValueError("eliminate: a block is either condensed as a whole or kept")
        ids = [k for k in ids if int(k) not in gone] + [k for k in ids if int(k) in gone]
        if not eliminate.any() or eliminate.all():
            eliminate = None
    rank = {int(k): i for i, k in enumerate(ids)}
    key = np.array([rank[int(k)] for k in block_of])
    order = np.argsort(key, kind="stable")
    A2 = sps.csr_matrix(A1[order][:, order])
    A2.sort_indices()
    b2 = b1[order]
    ptr = np.concatenate(([0], np.cumsum(np.bincount(key, minlength=len(ids))))).astype(np.int64)
    ctx = context if context is not None else _lib.Context(device, library)
    if eliminate is None:
        ctx.set_system(A2, b2)
    else:
        from .device_csr import DeviceCsr

        e2 = eliminate[order]
        d = A2.diagonal()
        if np.any(d[e2] == 0.0):
            raise ValueError("eliminate: a condensed unknown has a zero diagonal entry (pair the rows first)")
        # G = A_KE D^-1 as an n x n matrix (rows K, columns E): entries of A2 picked by index on the host, scaled;
        # the products L A = A - G A and L b = b - G b are the device's
        coo = A2.tocoo()
        pick = ~e2[coo.row] & e2[coo.col]
        Gfull = sps.csr_matrix((coo.data[pick] / d[coo.col[pick]], (coo.row[pick], coo.col[pick])), shape=(n, n))
        Gfull.sort_indices()
        J = DeviceCsr.from_scipy(A2, ctx)
        Gd = DeviceCsr.from_scipy(Gfull, ctx)
        LA = J - Gd @ J
        Lb = b2 - (Gd @ b2)
        LA.as_system(Lb, ctx)
        for m in (J, Gd, LA):
            m.close()
    ctx.set_block_preconditioner(ptr, gauss_seidel)
    x2, info = ctx.solve(method=method, rtol=rtol, maxit=maxit, restart=restart, n=n, precond="block", raise_on_fail=False)
    x = np.empty(n)
    x[order] = np.atleast_1d(x2)
    info = dict(info)
    info["blocks"] = len(ids)
    info["rows_matched"] = perm is not None
    info["condensed_unknowns"] = 0 if eliminate is None else int(eliminate.sum())
    info["true_rel_residual"] = float(np.linalg.norm(b - A @ x) / max(np.linalg.norm(b), 1e-300))
    return x, info


def solve_block_system_sharded(A, b, block_of, owner, dist, rtol: float = 1e-12, maxit: int = 5000, device: str = "cuda",
                               local_device_index: int = 0, library=None, gauss_seidel: bool = True, row_perm=None,
                               eliminate=None):
    """:func:`solve_block_system` over the ranks of a process group: the unknowns are dealt out by
    ``owner`` (by subdomain for a fracture network), every rank keeps the rows of its own unknowns
    (:class:`porepy_amd.distributed.ShardedCsr`: halo plan over cell, fracture and mortar unknowns) and preconditions
    with the block lower-triangular sweep over ITS (variable, subdomain) blocks -- block Jacobi across ranks,
    Gauss-Seidel inside -- in the library's fused BiCGStab loop (``pfv_solve_sharded`` with PFV_PRECOND_BLOCK).  The
    condensation of the interface fluxes (``eliminate``, see :func:`solve_block_system`) is formed by every rank for the
    whole system on the host (two sparse products of a matrix that every rank holds anyway at this point: the
    reference assembles its Jacobian on every process), then sharded.  Returns (x, info) with the full solution on every
    rank, in the caller's numbering."""
    import scipy.sparse as sps

    from .distributed import ShardedCsr

    A = sps.csr_matrix(A)
    b = np.asarray(b, dtype=float)
    n = A.shape[0]
    block_of = np.asarray(block_of)
    owner = np.asarray(owner)
    perm = match_rows(A) if row_perm is None else np.asarray(row_perm, dtype=np.int64)
    A1, b1 = (A, b) if perm is None else (A[perm], b[perm])
    ids = list(np.unique(block_of))
    if eliminate is not None:
        eliminate = np.asarray(eliminate, dtype=bool)
        gone = {int(k) for k in np.unique(block_of[eliminate])}
        ids = [k for k in ids if int(k) not in gone] + [k for k in ids if int(k) in gone]
        if not eliminate.any() or eliminate.all():
            eliminate = None
    rank_of = {int(k): i for i, k in enumerate(ids)}
    key = np.array([rank_of[int(k)] for k in block_of])
    order = np.argsort(key, kind="stable")
    A2 = sps.csr_matrix(A1[order][:, order])
    A2.sort_indices()
    b2 = b1[order]
    own2 = owner[order]
    key2 = key[order]
    if eliminate is not None:
        e2 = eliminate[order]
        d = A2.diagonal()
        if np.any(d[e2] == 0.0):
            raise ValueError("eliminate: a condensed unknown has a zero diagonal entry (pair the rows first)")
        coo = A2.tocoo()
        pick = ~e2[coo.row] & e2[coo.col]
        G = sps.csr_matrix((coo.data[pick] / d[coo.col[pick]], (coo.row[pick], coo.col[pick])), shape=(n, n))
        A2 = sps.csr_matrix(A2 - G @ A2)
        A2.sort_indices()
        b2 = b2 - G @ b2
    sh = ShardedCsr(A2, b2, own2, device=device, local_device_index=local_device_index, library=library, dist=dist)
    mine = sh.owned_gid                      # ascending: sorted by block key
    counts = np.bincount(key2[mine], minlength=len(ids))
    ptr = np.concatenate(([0], np.cumsum(counts[counts > 0]))).astype(np.int64)
    sh.set_block_preconditioner(ptr, gauss_seidel)
    x_own, info = sh.solve("bicgstab", rtol=rtol, maxit=maxit, precond="block")
    parts = [None] * (dist.get_world_size() if dist is not None else 1)
    payload = (mine, np.asarray(x_own.cpu().numpy() if hasattr(x_own, "cpu") else x_own))
    if dist is not None:
        dist.all_gather_object(parts, payload)
    else:
        parts = [payload]
    x2 = np.empty(n)
    for idx, vals in parts:
        x2[idx] = vals
    x = np.empty(n)
    x[order] = x2
    info = dict(info)
    info["blocks_here"] = int(ptr.size - 1)
    info["condensed_unknowns"] = 0 if eliminate is None else int(eliminate.sum())
    info["true_rel_residual"] = float(np.linalg.norm(b - A @ x) / max(np.linalg.norm(b), 1e-300))
    return x, info


def pair_equation_blocks(A, row_blocks, col_blocks):
    """Pair equation blocks with variable blocks of a coupled Jacobian.  ``row_blocks`` / ``col_blocks``: lists of
    ``(group, indices)`` -- group = the grid the equation / variable lives on; blocks of one group with equal sizes are
    candidates for each other (mass balance and energy balance against pressure and temperature of the same
    subdomain).  Within a group the assignment maximises the mean of log(|a_ii| / row maximum) of the candidate
    sub-block's diagonal -- how dominant the entry of the block's own unknown is in each of its equations.  Returns
    ``row_perm`` (see :func:`solve_block_system`)."""
    import itertools

    import scipy.sparse as sps

    A = sps.csr_matrix(A)
    n = A.shape[0]
    rowmax = np.maximum.reduceat(np.abs(A.data), A.indptr[:-1])
    rowmax[rowmax == 0] = 1.0
    perm = np.full(n, -1, dtype=np.int64)
    groups = {}
    for g, idx in row_blocks:
        groups.setdefault(g, ([], []))[0].append(np.asarray(idx))
    for g, idx in col_blocks:
        groups.setdefault(g, ([], []))[1].append(np.asarray(idx))
    for g, (rows, cols) in groups.items():
        if len(rows) != len(cols):
            raise ValueError(f"group {g!r}: {len(rows)} equation blocks for {len(cols)} variable blocks")
        m = len(rows)
        W = np.full((m, m), -1e6)
        for i, r in enumerate(rows):
            for j, c in enumerate(cols):
                if r.size == c.size:
                    d = np.abs(np.asarray(A[r, c]).ravel())
                    W[i, j] = float(np.mean(np.log(np.maximum(d, 1e-300) / rowmax[r]))) if np.all(d > 0) else -1e5
        best = max(itertools.permutations(range(m)), key=lambda p: sum(W[i, p[i]] for i in range(m))) if m <= 6 else None
        if best is None:
            from scipy.optimize import linear_sum_assignment

            ri, ci = linear_sum_assignment(-W)
            best = tuple(ci[np.argsort(ri)])
        for i, r in enumerate(rows):
            perm[cols[best[i]]] = r
    if np.any(perm < 0) or np.unique(perm).size != n:
        raise ValueError("the blocks do not cover the system")
    return perm


class HipLinearSolver:
    """Mixin for PorePy models (put it before the model class in the bases)."""

    #: library override for tests (host-emulation build); None = the gfx950 product library
    hip_library = None

    def _initialize_linear_solver(self) -> None:
        # the reference raises ValueError for names it does not know (solution_strategy.py:761-780)
        solver = self.params["linear_solver"]
        if solver in _METHODS:
            self.linear_solver = solver
        else:
            super()._initialize_linear_solver()

    def solve_linear_system(self) -> np.ndarray:
        solver = str(getattr(self, "linear_solver", self.params.get("linear_solver", "")))
        if solver not in _METHODS:
            return super().solve_linear_system()
        A, b = self.linear_system
        opts = self.params.get("hip_solver_options", {})
        if getattr(self, "_hip_solver_context", None) is None:
            self._hip_solver_context = _lib.Context(int(opts.get("device", 0)), self.hip_library)
        from .device_csr import DeviceCsr

        if isinstance(A, DeviceCsr) and str(opts.get("precond", "jacobi")) != "block":
            # the Jacobian was assembled on the device (DeviceAssembly): it becomes the active system of its own handle
            # device-to-device, the solve runs there -- no host copy of the matrix at any point
            ctx = A.as_system(np.asarray(b, dtype=float))
            x, info = ctx.solve(method=_METHODS[solver], rtol=float(opts.get("rtol", 1e-12)),
                                maxit=int(opts.get("maxit", 50000)), restart=int(opts.get("restart", 0)),
                                n=A.shape[0], precond=str(opts.get("precond", "jacobi")), raise_on_fail=False)
            if not info["converged"]:
                raise RuntimeError(f"hip solver did not converge on the device Jacobian: {info}")
            self.hip_solver_info = dict(info, device_jacobian=True)
            x = np.atleast_1d(x)
            if self._apply_schur_complement_reduction():
                x = self.equation_system.expand_schur_complement_solution(x)
            return x
        if isinstance(A, DeviceCsr):
            A = A.to_scipy()  # (the block path permutes rows and columns on the host)
        if str(opts.get("precond", "jacobi")) == "block":
            block_of, row_perm = self._hip_blocks(opts)
            # interface unknowns are condensed into the cell unknowns (solve_block_system: eliminate) once there are
            # more interfaces than a sweep over (variable, grid) blocks copes with; opts["condense_interfaces"]
            # (True / False) overrides the threshold
            elim = getattr(self, "_hip_interface_mask", None)
            if elim is None or not elim.any() or row_perm is None:
                elim = None
            shard = opts.get("sharded")
            if shard is not None:
                # The coupled Jacobian solved over the ranks of a process group (round 6: the md model with its
                # discretization loop AND its Newton solves sharded): ``opts["sharded"] = {"dist": torch.distributed (or
                # a stand-in with the same calls), "device": ..., "local_device_index": ...}``.  Unknowns are dealt out by
                # position inside every variable-wide block -- the reference numbers a variable grid by grid, the 3-D
                # matrix first: equal shares are a slab of matrix cells plus runs of whole fracture planes, lines and
                # mortar grids (their unknowns reach the other ranks through the halo plan of ShardedCsr).
                dist = shard["dist"]
                world = int(dist.get_world_size())
                owner = np.zeros(A.shape[0], dtype=np.int64)
                for k in np.unique(block_of):
                    idx = np.flatnonzero(block_of == k)
                    owner[idx] = (np.arange(idx.size) * world) // idx.size
                x, info = solve_block_system_sharded(A, b, block_of, owner, dist, rtol=float(opts.get("rtol", 1e-12)),
                                                     maxit=int(opts.get("maxit", 4000)),
                                                     device=str(shard.get("device", "cuda")),
                                                     local_device_index=int(shard.get("local_device_index", 0)),
                                                     library=self.hip_library,
                                                     gauss_seidel=bool(opts.get("gauss_seidel", True)), row_perm=row_perm,
                                                     eliminate=elim)
                if not info["converged"]:
                    raise RuntimeError(f"hip sharded block solver did not converge: {info}")
                self.hip_solver_info = dict(info, sharded_world=world)
                x = np.atleast_1d(x)
                if self._apply_schur_complement_reduction():
                    x = self.equation_system.expand_schur_complement_solution(x)
                return x
            x, info = solve_block_system(A, b, block_of, method=_METHODS[solver],
                                         rtol=float(opts.get("rtol", 1e-12)), maxit=int(opts.get("maxit", 2000)),
                                         restart=int(opts.get("restart", 60)), context=self._hip_solver_context,
                                         gauss_seidel=bool(opts.get("gauss_seidel", True)), row_perm=row_perm,
                                         eliminate=elim)
            if not info["converged"]:
                raise RuntimeError(f"hip block solver did not converge: {info}")
            self.hip_solver_info = info
            x = np.atleast_1d(x)
            if self._apply_schur_complement_reduction():
                x = self.equation_system.expand_schur_complement_solution(x)
            return x
        x, info = solve_csr(A, b, method=_METHODS[solver], rtol=float(opts.get("rtol", 1e-12)),
                            maxit=int(opts.get("maxit", 50000)), restart=int(opts.get("restart", 0)),
                            context=self._hip_solver_context, precond=str(opts.get("precond", "jacobi")))
        self.hip_solver_info = info
        x = np.atleast_1d(x)
        if self._apply_schur_complement_reduction():
            x = self.equation_system.expand_schur_complement_solution(x)
        return x

    def _hip_blocks(self, opts):
        """``(block_of, row_perm)`` for ``self.linear_system``.  block_of: one block per (variable, grid) of the model's
        ``equation_system`` (numerics/ad/equation_system.py: every md-variable occupies contiguous dofs on each of its
        grids), cell variables first -- by ``opts["variable_order"]`` if given, else pressure-like names first -- then
        the interface variables.  row_perm: the model's equation blocks (``assembled_equation_indices``, split by the
        grids of their image space) paired with the variable blocks of the same grid (:func:`pair_equation_blocks`);
        None (entry-wise matching) where the model does not expose that structure, e.g. after a Schur reduction."""
        es = self.equation_system
        A = self.linear_system[0]
        if hasattr(A, "to_scipy"):  # (a device Jacobian of DeviceAssembly: the pairing below reads entries on the host)
            A = A.to_scipy()
        n = A.shape[0]
        block = np.full(n, -1, dtype=np.int64)
        wanted = list(opts.get("variable_order", []))

        def rank(var):
            name = var.name
            if name in wanted:
                return (0, wanted.index(name))
            is_intf = "interface" in name or hasattr(var.domain, "mortar_grid") or not hasattr(var.domain, "cell_faces")
            return (2 if is_intf else 1, 0 if "pressure" in name else 1)

        variables = sorted(es.variables, key=lambda v: (rank(v), v.name, -getattr(v.domain, "dim", 0), getattr(v.domain, "id", 0)))
        intf_blocks = sum(1 for v in variables if rank(v)[0] == 2)
        # one block per (variable, grid); per variable over all its grids when the interfaces are condensed (the Schur
        # complement couples a variable across its grids directly: it is ONE elliptic operator, coarsened as one)
        condensing = opts.get("condense_interfaces", intf_blocks > int(opts.get("condense_above_interface_blocks", 24)))
        per_variable = str(opts.get("block_granularity", "variable" if condensing else "grid")) == "variable"
        k = 0
        col_blocks = []
        mask = np.zeros(n, dtype=bool)
        number: dict = {}
        for var in variables:
            dofs = np.asarray(es.dofs_of([var]))
            if dofs.size and dofs.max() < n:
                if per_variable:
                    block[dofs] = number.setdefault(var.name, len(number))
                else:
                    block[dofs] = k
                mask[dofs] = rank(var)[0] == 2
                col_blocks.append((id(var.domain), dofs))
                k += 1
        self._hip_interface_mask = mask if condensing else None
        if np.any(block < 0):
            block[block < 0] = k  # (unknowns outside the variable list: one block)
            self._hip_interface_mask = None
            return block, None
        row_perm = None
        try:
            row_blocks = []
            for name, rows in es.assembled_equation_indices.items():
                rows = np.asarray(rows)
                for grid, loc in es._equation_image_space_composition[name].items():
                    loc = np.asarray(loc)
                    if loc.size:
                        row_blocks.append((id(grid), rows[loc]))
            row_perm = pair_equation_blocks(A, row_blocks, col_blocks)
        except Exception:  # noqa: BLE001 - structure not exposed: fall back to the entry-wise matching
            row_perm = None
        return block, row_perm


class DeviceAssembly:
    """Mixin for PorePy models (before the model class, beside :class:`HipLinearSolver`): ``assemble_linear_system``
    evaluates the model's operator trees with device-resident Jacobians AND device-resident discretization-matrix leaves
    (``porepy_amd.ad.assemble_on_device(..., device_leaves=True)``; the reference: models/solution_strategy.py:782-827 ->
    ``EquationSystem.assemble``, numerics/ad/equation_system.py:1579).  ``self.linear_system`` then holds
    ``(DeviceCsr, numpy residual)``; :class:`HipLinearSolver` solves it without a host copy of the matrix.  With
    ``pp.Mpfa = as_porepy_discretization(lazy=True)`` the discretization matrices never leave the device either:
    discretize -> operator tree -> Jacobian -> Krylov solve, all in HBM; what crosses PCIe per Newton iteration is the
    state vector, the residual and the increment."""

    hip_library = None

    def assemble_linear_system(self) -> None:
        if self._apply_schur_complement_reduction():
            return super().assemble_linear_system()
        from . import ad

        if getattr(self, "_hip_assembly_context", None) is None:
            opts = self.params.get("hip_solver_options", {})
            self._hip_assembly_context = _lib.Context(int(opts.get("device", 0)), self.hip_library)
        J, b = ad.assemble_on_device(self.equation_system, self._hip_assembly_context, device_leaves=True)
        self.linear_system = (J, b)
