"""Mixed-dimensional grids shard BY SUBDOMAIN: the reference's discretization loop, one share per GPU.

The reference discretizes a mixed-dimensional model in one serial loop over (discretization object, grid)
pairs -- ``pp.ad.discretize_from_list`` (/root/reference/src/porepy/numerics/ad/ad_utils.py:281-308), called by
``EquationSystem.discretize`` (numerics/ad/equation_system.py:1529-1559) and by the nonlinear re-discretization of
the models (models/solution_strategy.py:995, 1014).  Every pair is independent of the others: the call reads
``data[PARAMETERS]`` of its grid(s) and stores matrices under ``data[DISCRETIZATION_MATRICES]``.  That is the natural
partition of BASELINE configs[4] (a 52-fracture network: one 3-D matrix grid, 52 fracture planes, ~100 intersection
lines, the mortar grids between them): no halo, no collective inside the assembly.

``discretize_from_list_sharded`` is the same loop with the pairs dealt out to the ranks (longest job first onto
the least loaded rank, the cost of a pair taken from its cell count and dimension), each rank running its share on
its own GPU through whatever ``pp.Mpfa`` / ``pp.Mpsa`` are bound to, and ONE exchange at the end in which every
rank hands the matrices its jobs stored to all others (``torch.distributed.all_gather_object`` on the group given
-- RCCL on the GPU box, gloo in the CPU tests).  After the call every rank holds the data dictionaries the serial
loop would have left: the AD assembly of the Jacobian (which the reference does in one process) can run anywhere.
``sharded_discretization(pp, ...)`` rebinds ``pp.ad.discretize_from_list`` for the duration of a ``with`` block, so
an unmodified model shards its discretization -- the same kind of rebind as ``pp.Mpfa = HipMpfa``.

With one 3-D grid carrying most of the cells the speed-up of a loop that hands out WHOLE grids is bounded by
``sum(cost) / max(cost)`` (2.3 on the 52-fracture model).  Round 6 composes the two partitions: a job that would bound the
loop -- cost above the mean load per rank -- and whose discretization object can discretize one PIECE of a grid
(``discretize_piece`` / ``merge_piece_payloads``: ``porepy_amd.Mpfa`` and its drop-in subclass; cells along a Morton curve,
one node ring of overlap, ``distributed.extract_subdomain`` -- the routine the cell-sharded single-grid path uses) is cut into
piece jobs that are dealt out like any other job; the rows travel in the same single exchange and EVERY rank merges them
(host scipy, as the reference's own merge of its sub-problems, numerics/fv/mpfa.py:298-372), so all ranks again hold what
the serial loop would have left -- to rounding: the rows of a face two pieces computed are averaged.  ``Plan.bound`` is then
``sum(cost) / max(cost)`` over the PIECE jobs.
"""
from __future__ import annotations

import contextlib
from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sps


@dataclass
class Job:
    """One call of the reference's loop: ``discr.discretize(grid, data)`` or the interface form."""

    discr: object
    grid: object
    is_interface: bool
    cost: float = 0.0
    owner: int = 0
    piece: tuple | None = None   # (index, number of pieces, index of the parent job in the reference's loop order)


@dataclass
class Plan:
    jobs: list
    world: int
    load: np.ndarray = field(default_factory=lambda: np.zeros(0))
    serial_cost: float = 0.0   # cost of the undivided loop (pieces recompute a node ring: their costs add up to more)

    @property
    def bound(self) -> float:
        """Largest possible speed-up of the loop: total cost / largest single job."""
        c = np.array([j.cost for j in self.jobs])
        return float(c.sum() / c.max()) if c.size and c.max() > 0 else 1.0

    @property
    def speedup(self) -> float:
        """Speed-up of this assignment by the cost model: total cost / load of the busiest rank."""
        return float(self.load.sum() / self.load.max()) if self.load.size and self.load.max() > 0 else 1.0

    def summary(self) -> dict:
        return {"jobs": len(self.jobs), "world": self.world, "load": [float(x) for x in self.load],
                "speedup_vs_the_serial_loop_by_cost_model": (float(self.serial_cost / self.load.max())
                                                             if self.load.size and self.load.max() > 0 and self.serial_cost > 0 else None),
                "piece_jobs": int(sum(1 for j in self.jobs if j.piece is not None)),
                "subdomains_cut_into_pieces": len({j.piece[2] for j in self.jobs if j.piece is not None}),
                "speedup_by_cost_model": self.speedup, "bound_total_over_largest_job": self.bound,
                "jobs_per_rank": [int(sum(1 for j in self.jobs if j.owner == r)) for r in range(self.world)]}


# cost of one (discretization, grid) pair relative to a 3-D cell of a local-system discretization
_LOCAL_SYSTEM = ("Mpfa", "Mpsa", "Biot")


def default_cost(discr, grid, is_interface: bool) -> float:
    """Relative cost: the interaction-region discretizations dominate (a 3-D tetrahedral cell costs ~7x a 2-D one:
    36 sub-faces per node against 12, cubic in the elimination); two-point / upwind / coupling discretizations
    are a pass over the cells."""
    n = float(getattr(grid, "num_cells", 1))
    if is_interface:
        return 0.02 * n + 1.0
    name = type(discr).__name__
    bases = {b.__name__ for b in type(discr).__mro__}
    dim = int(getattr(grid, "dim", 0))
    if dim >= 2 and (bases & set(_LOCAL_SYSTEM) or any(t in name for t in _LOCAL_SYSTEM)):
        return n * (1.0 if dim == 3 else 0.15) * (3.0 if ("Mpsa" in bases or "Biot" in bases) else 1.0) + 1.0
    return 0.02 * n + 1.0


PIECE_OVERLAP = 1.15  # cost of a piece relative to its share of the cells: the node ring it recomputes


def plan(discretizations: dict, world: int, cost=None, is_interface=None, split: bool = True) -> Plan:
    """The jobs of ``discretize_from_list`` in the reference's own order, with an owner each.

    Longest-processing-time-first: jobs by decreasing cost (ties: loop order) onto the least loaded rank (ties:
    lowest rank) -- deterministic, every rank computes the same plan without talking.
    ``split``: a subdomain job whose cost exceeds the mean load per rank, on an object with ``discretize_piece``, becomes
    ``k`` piece jobs (``k`` = the smallest number of pieces that brings one piece under half the mean load, at most
    ``2 * world``); their ``piece`` field carries (index, k, position of the parent in the loop)."""
    cost = cost or default_cost
    if is_interface is None:
        def is_interface(g):
            return hasattr(g, "mortar_to_primary_int") or type(g).__name__ == "MortarGrid"
    whole = []
    for discr in discretizations:
        for grid in discretizations[discr]:
            intf = bool(is_interface(grid))
            whole.append(Job(discr, grid, intf, float(cost(discr, grid, intf))))
    jobs = []
    world_i = max(1, int(world))
    mean = sum(j.cost for j in whole) / world_i
    for pos, j in enumerate(whole):
        can = (split and world_i > 1 and not j.is_interface and hasattr(j.discr, "discretize_piece")
               and hasattr(j.discr, "merge_piece_payloads") and j.cost > mean and getattr(j.grid, "dim", 0) >= 2)
        if not can:
            jobs.append(j)
            continue
        # pieces of at most HALF the mean load: the greedy assignment below then ends within a few per cent of the mean
        # (a job that is all of the work on 4 ranks: 8 pieces, two per rank; pieces of a whole mean load: 5, one rank
        # takes two -- 2.5x instead of 3.5x)
        k = int(min(2 * world_i, max(2, np.ceil(2.0 * PIECE_OVERLAP * j.cost / max(mean, 1e-300)))))
        for p in range(k):
            jobs.append(Job(j.discr, j.grid, False, PIECE_OVERLAP * j.cost / k, 0, (p, k, pos)))
    load = np.zeros(max(1, int(world)))
    order = sorted(range(len(jobs)), key=lambda i: (-jobs[i].cost, i))
    for i in order:
        r = int(np.argmin(load))
        jobs[i].owner = r
        load[r] += jobs[i].cost
    return Plan(jobs, max(1, int(world)), load, float(sum(j.cost for j in whole)))


_ABSENT = object()


def _host_value(v):
    """What travels: scipy matrices / numpy arrays / plain Python values.  Device-resident and lazily fetched
    matrices of this package are fetched (``DeviceCsr.to_scipy`` / ``LazyCsr.tocsr``)."""
    if sps.issparse(v) or isinstance(v, (np.ndarray, float, int, str, bool, type(None))):
        return v
    if hasattr(v, "to_scipy"):
        return v.to_scipy()
    if hasattr(v, "tocsr"):
        return v.tocsr()
    return v


def _matrix_slots(pp, mdg, job: Job):
    """The data dictionaries a job may write to (ad_utils.py:294-305)."""
    if job.is_interface:
        g_primary, g_secondary = mdg.interface_to_subdomain_pair(job.grid)
        return [mdg.subdomain_data(g_primary), mdg.subdomain_data(g_secondary), mdg.interface_data(job.grid)]
    return [mdg.subdomain_data(job.grid)]


def _digest(v):
    """Cheap content digest of a stored matrix / array: a discretization that updates a stored matrix IN PLACE (same
    object, new values) is then still seen as a change by the job that did it."""
    try:
        if sps.issparse(v):
            d = np.asarray(v.data)
            return (v.shape, int(v.nnz), float(d.sum()), float(np.abs(d).sum()))
        if isinstance(v, np.ndarray):
            return (v.shape, float(np.sum(v)), float(np.abs(v).sum()))
    except Exception:  # noqa: BLE001 - a value without a cheap digest is compared by identity only
        pass
    return None


def _snapshot(pp, slots):
    snap = {}
    for s, d in enumerate(slots):
        for kw, md in d.get(pp.DISCRETIZATION_MATRICES, {}).items():
            for name, v in md.items():
                # (the object itself: an id could be recycled once the job overwrites it; the digest catches in-place updates)
                snap[(s, kw, name)] = (v, _digest(v))
    return snap


def _run(pp, mdg, job: Job, slots):
    if job.is_interface:
        g_primary, g_secondary = mdg.interface_to_subdomain_pair(job.grid)
        job.discr.discretize(g_primary, g_secondary, job.grid, slots[0], slots[1], slots[2])
    else:
        try:
            job.discr.discretize(job.grid, slots[0])
        except NotImplementedError:  # (as the reference: GradP and other Biot helpers, ad_utils.py:306-308)
            pass


def _merge_pieces(pp, mdg, pl: Plan, entries: dict, slots_of: dict) -> None:
    """All pieces of ONE parent job are in ``entries`` (job index -> [("__piece__", p, k, payload)]): merged into the
    grid's data dictionary by the discretization object -- on every rank, from the same payloads in the same order."""
    idx = sorted(entries)
    job = pl.jobs[idx[0]]
    k = job.piece[1]
    payloads = [entries[i][0][3] for i in idx]
    if len(payloads) != k:
        raise RuntimeError(f"piece exchange incomplete: {len(payloads)} of {k} pieces of one subdomain arrived")
    data = slots_of[idx[0]][0] if idx[0] in slots_of else _matrix_slots(pp, mdg, job)[0]
    job.discr.merge_piece_payloads(job.grid, data, payloads)


def discretize_from_list_sharded(discretizations: dict, mdg, pp=None, group=None, rank: int | None = None,
                                 world: int | None = None, cost=None, exchange=None, stats: dict | None = None,
                                 batch: bool = True, split: bool = True):
    """``pp.ad.discretize_from_list`` (ad_utils.py:281-308) with the (discretization, grid) pairs dealt out to ranks.

    ``exchange(payload) -> list of payloads by rank`` defaults to ``torch.distributed.all_gather_object`` on
    ``group``; with ``world == 1`` the call IS the reference's loop.  ``stats`` (a dict) receives the plan summary,
    the jobs run here and the bytes of matrices sent.
    ``batch``: the subdomain jobs a rank owns for one discretization object go to its ``discretize_batch`` in one call
    where the object has one (``porepy_amd.Mpfa``: grids of one dimension as ONE disjoint union on the device -- 52
    fracture planes in one discretization instead of 52), in the loop's order otherwise."""
    if pp is None:
        import porepy as pp  # noqa: PLC0415  (the reference package this loop belongs to)
    if exchange is None and (world is None or rank is None):
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            world = dist.get_world_size(group) if world is None else world
            rank = dist.get_rank(group) if rank is None else rank
        else:
            world, rank = 1, 0
    pl = plan(discretizations, world, cost, is_interface=lambda g: isinstance(g, pp.MortarGrid), split=split)
    mine: dict = {}
    mine_objects: dict = {}  # the same entries with the objects as stored here (device-resident proxies stay what they are)
    own = [i for i, job in enumerate(pl.jobs) if job.owner == rank]
    slots_of = {i: _matrix_slots(pp, mdg, pl.jobs[i]) for i in own}
    before_of = {i: _snapshot(pp, slots_of[i]) for i in own}
    done = set()
    batches = 0
    if batch:
        by_discr: dict = {}
        for i in own:
            job = pl.jobs[i]
            if not job.is_interface and job.piece is None and hasattr(job.discr, "discretize_batch"):
                by_discr.setdefault(id(job.discr), (job.discr, []))[1].append(i)
        for discr, idx in by_discr.values():
            if len(idx) > 1:
                discr.discretize_batch([(pl.jobs[i].grid, slots_of[i][0]) for i in idx])
                done.update(idx)
                batches += 1
    piece_payloads: dict = {}  # job index -> payload of a piece this rank discretized
    for i in own:
        if pl.jobs[i].piece is not None:
            # one piece of a large subdomain: nothing is stored here, the rows travel and every rank merges them below
            p, k, _ = pl.jobs[i].piece
            piece_payloads[i] = pl.jobs[i].discr.discretize_piece(pl.jobs[i].grid, slots_of[i][0], p, k)
            mine[i] = [("__piece__", p, k, piece_payloads[i])]
            mine_objects[i] = mine[i]
            continue
        if i not in done:
            _run(pp, mdg, pl.jobs[i], slots_of[i])
        out, kept = [], []
        for s, d in enumerate(slots_of[i]):
            for kw, md in d.get(pp.DISCRETIZATION_MATRICES, {}).items():
                for name, v in md.items():
                    was = before_of[i].get((s, kw, name), _ABSENT)
                    if was is _ABSENT or was[0] is not v or was[1] != _digest(v):
                        out.append((s, kw, name, _host_value(v)))
                        kept.append((s, kw, name, v))
        mine[i] = out
        mine_objects[i] = kept
    sent = 0
    if world == 1 and piece_payloads:  # (cannot happen with plan(): one rank never splits; kept for callers' own plans)
        _merge_pieces(pp, mdg, pl, {i: mine[i] for i in piece_payloads}, slots_of)
    if world > 1:
        if exchange is None:
            import torch.distributed as dist

            def exchange(payload):
                got = [None] * world
                dist.all_gather_object(got, payload, group=group)
                return got
        # Every rank applies ALL results -- its own too -- in ascending job index, i.e. in the order of the reference's
        # loop (ad_utils.py:288-308): where two jobs write the same (slot, keyword, name) (an interface discretization
        # and a subdomain discretization storing into the primary grid's dictionary), the last writer of the serial
        # loop wins on every rank, whoever ran it.
        results: dict = {}
        for r, theirs in enumerate(exchange(mine)):
            for i, payload in (mine_objects if r == rank else theirs).items():
                results[i] = payload
        merged_parents = set()
        for i in sorted(results):
            slots = slots_of[i] if i in slots_of else _matrix_slots(pp, mdg, pl.jobs[i])
            if pl.jobs[i].piece is not None:
                # the pieces of one parent job are merged where its FIRST piece stands in the order of application
                parent = pl.jobs[i].piece[2]
                if parent not in merged_parents:
                    merged_parents.add(parent)
                    _merge_pieces(pp, mdg, pl, {q: results[q] for q in results
                                                if pl.jobs[q].piece is not None and pl.jobs[q].piece[2] == parent}, slots_of)
                continue
            for (s, kw, name, v) in results[i]:
                slots[s].setdefault(pp.DISCRETIZATION_MATRICES, {}).setdefault(kw, {})[name] = v
        for out in mine.values():
            for (_, _, _, v) in out:
                if isinstance(v, dict):  # a piece's rows
                    for trip in list(v.get("mats", {}).values()) + ([v["system"]] if v.get("system") is not None else []):
                        sent += sum(int(np.asarray(x).nbytes) for x in trip)
                    continue
                if sps.issparse(v):
                    m = v.tocsr() if not sps.isspmatrix_csr(v) and not sps.isspmatrix_csc(v) else v
                    sent += m.data.nbytes + m.indices.nbytes + m.indptr.nbytes
                elif isinstance(v, np.ndarray):
                    sent += v.nbytes
    if stats is not None:
        stats.setdefault("calls", 0)
        stats["calls"] += 1
        stats["plan"] = pl.summary()
        stats.setdefault("plans", []).append(stats["plan"])
        stats["jobs_run_here"] = stats.get("jobs_run_here", 0) + len(mine)
        stats["batch_calls"] = stats.get("batch_calls", 0) + batches
        stats["jobs_in_batches"] = stats.get("jobs_in_batches", 0) + len(done)
        stats["matrix_bytes_sent"] = stats.get("matrix_bytes_sent", 0) + int(sent)
        stats["rank"] = rank
    return pl


@contextlib.contextmanager
def sharded_discretization(pp, group=None, rank: int | None = None, world: int | None = None, cost=None,
                           exchange=None, stats: dict | None = None, batch: bool = True, split: bool = True):
    """Rebind ``pp.ad.discretize_from_list`` (the loop every model discretizes through: equation_system.py:1559,
    solution_strategy.py:995 / 1014, operators.py:487) to the sharded loop inside the ``with`` block."""
    orig = pp.ad.discretize_from_list

    def sharded(discretizations, mdg):
        return discretize_from_list_sharded(discretizations, mdg, pp=pp, group=group, rank=rank, world=world,
                                            cost=cost, exchange=exchange, stats=stats, batch=batch, split=split)

    pp.ad.discretize_from_list = sharded
    try:
        yield sharded
    finally:
        pp.ad.discretize_from_list = orig


def batched_discretization(pp, stats: dict | None = None):
    """One process: the reference's loop with the subdomain jobs of every discretization object handed to its
    ``discretize_batch`` (``sharded_discretization`` with one rank)."""
    return sharded_discretization(pp, rank=0, world=1, stats=stats, batch=True)
