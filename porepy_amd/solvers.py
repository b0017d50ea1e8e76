"""Device linear solve for assembled PorePy systems — the caller one level above the
discretization: ``SolutionStrategy.solve_linear_system``
(models/solution_strategy.py:830-884) holds the global Jacobian as a scipy CSR matrix and the
residual as a numpy vector and hands them to a direct solver; :class:`HipLinearSolver` routes
them to the Jacobi-preconditioned Krylov solvers behind ``pfv_set_system`` / ``pfv_solve``.

    class Model(porepy_amd.HipLinearSolver, SinglePhaseFlow): ...
    params = {"linear_solver": "hip_bicgstab", ...}      # or "hip_gmres", "hip_cg"
    params["hip_solver_options"] = {"precond": "amg", "rtol": 1e-12}   # optional

Any other ``linear_solver`` value falls through to the reference implementation.  Systems with
zero diagonal entries (saddle-point blocks of mixed-dimensional models) are refused by the
library, not solved badly.
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
        x, info = solve_csr(A, b, method=_METHODS[solver], rtol=float(opts.get("rtol", 1e-12)),
                            maxit=int(opts.get("maxit", 50000)), restart=int(opts.get("restart", 0)),
                            context=self._hip_solver_context, precond=str(opts.get("precond", "jacobi")))
        self.hip_solver_info = info
        x = np.atleast_1d(x)
        if self._apply_schur_complement_reduction():
            x = self.equation_system.expand_schur_complement_solution(x)
        return x
