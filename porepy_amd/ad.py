"""Evaluation of the reference's AD operator trees with the Jacobians resident on the device.

``EquationSystem.assemble`` (numerics/ad/equation_system.py:1579) evaluates every equation of a model by forward-mode
AD: the parser (numerics/ad/_ad_parser.py) walks the operator tree, leaves become numpy arrays / scipy matrices / slices
``ad_base[dofs]`` of the identity ``AdArray``, inner nodes apply ``+ - * / ** @`` and operator functions to them
(numerics/ad/forward_mode.py).  All of the Jacobian work in there is sparse algebra on matrices with one column per
degree of freedom -- products with divergences, projections and discretization matrices, row scalings by the values
of the other factor, sums.

Here the same walk (the reference's own parser and ``AdArray`` arithmetic, untouched) starts from an ``AdArray`` whose
Jacobian is a :class:`porepy_amd.DeviceCsr` identity: every Jacobian the tree produces is then formed on the device
(``csrc/csr_algebra.inc``) with scipy's conventions -- sorted rows, products and sums in scipy's order -- and stays
there; the values (vectors) are numpy arrays as in the reference.  The assembled Jacobian can go straight to the device
solver (``DeviceCsr.as_system`` / ``solvers.solve_csr``) without having been on the host.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from .device_csr import DeviceCsr, vstack


def device_ad_base(equation_system, context: "_lib.Context", state=None):
    """The identity ``AdArray`` of the reference (``forward_mode.initAdArrays``) with its Jacobian on the device."""
    import porepy as pp
    import scipy.sparse as sps

    if state is None:
        state = equation_system.get_variable_values(iterate_index=0)
    state = np.asarray(state, dtype=float)
    eye = DeviceCsr.from_scipy(sps.identity(state.size, format="csr"), context)
    return pp.ad.AdArray(state, eye)


def evaluate_on_device(operator, equation_system, context: "_lib.Context", state=None, ad_base=None):
    """Value and Jacobian of one ``pp.ad.Operator``: a ``pp.ad.AdArray`` whose ``jac`` is a ``DeviceCsr`` (operators
    without a dependence on the variables get an empty device Jacobian)."""
    import porepy as pp
    import scipy.sparse as sps

    parser = equation_system._ad_parser
    base = device_ad_base(equation_system, context, state) if ad_base is None else ad_base
    res = parser._evaluate_single(operator, base, equation_system)
    if ad_base is None:
        parser.clear_cache()
    if isinstance(res, (int, float)):
        res = np.array([float(res)])
    if isinstance(res, np.ndarray):
        res = pp.ad.AdArray(res, DeviceCsr.from_scipy(sps.csr_matrix((res.shape[0], equation_system.num_dofs())), context))
    if not isinstance(res.jac, DeviceCsr):
        # (an operator function that built its Jacobian with scipy constructors: brought back to the device)
        res = pp.ad.AdArray(res.val, DeviceCsr.from_any(sps.csr_matrix(res.jac), context))
    return res


import contextlib


@contextlib.contextmanager
def device_matrix_leaves(context: "_lib.Context"):
    """Inside the block the discretization-matrix leaves of the operator trees (``MergedOperator.parse``,
    numerics/ad/ad_utils.py:597-663: the block-diagonal concatenation of the subdomains' matrices) are formed ON THE
    DEVICE as ``DeviceCsr`` -- from lazily kept device matrices without a host copy (``Mpfa(lazy=True)``), from host
    matrices by an upload that is remembered per matrix object.  Products with them (value: device SpMV, Jacobian:
    device product) then never see a host copy of the discretization."""
    import porepy as pp
    from .device_csr import block_diag

    cls = pp.ad.MergedOperator
    orig = cls.parse

    def parse(self, mdg):
        if len(self.domains) == 0:
            return orig(self, mdg)
        mats = []
        for grid in self.domains:
            if isinstance(grid, pp.MortarGrid):
                data = mdg.interface_data(grid)
            elif isinstance(grid, pp.Grid):
                data = mdg.subdomain_data(grid)
            else:
                return orig(self, mdg)
            md = data[pp.DISCRETIZATION_MATRICES][self._physics_key]
            m = md[getattr(self._discr, self._discretization_matrix_key + "_matrix_key")]
            if self._inner_physics_key is not None:
                m = m[self._inner_physics_key]
            if isinstance(m, np.ndarray):
                return orig(self, mdg)
            mats.append(m)
        return block_diag(mats, context)

    cls.parse = parse
    try:
        yield
    finally:
        cls.parse = orig


def assemble_on_device(equation_system, context: "_lib.Context", state=None, equations=None, device_leaves: bool = False):
    """``EquationSystem.assemble`` with the Jacobian formed and kept on the device: returns ``(J, b)`` with ``J`` a
    ``DeviceCsr`` (all equations stacked in the order of ``equation_system.equations``, all variables) and
    ``b = -residual`` a numpy vector, i.e. the linear system ``J dx = b`` of a Newton iteration
    (equation_system.py:1579-1700).  ``device_leaves``: the discretization matrices enter the trees as ``DeviceCsr`` too
    (:func:`device_matrix_leaves`): values by device SpMV -- equal to the host products to rounding, not to the bit."""
    names = list(equation_system.equations) if equations is None else list(equations)
    base = device_ad_base(equation_system, context, state)
    vals, jacs = [], []
    leaves = device_matrix_leaves(context) if device_leaves else contextlib.nullcontext()
    # rows of every assembled equation, as EquationSystem.assemble leaves them (equation_system.py:1654-1693): the
    # block preconditioner (HipLinearSolver) and the reference's diagnostics read them
    indices, start = {}, 0
    try:
        with leaves:
            for name in names:
                ad = evaluate_on_device(equation_system.equations[name], equation_system, context, ad_base=base)
                rows = int(np.asarray(ad.val).size)
                indices[name] = np.arange(rows) + start
                start += rows
                if ad.val.size:
                    vals.append(ad.val)
                    jacs.append(ad.jac)
        equation_system.assembled_equation_indices = indices
    finally:
        equation_system._ad_parser.clear_cache()
    J = jacs[0] if len(jacs) == 1 else vstack(jacs, context)
    return J, -np.concatenate(vals)
