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


def assemble_on_device(equation_system, context: "_lib.Context", state=None, equations=None):
    """``EquationSystem.assemble`` with the Jacobian formed and kept on the device: returns ``(J, b)`` with ``J`` a
    ``DeviceCsr`` (all equations stacked in the order of ``equation_system.equations``, all variables) and
    ``b = -residual`` a numpy vector, i.e. the linear system ``J dx = b`` of a Newton iteration
    (equation_system.py:1579-1700)."""
    names = list(equation_system.equations) if equations is None else list(equations)
    base = device_ad_base(equation_system, context, state)
    vals, jacs = [], []
    try:
        for name in names:
            ad = evaluate_on_device(equation_system.equations[name], equation_system, context, ad_base=base)
            if ad.val.size:
                vals.append(ad.val)
                jacs.append(ad.jac)
    finally:
        equation_system._ad_parser.clear_cache()
    J = jacs[0] if len(jacs) == 1 else vstack(jacs, context)
    return J, -np.concatenate(vals)
