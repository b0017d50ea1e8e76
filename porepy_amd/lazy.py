"""Lazily fetched discretization matrices (opt-in: ``Mpfa(keyword, lazy=True)``).

The six MPFA matrices of a 2 M-cell grid are 21.6 GB; copying them to the host takes several times longer
than computing them, and most callers only ever touch ``flux`` / ``bound_flux`` (through
``assemble_matrix_rhs``, which works on the device-resident copy anyway) or a few rows.  A ``LazyCsr`` stands
in ``data[DISCRETIZATION_MATRICES][kw][name]``: it knows its shape without a transfer, hands out row slices
through ``pfv_get_matrix_rows`` (gathered on the device), multiplies vectors on the device, and turns into a
plain ``scipy.sparse.csr_matrix`` (fetched once, then cached) the first time anything else is asked of it.
Before the handle overwrites a discretization, every proxy still alive is materialized, so a proxy always
holds the values of the ``discretize`` call that created it -- the semantics of the eager path.

The reference's own models keep the eager path (``isinstance(m, scipy.sparse.spmatrix)`` checks in its AD
layer would not accept a proxy)."""
from __future__ import annotations

import weakref

import numpy as np


class LazyCsr:
    __array_priority__ = 20.0  # numpy defers to our __rmatmul__ / __rmul__

    def __init__(self, ctx, which: int, post=None, shape=None, right=None):
        # right: when the post-processing is a product with a (small) sparse matrix from the right, that matrix -- a
        # device-side consumer (device_csr.DeviceCsr.from_any) then forms the product on the device instead of fetching
        self._right = right
        self._ctx = ctx
        self._which = int(which)
        self._post = post          # optional host-side post-processing of the fetched matrix
        self._m = None
        nrows, ncols, nnz = ctx.matrix_info(which)
        # (shape: what the post-processing turns the matrix into, when it changes it)
        self._shape = (int(nrows), int(ncols)) if shape is None else (int(shape[0]), int(shape[1]))
        self._nnz = int(nnz)
        ctx._lazy_refs.append(weakref.ref(self))

    # ---- no transfer -------------------------------------------------------------------
    @property
    def shape(self):
        return self._m.shape if self._m is not None else self._shape

    @property
    def nnz(self):
        return self._m.nnz if self._m is not None else self._nnz

    @property
    def ndim(self):
        return 2

    @property
    def materialized(self) -> bool:
        return self._m is not None

    # ---- the plain matrix ----------------------------------------------------------------
    def tocsr(self, copy: bool = False):
        if self._m is None:
            m = self._ctx.matrix(self._which)
            self._m = self._post(m) if self._post is not None else m
            self._ctx = None
        return self._m.copy() if copy else self._m

    def _detach(self):
        """Called by the handle before it overwrites the values this proxy stands for."""
        if self._m is None and self._ctx is not None:
            self.tocsr()

    # ---- cheap paths -----------------------------------------------------------------------
    def __getitem__(self, key):
        if self._m is None and self._post is None and not isinstance(key, tuple):
            rows = np.atleast_1d(np.asarray(key))
            if rows.dtype == bool:
                rows = np.flatnonzero(rows)
            if rows.ndim == 1 and np.issubdtype(rows.dtype, np.integer):
                rows = np.where(rows < 0, rows + self._shape[0], rows)
                return self._ctx.matrix_rows(self._which, rows)
        return self.tocsr()[key]

    def __matmul__(self, other):
        if self._m is None and self._post is None and isinstance(other, np.ndarray) and other.ndim == 1:
            return self._ctx.spmv(self._which, other)
        return self.tocsr() @ other

    def dot(self, other):
        return self.__matmul__(other)

    def __rmatmul__(self, other):
        return other @ self.tocsr()

    def __mul__(self, other):
        return self.tocsr() * other

    def __rmul__(self, other):
        return other * self.tocsr()

    def __neg__(self):
        return -self.tocsr()

    def __add__(self, other):
        return self.tocsr() + (other.tocsr() if isinstance(other, LazyCsr) else other)

    def __radd__(self, other):
        return other + self.tocsr()

    def __sub__(self, other):
        return self.tocsr() - (other.tocsr() if isinstance(other, LazyCsr) else other)

    def __rsub__(self, other):
        return other - self.tocsr()

    def __getattr__(self, name):
        # anything else (data, indices, indptr, T, tocsc, toarray, eliminate_zeros, ...): the plain matrix
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return getattr(self.tocsr(), name)

    def __repr__(self):
        state = "materialized" if self._m is not None else "on device"
        return f"<LazyCsr {self._shape[0]}x{self._shape[1]}, {self._nnz} stored entries, {state}>"


def detach_all(ctx) -> None:
    """Materialize every proxy of ``ctx`` that is still alive (before its values are overwritten)."""
    refs, ctx._lazy_refs = ctx._lazy_refs, []
    for r in refs:
        p = r()
        if p is not None:
            p._detach()
