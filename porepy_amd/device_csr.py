"""Device-resident sparse matrices: the algebra of the step AFTER ``discretize``.

In the reference a Newton iteration on a fixed grid re-does, on the host, what ``MergedOperator.parse``
(``numerics/ad/ad_utils.py:597-663``: the matrices of all subdomains concatenated block-diagonally through
``matrix_operations.csr_matrix_from_sparse_blocks``) and the operator tree above it spell out in scipy: products with
projections and divergences, diagonal scalings, sums (``EquationSystem.assemble``, ``numerics/ad/equation_system.py:1579``).
A ``DeviceCsr`` is such a matrix kept in HBM behind the C ABI (``pfv_csr_*``, ``csrc/csr_algebra.inc``): made from the
discretization matrices of a handle without a host copy, combined with ``@``, ``+``, ``-``, scalar ``*``,
``block_diag`` and row / column scalings, multiplied with vectors, handed to the device solver as its system --
a discretization matrix never crosses PCIe.  The results follow scipy's conventions bit for bit (sorted rows,
products and sums accumulated in scipy's order, exact zeros dropped), which is how the tests pin them.

The operator TREE is walked by the reference's own parser (``numerics/ad/_ad_parser.py``): a ``DeviceCsr`` offers
exactly what ``forward_mode.AdArray`` asks of its Jacobian -- ``shape``, ``astype``, row selection ``J[dofs]``,
``sps.diags(v) * J`` (row scaling), ``J + K``, ``-J``, ``scalar * J``, ``M @ J`` for a scipy ``M`` -- so an ``AdArray``
whose Jacobian lives on the device flows through the reference's operator arithmetic unchanged (``porepy_amd.ad``:
``assemble_on_device``).  Operands that arrive as scipy matrices (divergences, projections, discretization matrices of
an eager ``Mpfa``) are uploaded once and remembered while the host object lives."""
from __future__ import annotations

import ctypes as C
import weakref

import numpy as np

from . import _lib


try:  # (offline wheelhouse of the image; ~10 GB/s on one host core)
    from xxhash import xxh3_64_intdigest as _digest64
except ImportError:  # pragma: no cover - the fallback digests the same bytes, just slower
    import zlib

    def _digest64(buf):
        return (zlib.crc32(buf) << 32) | zlib.adler32(buf)


def _content_key(m):
    """What a remembered upload of the host matrix ``m`` is valid for: shape, nnz and a 64-bit digest of ALL of its
    VALUES (one pass over the buffer without a copy), so that a matrix changed in place (``m.data[:] = ...``,
    ``m *= s``, a single entry) is uploaded again instead of being served from the stale device copy.
    ``DeviceCsr.invalidate(m)`` forgets one matrix explicitly."""
    data = getattr(m, "data", None)
    if not isinstance(data, np.ndarray) or data.ndim != 1:
        return (getattr(m, "shape", None), getattr(m, "nnz", None), None)
    buf = data if data.flags.c_contiguous else np.ascontiguousarray(data)
    return (m.shape, m.nnz, data.dtype.str, _digest64(memoryview(buf).cast("B")))


def clear_upload_cache() -> None:
    """Forget every remembered upload of a host matrix (and release the HBM of the device copies nobody else holds)."""
    _UPLOADS.clear()


class DeviceCsr:
    """One CSR matrix (FP64 values, int32 indices) resident on the device of ``context``."""

    __array_priority__ = 20.0

    def __init__(self, context: "_lib.Context", handle):
        self.ctx = context
        self._c = handle
        refs = context._csr_refs
        if len(refs) >= 64 and len(refs) % 64 == 0:  # drop the entries of matrices that are gone (temporaries of @, +)
            refs[:] = [r for r in refs if r() is not None]
        refs.append(weakref.ref(self))

    # ---- construction ------------------------------------------------------------------------------------------
    @classmethod
    def from_scipy(cls, A, context: "_lib.Context") -> "DeviceCsr":
        import scipy.sparse as sps

        A = sps.csr_matrix(A)
        if not A.has_canonical_format:
            A = A.copy()
            A.sum_duplicates()
        if A.nnz >= 2 ** 31:
            raise ValueError("more than 2^31 matrix entries")
        ip = np.ascontiguousarray(A.indptr, dtype=np.int32)
        ix = np.ascontiguousarray(A.indices, dtype=np.int32)
        dv = np.ascontiguousarray(A.data, dtype=np.float64)
        out = _lib._h()
        context._check(context.lib.pfv_csr_from_host(context._h, A.shape[0], A.shape[1], _lib._ptr(ip, _lib._ip),
                                                     _lib._ptr(ix, _lib._ip), _lib._ptr(dv, _lib._dp), C.byref(out)))
        return cls(context, out)

    @classmethod
    def from_discretization(cls, source: "_lib.Context", which: int, context: "_lib.Context | None" = None) -> "DeviceCsr":
        """Device-to-device copy of discretization matrix ``which`` (``_lib.MAT_*``) of ``source``."""
        context = source if context is None else context
        out = _lib._h()
        context._check(context.lib.pfv_csr_from_matrix(context._h, source._h, int(which), C.byref(out)))
        return cls(context, out)

    @staticmethod
    def invalidate(m) -> None:
        """Forget the remembered upload of the host matrix ``m`` (the next use uploads it again)."""
        _UPLOADS.pop(id(m), None)

    @classmethod
    def from_any(cls, m, context: "_lib.Context") -> "DeviceCsr":
        """A ``DeviceCsr`` as it is; a lazily fetched discretization matrix (``lazy.LazyCsr``) that has not left the
        device yet by a device-to-device copy; anything scipy understands by upload."""
        from .lazy import LazyCsr

        if isinstance(m, DeviceCsr):
            return m
        if not isinstance(m, LazyCsr):
            # an operand that lives on the host (divergence, projection, eager discretization matrix): uploaded once,
            # remembered while the host object is alive and unchanged in size
            hit = _UPLOADS.get(id(m))
            key = _content_key(m)
            if hit is not None and hit[0]() is m and hit[1].ctx is context and hit[1]._c and hit[2] == key:
                return hit[1]
            d = cls.from_scipy(m, context)
            try:
                _UPLOADS[id(m)] = (weakref.ref(m, lambda _r, k=id(m): _UPLOADS.pop(k, None)), d, key)
            except TypeError:
                pass
            return d
        if isinstance(m, LazyCsr) and not m.materialized and m._ctx is not None:
            if m._post is None:
                return cls.from_discretization(m._ctx, m._which, context)
            if getattr(m, "_right", None) is not None:
                # (the lift of a fracture plane's vector-source columns into the ambient space: a product on the device)
                return cls.from_discretization(m._ctx, m._which, context) @ cls.from_scipy(m._right, context)
        return cls.from_scipy(m.tocsr() if isinstance(m, LazyCsr) else m, context)

    # ---- facts -------------------------------------------------------------------------------------------------
    def _info(self):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        self.ctx._check(self.ctx.lib.pfv_csr_info(self._c, C.byref(a), C.byref(b), C.byref(c)))
        return int(a.value), int(b.value), int(c.value)

    @property
    def shape(self):
        r, c, _ = self._info()
        return (r, c)

    @property
    def nnz(self) -> int:
        return self._info()[2]

    ndim = 2

    def astype(self, dtype):
        """(values are always FP64)"""
        return self

    def __getitem__(self, key) -> "DeviceCsr":
        """Row selection ``J[rows]`` (index array, boolean mask, slice or one index): the product with the selection
        matrix, on the device -- how ``AdArray.__getitem__`` restricts a Jacobian to the dofs of a variable."""
        import scipy.sparse as sps

        n = self.shape[0]
        if isinstance(key, tuple):
            raise IndexError("only rows can be selected")
        rows = np.arange(n)[key]
        rows = np.atleast_1d(rows).astype(np.int64)
        S = sps.csr_matrix((np.ones(rows.size), rows, np.arange(rows.size + 1)), shape=(rows.size, n))
        return DeviceCsr.from_scipy(S, self.ctx) @ self

    @staticmethod
    def _diagonal_of(m):
        """The diagonal of a scipy diagonal matrix (``sps.diags(v)``), or None."""
        import scipy.sparse as sps

        if sps.issparse(m) and m.format == "dia" and m.shape[0] == m.shape[1] and len(m.offsets) == 1 and m.offsets[0] == 0:
            return np.asarray(m.diagonal(), dtype=np.float64)
        return None

    # ---- algebra -----------------------------------------------------------------------------------------------
    def _binary(self, fn, *args):
        out = _lib._h()
        self.ctx._check(fn(self.ctx._h, *args, C.byref(out)))
        return DeviceCsr(self.ctx, out)

    def __matmul__(self, other):
        import scipy.sparse as sps

        if isinstance(other, DeviceCsr):
            return self._binary(self.ctx.lib.pfv_csr_matmul, self._c, other._c)
        if sps.issparse(other):
            d = self._diagonal_of(other)
            if d is not None:
                return self.scaled(cols=d)
            return self._binary(self.ctx.lib.pfv_csr_matmul, self._c, DeviceCsr.from_any(other, self.ctx)._c)
        if hasattr(other, "val") and hasattr(other, "jac"):
            # a forward-mode AD array of the reference (numerics/ad/forward_mode.py:565-592: ``AdArray.__rmatmul__`` only
            # takes scipy matrices): value by the device SpMV, Jacobian by the device product
            return type(other)(self @ np.asarray(other.val, dtype=np.float64), self @ other.jac)
        x = np.asarray(other)
        if x.ndim != 1 or x.shape[0] != self.shape[1]:
            raise ValueError("DeviceCsr @ x expects a vector of matching length (or another DeviceCsr)")
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty(self.shape[0])
        self.ctx._check(self.ctx.lib.pfv_csr_spmv(self._c, _lib._ptr(x, _lib._dp), _lib._ptr(y, _lib._dp)))
        return y

    dot = __matmul__

    def matvec_device(self, x_ptr: int, y_ptr: int):
        """y = A x on device vectors (raw addresses), on the stream of the owning handle."""
        self.ctx._check(self.ctx.lib.pfv_csr_spmv_device(self._c, C.c_void_p(x_ptr), C.c_void_p(y_ptr)))

    def axpby(self, alpha: float, other: "DeviceCsr", beta: float) -> "DeviceCsr":
        """alpha * self + beta * other."""
        return self._binary(self.ctx.lib.pfv_csr_axpby, float(alpha), self._c, float(beta), other._c)

    def __rmatmul__(self, other):
        """``M @ J`` for a scipy ``M`` (scipy's own ``@`` returns NotImplemented for a foreign right operand)."""
        d = self._diagonal_of(other)
        if d is not None:
            return self.scaled(rows=d)
        return DeviceCsr.from_any(other, self.ctx) @ self

    def __add__(self, other):
        return self.axpby(1.0, DeviceCsr.from_any(other, self.ctx), 1.0)

    def __radd__(self, other):
        return DeviceCsr.from_any(other, self.ctx).axpby(1.0, self, 1.0)

    def __sub__(self, other):
        return self.axpby(1.0, DeviceCsr.from_any(other, self.ctx), -1.0)

    def __rsub__(self, other):
        return DeviceCsr.from_any(other, self.ctx).axpby(1.0, self, -1.0)

    @property
    def T(self) -> "DeviceCsr":
        """The transpose (rows sorted: what ``A.T.tocsr()`` gives in scipy)."""
        return self._binary(self.ctx.lib.pfv_csr_transpose, self._c)

    transpose = lambda self: self.T  # noqa: E731

    def copy(self) -> "DeviceCsr":
        return self.scaled()

    def scaled(self, rows=None, cols=None) -> "DeviceCsr":
        """diag(rows) @ self @ diag(cols) as a new matrix (either may be None)."""
        out = block_diag([self])
        r = None if rows is None else np.ascontiguousarray(rows, dtype=np.float64)
        c = None if cols is None else np.ascontiguousarray(cols, dtype=np.float64)
        if r is not None and r.shape != (self.shape[0],) or c is not None and c.shape != (self.shape[1],):
            raise ValueError("scaling vector of the wrong length")
        if r is not None or c is not None:
            self.ctx._check(self.ctx.lib.pfv_csr_scale(out._c, None if r is None else _lib._ptr(r, _lib._dp),
                                                       None if c is None else _lib._ptr(c, _lib._dp)))
        return out

    def __mul__(self, alpha):
        """scalar * J; ``J * sps.diags(v)`` (column scaling, the old-style matrix product of scipy's ``*``).  Entries that
        become exact zeros stay stored (scipy's scalar product keeps them too)."""
        if np.isscalar(alpha):
            return self.scaled(rows=np.full(self.shape[0], float(alpha)))
        d = self._diagonal_of(alpha)
        if d is not None:
            return self.scaled(cols=d)
        return NotImplemented

    def __rmul__(self, alpha):
        """scalar * J; ``sps.diags(v) * J`` (row scaling: ``AdArray._diagvec_mul_jac``)."""
        if np.isscalar(alpha):
            return self.scaled(rows=np.full(self.shape[0], float(alpha)))
        d = self._diagonal_of(alpha)
        if d is not None:
            return self.scaled(rows=d)
        import scipy.sparse as sps

        if sps.issparse(alpha):
            return self.__rmatmul__(alpha)
        return NotImplemented

    def __truediv__(self, alpha):
        if not np.isscalar(alpha):
            return NotImplemented
        out = block_diag([self])
        self.ctx._check(self.ctx.lib.pfv_csr_divide(out._c, float(alpha)))
        return out

    def __neg__(self):
        return self * -1.0

    # ---- leaving the device ------------------------------------------------------------------------------------
    def to_scipy(self):
        import scipy.sparse as sps

        nr, nc, nnz = self._info()
        ip = np.empty(nr + 1, dtype=np.int32)
        ix = np.empty(nnz, dtype=np.int32)
        dv = np.empty(nnz, dtype=np.float64)
        self.ctx._check(self.ctx.lib.pfv_csr_get(self._c, _lib._ptr(ip, _lib._ip), _lib._ptr(ix, _lib._ip),
                                                 _lib._ptr(dv, _lib._dp)))
        return sps.csr_matrix((dv, ix, ip), shape=(nr, nc))

    tocsr = to_scipy

    def as_system(self, rhs, context: "_lib.Context | None" = None, rhs_device_ptr: int | None = None):
        """Make (self, rhs) the active system of ``context`` (default: the owning handle): ``context.solve(...)`` then
        works on it.  The matrix moves device-to-device; ``rhs`` is a host array unless ``rhs_device_ptr`` is given."""
        ctx = self.ctx if context is None else context
        n = self.shape[0]
        if rhs_device_ptr is not None:
            ctx._check(ctx.lib.pfv_csr_set_system(ctx._h, self._c, C.c_void_p(rhs_device_ptr), 1))
        else:
            b = np.ascontiguousarray(rhs, dtype=np.float64)
            if b.shape != (n,):
                raise ValueError("right-hand side of the wrong length")
            ctx._check(ctx.lib.pfv_csr_set_system(ctx._h, self._c, b.ctypes.data_as(C.c_void_p), 0))
        ctx._user_n = n
        return ctx

    def close(self):
        if getattr(self, "_c", None) is not None and self._c and getattr(self.ctx, "_h", None):
            self.ctx.lib.pfv_csr_free(self._c)
        self._c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __repr__(self):
        r, c, z = self._info()
        return f"<DeviceCsr {r}x{c}, {z} stored entries, on device>"


_UPLOADS: dict = {}  # id(host matrix) -> (weakref to it, its DeviceCsr, (shape, nnz)); see DeviceCsr.from_any


def vstack(mats, context: "_lib.Context | None" = None) -> DeviceCsr:
    """``scipy.sparse.vstack`` on the device (the rows of the equations of a model, one block per equation)."""
    return bmat([[m] for m in mats], context)


def block_diag(mats, context: "_lib.Context | None" = None) -> DeviceCsr:
    """Block-diagonal concatenation on the device: ``matrix_operations.csr_matrix_from_sparse_blocks`` of the reference,
    i.e. what ``MergedOperator.parse`` returns for the discretization matrices of a list of subdomains."""
    mats = list(mats)
    if context is None:
        context = next((m.ctx for m in mats if isinstance(m, DeviceCsr)), None)
        if context is None:
            raise ValueError("block_diag of host matrices needs a context")
    dm = [DeviceCsr.from_any(m, context) for m in mats]
    arr = (_lib._h * max(len(dm), 1))(*[m._c for m in dm])
    out = _lib._h()
    context._check(context.lib.pfv_csr_block_diag(context._h, len(dm), arr, C.byref(out)))
    return DeviceCsr(context, out)


def bmat(blocks, context: "_lib.Context | None" = None) -> DeviceCsr:
    """``scipy.sparse.bmat`` on the device: a block matrix from a grid (list of rows) of blocks, ``None`` = zero block.
    Stacking the equations of several variables -- the last step of ``EquationSystem.assemble`` -- without the pieces
    leaving HBM.  Every block row and every block column needs at least one block to take its size from."""
    rows = [list(r) for r in blocks]
    nbr, nbc = len(rows), len(rows[0])
    if any(len(r) != nbc for r in rows):
        raise ValueError("ragged block grid")
    if context is None:
        context = next((m.ctx for r in rows for m in r if isinstance(m, DeviceCsr)), None)
        if context is None:
            raise ValueError("bmat of host matrices needs a context")
    dm = [[None if m is None else DeviceCsr.from_any(m, context) for m in r] for r in rows]
    rs, cs = [None] * nbr, [None] * nbc
    for i in range(nbr):
        for j in range(nbc):
            if dm[i][j] is not None:
                sh = dm[i][j].shape
                rs[i] = sh[0] if rs[i] is None else rs[i]
                cs[j] = sh[1] if cs[j] is None else cs[j]
    if any(v is None for v in rs + cs):
        raise ValueError("a block row or column without any block: its size is unknown")
    arr = (_lib._h * (nbr * nbc))(*[(m._c if m is not None else None) for r in dm for m in r])
    rsa, csa = np.asarray(rs, dtype=np.int64), np.asarray(cs, dtype=np.int64)
    out = _lib._h()
    context._check(context.lib.pfv_csr_bmat(context._h, nbr, nbc, arr, _lib._ptr(rsa, _lib._lp), _lib._ptr(csa, _lib._lp),
                                            C.byref(out)))
    return DeviceCsr(context, out)


def merged_matrix(discretization_data, keyword: str, matrix_key: str, context: "_lib.Context") -> DeviceCsr:
    """``MergedOperator.parse`` on the device: the matrices ``data[DISCRETIZATION_MATRICES][keyword][matrix_key]`` of a
    list of data dictionaries (one per subdomain, in the order of the operator's domains) as one block-diagonal
    ``DeviceCsr``.  Lazily fetched matrices (``Mpfa(keyword, lazy=True)``) are copied device-to-device."""
    from .params import DISCRETIZATION_MATRICES

    return block_diag([d[DISCRETIZATION_MATRICES][keyword][matrix_key] for d in discretization_data], context)
