"""Import shim so the *reference* PorePy (under /root/reference/src) can be imported in
this container for oracle validation and golden-vector generation ONLY.

TEST INFRASTRUCTURE. Never imported by the product (porepy_amd/), never present on the
GPU box. Usage:
    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=oracle/shim:/root/reference/src python ...

It (i) back-fills typing names missing on py3.10 and (ii) installs a meta-path finder
that serves permissive stub modules for third-party packages that are absent here
(meshio, gmsh, shapely, deepdiff, seaborn, future, numba, pypardiso).  With the stub,
``@njit`` is a pass-through decorator and ``prange`` is ``range``.
"""
import importlib.abc
import importlib.machinery
import sys
import types
import typing

import typing_extensions

for _n in ("Self", "NotRequired", "Required", "override"):
    if not hasattr(typing, _n):
        setattr(typing, _n, getattr(typing_extensions, _n))

_ABSENT = {"meshio", "gmsh", "shapely", "deepdiff", "seaborn", "future", "numba", "pypardiso"}


class _Whatever:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        # bare decorator use: @njit
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return _Whatever()

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Whatever()

    def __getitem__(self, item):
        return _Whatever()

    def __iter__(self):
        return iter(())

    def __mro_entries__(self, bases):
        return (object,)


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        if name == "prange":
            return range
        return _Whatever()


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in _ABSENT:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        mod = _StubModule(spec.name)
        mod.__path__ = []
        return mod

    def exec_module(self, module):
        pass


sys.meta_path.insert(0, _StubFinder())
