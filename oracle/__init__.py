"""TEST INFRASTRUCTURE: CPU oracles, golden generators and the recipe-built reference archive.

Only ``tests/``, ``__graft_entry__.smoke()``, ``bench.py: cpu_baseline`` and ``tools/fuzz_*`` may import
anything from here; ``porepy_amd/`` never does.
"""
from __future__ import annotations

import os

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_LIVE = "/root/reference/src"
REF_ARCHIVE = os.path.join(_HERE, "_ref", "porepy_ref.zip")
SHIM = os.path.join(_HERE, "shim")


def ref_path(prefer_archive: bool = False):
    """``PYTHONPATH`` entries under which the REFERENCE PorePy imports in a subprocess: the import shim
    (stubs for absent third-party packages) + the live tree (build container) or the byte-compiled archive
    made by ``oracle/make_ref.py`` (the GPU box).  ``None`` where neither exists."""
    if not prefer_archive and os.path.isdir(os.path.join(REF_LIVE, "porepy")):
        return [SHIM, REF_LIVE]
    if os.path.exists(REF_ARCHIVE):
        from . import make_ref

        if make_ref.usable(REF_ARCHIVE):  # (written by another Python minor version: treated as absent)
            return [SHIM, REF_ARCHIVE]
    if os.path.isdir(os.path.join(REF_LIVE, "porepy")):
        return [SHIM, REF_LIVE]
    return None


def ref_env(extra_first=(), extra_last=(), prefer_archive: bool = False):
    """Environment for such a subprocess, or ``None``."""
    p = ref_path(prefer_archive)
    if p is None:
        return None
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join(list(extra_first) + p + list(extra_last))
    env["PYTHONDONTWRITEBYTECODE"] = "1"
    return env
