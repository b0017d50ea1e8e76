"""Recipe for ``oracle/_ref/``: the REFERENCE PorePy byte-compiled where it lies.

TEST INFRASTRUCTURE.  ``python oracle/make_ref.py`` compiles every module of
``/root/reference/src/porepy`` with ``py_compile`` straight from the reference tree into
``oracle/_ref/porepy_ref.zip`` (sourceless ``.pyc`` members — a binary built from the sources where
they lie, like a ``.so``; no reference source file is copied, the directory is git-ignored and only
travels to the GPU box with the gpurun snapshot).  Users of the archive:

* ``bench.py: cpu_baseline``   -> ``kind: "reference"``: ``pp.Mpfa.discretize`` + ``assemble_matrix_rhs``
  + the linear solve of the reference timed on the GPU box's host cores;
* ``tests/test_reference_dropin.py`` (``-m gpu``): the reference's own models with ``pp.Mpfa`` /
  ``pp.Mpsa`` / ``pp.Biot`` rebound to ``libporefv_hip.so``;
* ``tools/fuzz_vs_reference.py`` with ``PFV_FUZZ_DEVICE=1``.

The product (``porepy_amd/``) never imports it.  ``oracle.ref_path()`` gives the ``PYTHONPATH``
entries (import shim + reference) for a subprocess: the live tree when ``/root/reference`` exists
(the build container), otherwise the archive, otherwise ``None``.
"""
from __future__ import annotations

import os
import py_compile
import sys
import tempfile
import zipfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/src"
OUT_DIR = os.path.join(HERE, "_ref")
ARCHIVE = os.path.join(OUT_DIR, "porepy_ref.zip")
# modules of the reference's own test-suite that a drop-in test subclasses (archive member name -> source)
EXTRA = {"reference_test_tpfa.pyc": "/root/reference/tests/numerics/fv/test_tpfa.py"}


def build(force: bool = False) -> str | None:
    pkg = os.path.join(REF_SRC, "porepy")
    if not os.path.isdir(pkg):
        return ARCHIVE if os.path.exists(ARCHIVE) else None
    newest = 0.0
    files = []
    for d, _, names in os.walk(pkg):
        for n in names:
            if n.endswith(".py"):
                p = os.path.join(d, n)
                files.append(p)
                newest = max(newest, os.path.getmtime(p))
    if not force and os.path.exists(ARCHIVE) and os.path.getmtime(ARCHIVE) >= newest:
        return ARCHIVE
    os.makedirs(OUT_DIR, exist_ok=True)
    tmp = ARCHIVE + ".tmp"
    with tempfile.TemporaryDirectory() as scratch, zipfile.ZipFile(tmp, "w", zipfile.ZIP_DEFLATED) as z:
        for i, src in enumerate(sorted(files)):
            rel = os.path.relpath(src, REF_SRC)
            cfile = os.path.join(scratch, f"{i}.pyc")
            # dfile: what tracebacks show; unchecked-hash pyc so that the archive does not depend on mtimes
            py_compile.compile(src, cfile=cfile, dfile="<reference>/" + rel, doraise=True,
                               invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
            z.write(cfile, rel[:-3] + ".pyc")
        for j, (member, src) in enumerate(sorted(EXTRA.items())):
            cfile = os.path.join(scratch, f"x{j}.pyc")
            py_compile.compile(src, cfile=cfile, dfile="<reference>/" + os.path.relpath(src, "/root/reference"),
                               doraise=True, invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
            z.write(cfile, member)
    os.replace(tmp, ARCHIVE)
    return ARCHIVE


if __name__ == "__main__":
    out = build(force="--force" in sys.argv)
    print(out, os.path.getsize(out) if out else "")
