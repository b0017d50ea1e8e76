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

import importlib.util
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
# ... and the modules of it that tests/test_reference_suite.py runs under the rebound operators on the GPU box
# (package ``reference_tests`` of the archive; tests/_reference_suite_runner.py star-imports them into stubs)
SUITE = ["numerics/fv/test_mpfa", "numerics/fv/test_mpsa", "numerics/fv/test_biot", "numerics/fv/test_tpfa",
         "numerics/fv/test_fvutils", "models/test_fluid_mass_balance", "models/test_momentum_balance",
         "models/test_poromechanics"]
for _m in SUITE:
    EXTRA["reference_tests/" + os.path.basename(_m) + ".pyc"] = "/root/reference/tests/" + _m + ".py"


# the reference's source files of the hot path: their digest keys records of reference RUNS that are
# too long to repeat in every bench invocation (bench.py: cached_cpu_headline)
PATH_FILES = ["numerics/fv/mpfa.py", "numerics/fv/_fvutils.py", "numerics/fv/fv_elliptic.py", "numerics/fv/mpsa.py",
              "numerics/fv/tpfa.py", "numerics/fv/biot.py", "numerics/linalg/matrix_operations.py",
              "numerics/discretization.py"]


def _live_digest() -> str | None:
    import hashlib

    h = hashlib.sha256()
    for rel in PATH_FILES:
        p = os.path.join(REF_SRC, "porepy", rel)
        if not os.path.exists(p):
            return None
        h.update(rel.encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def source_digest(archive: str = ARCHIVE) -> str | None:
    """Digest of the reference's hot-path sources: of the live tree where it exists, else the one recorded in the
    archive when it was built."""
    d = _live_digest()
    if d is not None:
        return d
    try:
        with zipfile.ZipFile(archive) as z:
            return z.read("SOURCE_DIGEST").decode().strip()
    except (OSError, KeyError, zipfile.BadZipFile):
        return None


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
    newest = max([newest, os.path.getmtime(os.path.abspath(__file__))] +
                 [os.path.getmtime(p) for p in EXTRA.values() if os.path.exists(p)])
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
        init_src = os.path.join(scratch, "empty_init.py")
        open(init_src, "w").close()
        init_pyc = os.path.join(scratch, "empty_init.pyc")
        py_compile.compile(init_src, cfile=init_pyc, dfile="<recipe>/reference_tests/__init__.py", doraise=True,
                           invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
        z.write(init_pyc, "reference_tests/__init__.pyc")
        # the interpreter that wrote the members: an archive is only usable by the same bytecode magic
        z.writestr("PYC_MAGIC", importlib.util.MAGIC_NUMBER.hex())
        z.writestr("SOURCE_DIGEST", _live_digest() or "")
    os.replace(tmp, ARCHIVE)
    return ARCHIVE


def usable(archive: str = ARCHIVE) -> bool:
    """True when ``archive`` was written by an interpreter with this interpreter's bytecode magic (sourceless
    ``.pyc`` members do not import under another one)."""
    try:
        with zipfile.ZipFile(archive) as z:
            return z.read("PYC_MAGIC").decode().strip() == importlib.util.MAGIC_NUMBER.hex()
    except (OSError, KeyError, zipfile.BadZipFile):
        return False


if __name__ == "__main__":
    out = build(force="--force" in sys.argv)
    print(out, os.path.getsize(out) if out else "")
