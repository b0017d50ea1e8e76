"""Run modules of the REFERENCE's own test-suite, untouched or with ``pp.Mpfa`` / ``pp.Mpsa`` / ``pp.Biot``
rebound to the operators of this package, and print the outcome of every test.

Executed in a subprocess by tests/test_reference_suite.py with the reference importable
(oracle.ref_env): ``python _reference_suite_runner.py <variant> <module> [<module> ...]`` where
``variant`` is ``untouched`` (the reference as it is), ``emulation`` (rebound, host-emulation build) or
``product`` (rebound, libporefv_hip.so).  A module is ``numerics/fv/test_mpfa`` etc.

Where the live tree ``/root/reference/tests`` exists its files are collected as they lie; on the GPU box the
modules come byte-compiled from ``oracle/_ref/porepy_ref.zip`` (members ``reference_tests/<name>.pyc``, made
by oracle/make_ref.py) and are collected through one-line stub modules that star-import them.

Prints ``RESULT {"outcomes": {"test_mpfa::TestX::test_y[p]": "passed" | "failed" | "skipped" | "error"},
"library": ..., "device_calls": {...}}``.
"""
import json
import os
import sys
import tempfile

import pytest

LIVE = "/root/reference/tests"


class Recorder:
    def __init__(self):
        self.outcomes = {}
        self.messages = {}

    @staticmethod
    def key(report) -> str:
        _, _, rest = report.nodeid.partition("::")
        return os.path.splitext(os.path.basename(str(report.location[0])))[0] + "::" + rest

    def pytest_runtest_logreport(self, report):
        k = self.key(report)
        if report.when == "call":
            self.outcomes[k] = report.outcome
        elif report.outcome == "failed":  # setup / teardown error
            self.outcomes[k] = "error"
        elif report.when == "setup" and report.outcome == "skipped":
            self.outcomes[k] = "skipped"
        if report.outcome == "failed":
            self.messages[k] = str(report.longrepr)[-600:]


class Rebind:
    """``pp.Mpfa = HipMpfa`` etc. before any test module is imported (the classes are resolved on the
    ``porepy`` module at call time; parametrizations that name them are evaluated at import)."""

    def __init__(self, variant: str):
        self.variant = variant
        self.calls = {"mpfa": 0, "mpsa": 0, "biot": 0}
        self.library = None

    def pytest_configure(self, config):
        import porepy as pp

        import porepy_amd as pa
        from tests import _parity as P

        os.environ["PFV_DROPIN_LIBRARY"] = self.variant
        lib = P.dropin_library()
        self.library = str(lib._name)
        classes = {"mpfa": pa.as_porepy_discretization(library=lib), "mpsa": pa.as_porepy_mpsa(library=lib),
                   "biot": pa.as_porepy_biot(library=lib)}
        for key, cls in classes.items():
            orig = cls.discretize

            def counting(self_, sd, data, _o=orig, _k=key):
                self.calls[_k] += 1
                return _o(self_, sd, data)

            cls.discretize = counting
        pp.Mpfa, pp.Mpsa, pp.Biot = classes["mpfa"], classes["mpsa"], classes["biot"]


def main(argv):
    variant, modules = argv[0], argv[1:]
    scratch = tempfile.mkdtemp(prefix="refsuite_")
    targets = []
    for m in modules:
        live = os.path.join(LIVE, m + ".py")
        if os.path.exists(live) and os.environ.get("PFV_REFSUITE_ARCHIVE", "0") != "1":
            targets.append(live)
        else:
            name = os.path.basename(m)
            stub = os.path.join(scratch, name + ".py")
            with open(stub, "w") as f:
                f.write(f"from reference_tests.{name} import *  # noqa: F401,F403\n")
            targets.append(stub)
    rec = Recorder()
    plugins = [rec]
    rb = None
    if variant != "untouched":
        rb = Rebind(variant)
        plugins.append(rb)
    rc = pytest.main(["-q", "-o", "addopts=", "-p", "no:cacheprovider", "--rootdir=" + scratch,
                      "--import-mode=importlib", "-W", "ignore", "--tb=no", *targets], plugins=plugins)
    out = {"outcomes": rec.outcomes, "messages": rec.messages, "exit": int(rc),
           "library": rb.library if rb else None, "device_calls": rb.calls if rb else None}
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    main(sys.argv[1:])
