"""A fixed-seed slice of the randomized differential tests of tools/fuzz_parity.py (host-emulation build): random
small grids, condition types and tensors -- kernels against the oracles, conditions per sub-face, discretization in
pieces and updates against one-piece / fresh discretizations, the solvers against a direct solve, Biot terms."""
import importlib.util
import os

import pytest

from tests import _parity as P

_spec = importlib.util.spec_from_file_location(
    "fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py"))
fuzz = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(fuzz)


@pytest.fixture(scope="module")
def lib():
    return P.emulation_library()


@pytest.mark.parametrize("mode", sorted(fuzz.MODES))
def test_fuzz_slice(lib, mode):
    assert fuzz.run_mode(lib, mode, 4, 2026, verbose=False) == 0
