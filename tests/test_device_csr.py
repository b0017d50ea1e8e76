"""Device-resident CSR algebra on the host-emulation library: see tests/_device_csr_cases.py."""
import pytest

from tests import _device_csr_cases as cases
from tests import _parity as P


@pytest.fixture(scope="module")
def lib():
    return P.emulation_library()


@pytest.mark.parametrize("case", [cases.random_algebra, cases.input_checks, cases.discretization_to_system,
                                  cases.merged_subdomains, cases.forward_mode_array_operand], ids=lambda f: f.__name__)
def test_device_csr(lib, case):
    case(lib)
