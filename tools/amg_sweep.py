"""AMG robustness sweep on the GPU (tests/_parity.py: amg_robustness_sweep)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import _parity as P
import porepy_amd as pa
out = P.amg_robustness_sweep(pa._lib.product_library())
for k, v in out.items():
    print(f"{k:28s} unknowns {v[0]:8d}  iterations {v[1]:4d}  true residual {v[2]:.2e}")
