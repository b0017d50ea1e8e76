import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import porepy_amd as pa
from tests._golden import Case
from tests import _parity as P
lib = pa._lib.product_library()
emu = P.emulation_library()
for name in ["cart2d_4x3_mixed", "tet_2x2x2_dir_generic"]:
    c = Case(name)
    g = P.run_case(lib, c); e = P.run_case(emu, c)
    for w, nm in ((0, "Ainv"), (1, "T")):
        a, b = g.debug_array(w), e.debug_array(w)
        d = np.abs(a - b)
        bad = np.flatnonzero(d > 1e-9 * np.abs(b).max())
        print(name, nm, "len", a.size, "nbad", bad.size, "first bad idx", bad[:12], "gpu", a[bad[:6]], "emu", b[bad[:6]])
