import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import porepy_amd as pa
from oracle import mpfa_oracle as mo
from tests._golden import Case, ALL_KEYS, rel_max_err
from tests import _parity as P
lib = pa._lib.product_library()
for name in ["cart2d_4x3_mixed", "cart3d_3x3x3_hetero", "tri2d_4x4_mixed", "tet_2x2x2_dir_generic", "tet_3x3x3_mixed_aniso", "tet_4x4x4_iso_linear"]:
    c = Case(name)
    try:
        ctx = P.run_case(lib, c)
    except Exception as e:
        print(name, "EXC", e); continue
    ora = mo.discretize(c.grid, c.perm, c.bc, eta=c.eta)
    errs = {k: rel_max_err(ctx.matrix(P.WHICH[k]), ora[k]) for k in ALL_KEYS}
    print(name, {k: f"{v:.1e}" for k, v in errs.items()}, ctx.stats()["max_block"])
