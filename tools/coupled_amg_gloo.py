"""Iteration counts of the sharded solve under gloo (host emulation build): coupled hierarchy vs block Jacobi.

    python tools/coupled_amg_gloo.py [n_side] [worlds...]      e.g.  python tools/coupled_amg_gloo.py 12 1 2 4
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


from tests import _sharded_cases as S  # noqa: E402


def main():
    import tempfile

    n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    worlds = [int(a) for a in sys.argv[2:]] or [1, 2, 4]
    ref = None
    for world in worlds:
        with tempfile.TemporaryDirectory() as out:
            res = S.run(world, n_side, out)
        if ref is None:
            ref = res["amg"]["x"]
        for precond in res:
            x = res[precond].pop("x")
            res[precond]["rel_diff_vs_first"] = float(np.linalg.norm(x - ref) / np.linalg.norm(ref))
        print(json.dumps({"n_side": n_side, "cells": int(ref.size), "world": world, **res}))


if __name__ == "__main__":
    main()
