"""Lane-group width sweep for the one-item-per-group assembly kernels (PFV_SYMB_G, PFV_SYMB_CELL_G,
PFV_FACE_G, PFV_ASM_G): per-phase times of discretize + assemble on the benchmark grid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import porepy_amd as pa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 69
g, K, bc, bv, src = bench.make_problem(n)
ctx = pa.Context(0)
ctx.set_grid(pa.grid_to_raw(g))
ctx.set_params(K.values, pa.bc_flags(bc), None, 1.0 / 3.0)
ref = None
for knob in ("PFV_SYMB_G", "PFV_SYMB_CELL_G", "PFV_FACE_G", "PFV_ASM_G"):
    for G in (64, 32, 16):
        os.environ[knob] = str(G)
        best = None
        for rep in range(3):
            ctx.discretize(rebuild_topology=True)
            ctx.assemble(bv, None, src)
            ctx.sync()
            st = ctx.stats()
            row = (st["symbolic_ms"], st["node_ms"], st["face_ms"], st["assemble_ms"])
            best = row if best is None else tuple(min(a, b) for a, b in zip(best, row))
        chk = float(np.abs(ctx.rhs()).sum()) + float(abs(ctx.matrix(6).data).sum())
        ref = chk if ref is None else ref
        print(f"{knob}={G:2d}: symbolic {best[0]:6.2f} node {best[1]:6.2f} face {best[2]:6.2f} assemble {best[3]:6.2f} ms"
              f"   checksum {'same' if chk == ref else 'DIFFERENT %r' % (chk - ref)}", flush=True)
    os.environ[knob] = "64"
