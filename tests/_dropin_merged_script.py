"""Run inside a subprocess with the REFERENCE PorePy importable: ``MergedOperator.parse`` and the first products of the
flux expression ON THE DEVICE (porepy_amd.DeviceCsr, csrc/csr_algebra.inc) against the reference's own scipy results on
its mixed-dimensional single-phase model (3-D box, two intersecting fractures, their intersection line, mortars).

  1. the reference discretizes; ``MpfaAd(...).flux().parse(mdg)`` etc. are its block-diagonal concatenations
     (numerics/ad/ad_utils.py:597-663).  The same per-subdomain matrices uploaded and concatenated by
     ``porepy_amd.block_diag`` must give the same matrix, bit for bit;
  2. every subdomain of dimension >= 2 is discretized again by ``porepy_amd.Mpfa(lazy=True)``: its matrices stay on the
     device; ``merged_matrix`` concatenates them (device to device) with the 1-D / 0-D ones of the reference; the result
     equals the reference's to the parity tolerance of the discretization (1e-10);
  3. the Jacobian blocks of the mass balance that are pure matrix products -- d(div q)/dp = Div Flux and
     d(div q)/d(lambda) = Div BoundFlux P_mortar->primary -- are formed on the device and compared with scipy's on the
     reference's matrices."""
import json

import numpy as np
import scipy.sparse as sps

import porepy as pp

import porepy_amd as pa
from tests import _parity as P
from tests._dropin_md_script_model import Model

KEYS = ("flux", "bound_flux", "vector_source", "bound_pressure_cell", "bound_pressure_face")
lib = P.dropin_library()

m = Model({"times_to_export": [], "darcy_flux_discretization": "mpfa"})
m.prepare_simulation()
mdg = m.mdg
subdomains, interfaces = mdg.subdomains(), mdg.interfaces()
kw = m.darcy_keyword
discr = m.darcy_flux_discretization(subdomains)
ref = {k: sps.csr_matrix(getattr(discr, k)().parse(mdg)) for k in KEYS}
for v in ref.values():
    v.sort_indices()
div_ref = sps.csr_matrix(pp.ad.Divergence(subdomains).parse(mdg))
proj = pp.ad.MortarProjections(mdg, subdomains, interfaces)
m2p_ref = sps.csr_matrix(proj.mortar_to_primary_int().parse(mdg))
datas = [mdg.subdomain_data(sd) for sd in subdomains]

ctx = pa.Context(0, lib)
out = {"subdomains": len(subdomains), "dims": sorted({sd.dim for sd in subdomains}, reverse=True),
       "faces": int(ref["flux"].shape[0]), "cells": int(ref["flux"].shape[1]), "mortar_cells": int(m2p_ref.shape[1])}

# 1. the reference's own blocks through the device concatenation: identical to MergedOperator.parse
bit = True
for k in KEYS:
    blocks = [d[pp.DISCRETIZATION_MATRICES][kw][getattr(discr._discretization, k + "_matrix_key")] for d in datas]
    M = pa.block_diag(blocks, ctx).to_scipy()
    r = ref[k]
    bit = bit and M.shape == r.shape and np.array_equal(M.indptr, r.indptr) and np.array_equal(M.indices, r.indices) \
        and np.array_equal(M.data, r.data)
out["block_diag_bit_identical_to_reference_parse"] = bool(bit)

# 2. device-resident discretization matrices of the subdomains of dimension >= 2
hip = pa.Mpfa(kw, library=lib, lazy=True)
mine = []
device_resident = 0
for sd, d in zip(subdomains, datas):
    if sd.dim >= 2:
        dd = {pp.PARAMETERS: d[pp.PARAMETERS], pp.DISCRETIZATION_MATRICES: {kw: {}}}
        hip.discretize(sd, dd)
        mine.append(dd)
        device_resident += 1
    else:
        mine.append(d)
out["subdomains_resident_on_device"] = device_resident
merged = {k: pa.merged_matrix(mine, kw, k, ctx) for k in KEYS}
kinds = {}
for sd, dd in zip(subdomains, mine):
    if sd.dim >= 2:
        for k in KEYS:
            mm = dd[pp.DISCRETIZATION_MATRICES][kw][k]
            on_device = isinstance(mm, pa.lazy.LazyCsr) and not mm.materialized
            kinds.setdefault(f"dim{sd.dim}", {})[k] = "device" if on_device else type(mm).__name__
out["where_the_blocks_were"] = kinds
err = {}
for k in KEYS:
    M = merged[k].to_scipy()
    err[k] = float(abs(M - ref[k]).max() / max(abs(ref[k]).max(), 1e-300))
out["merged_rel_err"] = err

# 3. products of the flux expression on the device
div = pa.DeviceCsr.from_scipy(div_ref, ctx)
m2p = pa.DeviceCsr.from_scipy(m2p_ref, ctx)
J_pp = (div @ merged["flux"]).to_scipy()
J_pl = (div @ (merged["bound_flux"] @ m2p)).to_scipy()
R_pp = div_ref @ ref["flux"]
R_pl = div_ref @ (ref["bound_flux"] @ m2p_ref)
out["J_pp_rel_err"] = float(abs(J_pp - R_pp).max() / abs(R_pp).max())
out["J_pl_rel_err"] = float(abs(J_pl - R_pl).max() / abs(R_pl).max())
out["J_pp_shape"] = list(J_pp.shape)
out["J_pl_nnz"] = int(J_pl.nnz)
# the flux of a pressure / mortar-flux state
rng = np.random.default_rng(0)
p, lam = rng.random(ref["flux"].shape[1]), rng.random(m2p_ref.shape[1])
q = merged["flux"] @ p + merged["bound_flux"] @ (m2p @ lam)
q_ref = ref["flux"] @ p + ref["bound_flux"] @ (m2p_ref @ lam)
out["flux_rel_err"] = float(abs(q - q_ref).max() / abs(q_ref).max())
out["library"] = str(lib._name)
print("RESULT " + json.dumps(out))
