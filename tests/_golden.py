"""Loader for the committed golden fixtures (made by oracle/gen_golden.py from the reference)."""
from __future__ import annotations

import glob
import os

import numpy as np
import scipy.sparse as sps

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ALL_KEYS = (
    "flux",
    "bound_flux",
    "bound_pressure_cell",
    "bound_pressure_face",
    "vector_source",
    "bound_pressure_vector_source",
)


def case_names():
    names = [os.path.basename(p)[:-4] for p in sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))]
    return [n for n in names if n != "scalar_known_answers" and not n.startswith(("mpsa_", "mpsapartial_", "mpsasub_", "partial_", "tilted_", "tpfa_", "tpfaad_", "biot_", "subface_", "periodic_", "adflux_", "md_", "headline_", "persub_", "mpsawhole_", "biotwhole_", "mpsacontrast_"))]


def mpsa_case_names():
    names = [os.path.basename(p)[:-4] for p in sorted(glob.glob(os.path.join(GOLDEN_DIR, "mpsa_*.npz")))]
    return names


MPSA_KEYS = ("stress", "bound_stress", "bound_displacement_cell", "bound_displacement_face")


def mpsa_contrast_case_names():
    return [os.path.basename(p)[:-4] for p in sorted(glob.glob(os.path.join(GOLDEN_DIR, "mpsacontrast_*.npz")))]


class MpsaContrastCase:
    """MPSA fixture with Lame parameters of neighbouring cells apart by 1e8 ... 1e12 (oracle/gen_golden_mpsa_contrast.py):
    the matrices of the reference's ``pp.Mpsa``, the matrices of the reference's own local systems inverted in 60-digit
    arithmetic ("exact"), and how far the former are from the latter."""

    def __init__(self, name: str):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.name = name
        self.grid = {k[5:]: z[k] for k in z.files if k.startswith("grid_")}
        self.grid["dim"] = int(self.grid["dim"])
        self.grid["name"] = str(self.grid["name"])
        self.bc = {"is_dir": z["bc_is_dir"], "is_neu": z["bc_is_neu"]}
        self.stiffness = z["stiffness"]
        self.decades = float(z["decades"])
        self.ref, self.exact, self.ref_off_exact = {}, {}, {}
        for k in MPSA_KEYS:
            for tag, dst in (("ref", self.ref), ("exact", self.exact)):
                shape = tuple(int(v) for v in z[f"{tag}_{k}_shape"])
                dst[k] = sps.csr_matrix((z[f"{tag}_{k}_data"], z[f"{tag}_{k}_indices"], z[f"{tag}_{k}_indptr"]), shape=shape)
            self.ref_off_exact[k] = float(z[f"ref_off_exact_{k}"])


def mpsa_subface_case_names():
    return [os.path.basename(p)[:-4] for p in sorted(glob.glob(os.path.join(GOLDEN_DIR, "mpsasub_*.npz")))]


class MpsaSubfaceCase:
    """MPSA fixture with conditions per sub-face (oracle/gen_golden_mpsa_subface.py, from the reference)."""

    def __init__(self, name: str):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.name = name
        self.grid = {k[5:]: z[k] for k in z.files if k.startswith("grid_")}
        self.grid["dim"] = int(self.grid["dim"])
        self.grid["name"] = str(self.grid["name"])
        self.bc = {"is_dir": z["bc_is_dir"], "is_neu": z["bc_is_neu"], "is_rob": z["bc_is_rob"],
                   "robin_weight": z["bc_robin_weight"]}
        if "bc_basis" in z.files:  # (round 5: a basis per sub-face)
            self.bc["basis"] = z["bc_basis"]
        self.hf_eta = float(z["hf_eta"]) if "hf_eta" in z.files else None  # reconstruction_eta
        self.stiffness = z["stiffness"]
        self.ref = {}
        for k in MPSA_KEYS:
            shape = tuple(int(v) for v in z[f"ref_{k}_shape"])
            self.ref[k] = sps.csr_matrix((z[f"ref_{k}_data"], z[f"ref_{k}_indices"], z[f"ref_{k}_indptr"]), shape=shape)


class MpsaCase:
    """MPSA fixture made by oracle/gen_golden_mpsa.py from the reference."""

    def __init__(self, name: str):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.name = name
        self.grid = {k[5:]: z[k] for k in z.files if k.startswith("grid_")}
        self.grid["dim"] = int(self.grid["dim"])
        self.grid["name"] = str(self.grid["name"])
        self.bc = {"is_dir": z["bc_is_dir"], "is_neu": z["bc_is_neu"]}
        if "bc_is_rob" in z.files:
            self.bc["is_rob"] = z["bc_is_rob"]
            self.bc["robin_weight"] = z["bc_robin_weight"]
        if "bc_basis" in z.files:
            self.bc["basis"] = z["bc_basis"]
        self.stiffness = z["stiffness"]
        self.bc_values = z["bc_values"]
        self.source = z["source"]
        eta = float(z["eta"])
        self.eta = None if np.isnan(eta) else eta
        self.eta_sub = z["eta_sub"] if "eta_sub" in z.files else None  # continuity points per sub-face (sorted CSC order)
        self.hf_eta = float(z["hf_eta"]) if "hf_eta" in z.files else None  # reconstruction_eta
        if "hf_eta_sub" in z.files:  # ... one value per sub-face (sorted CSC order)
            self.hf_eta = z["hf_eta_sub"]
        self.ref = {}
        for k in MPSA_KEYS + ("A",):
            if f"ref_{k}_indptr" in z.files:
                shape = tuple(int(v) for v in z[f"ref_{k}_shape"])
                self.ref[k] = sps.csr_matrix(
                    (z[f"ref_{k}_data"], z[f"ref_{k}_indices"], z[f"ref_{k}_indptr"]), shape=shape)
        self.ref_rhs = z["ref_rhs"]
        self.ref_x = z["ref_x"]
        self.known_u = z["known_u"] if "known_u" in z.files else None
        self.known_stress = z["known_stress"] if "known_stress" in z.files else None
        self.known_rhs = z["known_rhs"] if "known_rhs" in z.files else None


class Case:
    def __init__(self, name: str):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.name = name
        self.grid = {k[5:]: z[k] for k in z.files if k.startswith("grid_")}
        self.grid["dim"] = int(self.grid["dim"])
        self.grid["name"] = str(self.grid["name"])
        self.bc = {k[3:]: z[k] for k in z.files if k.startswith("bc_") and k != "bc_values"}
        self.perm = z["perm"]
        self.bc_values = z["bc_values"]
        self.source = z["source"]
        eta = float(z["eta"])
        self.eta = None if np.isnan(eta) else eta
        self.eta_sub = z["eta_sub"] if "eta_sub" in z.files else None  # continuity points per sub-face (sorted CSC order)
        self.vector_source_values = z["vector_source_values"] if "vector_source_values" in z.files else None
        self.ref = {}
        for k in ALL_KEYS + ("A",):
            if f"ref_{k}_indptr" in z.files:
                shape = tuple(int(v) for v in z[f"ref_{k}_shape"])
                self.ref[k] = sps.csr_matrix(
                    (z[f"ref_{k}_data"], z[f"ref_{k}_indices"], z[f"ref_{k}_indptr"]), shape=shape
                )
        self.ref_rhs = z["ref_rhs"]
        self.ref_x = z["ref_x"]
        self.known_u = z["known_u"] if "known_u" in z.files else None
        self.known_flux = z["known_flux"] if "known_flux" in z.files else None


def rel_max_err(ours, ref) -> float:
    """max|ours - ref| / max|ref|  (SURVEY 8(d) parity metric (2))."""
    ours, ref = sps.csr_matrix(ours), sps.csr_matrix(ref)
    scale = abs(ref).max() if ref.nnz else 1.0
    diff = abs(ours - ref).max() if (ours.nnz + ref.nnz) else 0.0
    return float(diff / (scale if scale > 0 else 1.0))


def stored_pattern(m) -> set:
    c = sps.coo_matrix(m)
    return set(zip(c.row.tolist(), c.col.tolist()))


def check_pattern(ours, ref, tol=1e-12):
    """Pattern rule of SURVEY 8(d)(1): ref's stored pattern is a subset of ours and
    anything we store outside it is below tol * max|row of ref| (entries below the
    absolute roundoff floor 1e-15 * max|ref| are ignored: some reference rows are pure noise).  Returns (subset_ok, max_outside_ratio, equal)."""
    ours, ref = sps.csr_matrix(ours), sps.csr_matrix(ref)
    po, pr = stored_pattern(ours), stored_pattern(ref)
    subset = pr <= po
    rowmax = np.maximum(abs(ref).max(axis=1).toarray().ravel(), 0)
    gmax = abs(ref).max() if ref.nnz else 1.0
    worst = 0.0
    oc = ours.tocoo()
    for r, c, v in zip(oc.row, oc.col, oc.data):
        if (r, c) not in pr:
            if abs(v) <= 1e-15 * gmax:  # absolute roundoff floor (rows that are all noise)
                continue
            denom = rowmax[r] if rowmax[r] > 0 else gmax
            worst = max(worst, abs(v) / denom)
    return subset, worst, po == pr


class PartialCase:
    """Partial-discretization / update fixture made by oracle/gen_golden_partial.py."""

    def __init__(self, name: str):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.name = name
        self.grid = {k[5:]: z[k] for k in z.files if k.startswith("grid_")}
        self.grid["dim"] = int(self.grid["dim"])
        self.grid["name"] = str(self.grid["name"])
        self.bc = {k[3:]: z[k] for k in z.files if k.startswith("bc_")}
        self.perm, self.perm_new = z["perm"], z["perm_new"]
        self.modified_cells = z["modified_cells"]

        def mats(prefix):
            out = {}
            for k in ALL_KEYS:
                shape = tuple(int(v) for v in z[f"{prefix}_{k}_shape"])
                out[k] = sps.csr_matrix((z[f"{prefix}_{k}_data"], z[f"{prefix}_{k}_indices"],
                                         z[f"{prefix}_{k}_indptr"]), shape=shape)
            return out

        self.partial = []
        for i in range(int(z["num_partial"])):
            spec = {}
            for kind in ("cells", "faces", "nodes"):
                v = z[f"p{i}_spec_{kind}"]
                if not (v.size == 1 and v[0] == -1):
                    spec["specified_" + kind] = v
            self.partial.append({"spec": spec, "active_faces": z[f"p{i}_active_faces"],
                                 "active_cells": z[f"p{i}_active_cells"], "mats": mats(f"p{i}")})
        self.updated = mats("upd")


class TiltedCase:
    """2-D grid embedded in 3-D, ambient_dimension = 3 (oracle/gen_golden_tilted.py)."""

    def __init__(self, name: str):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.name = name
        self.grid = {k[5:]: z[k] for k in z.files if k.startswith("grid_")}
        self.grid["dim"] = int(self.grid["dim"])
        self.grid["name"] = str(self.grid["name"])
        self.bc = {k[3:]: z[k] for k in z.files if k.startswith("bc_") and k != "bc_values"}
        self.perm, self.bc_values = z["perm"], z["bc_values"]
        self.vector_source_values = z["vector_source_values"]
        self.ref = {}
        for k in ALL_KEYS + ("A",):
            shape = tuple(int(v) for v in z[f"ref_{k}_shape"])
            self.ref[k] = sps.csr_matrix((z[f"ref_{k}_data"], z[f"ref_{k}_indices"], z[f"ref_{k}_indptr"]),
                                         shape=shape)
        self.ref_rhs = z["ref_rhs"]
        self.vdim = int(z["vdim"]) if "vdim" in z.files else 3
        self.via_mpfa = bool(int(z["via_mpfa"])) if "via_mpfa" in z.files else False


class MpsaPartialCase:
    """Partial MPSA discretization / update fixture (oracle/gen_golden_mpsa_partial.py)."""

    def __init__(self, name: str):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.grid = {k[5:]: z[k] for k in z.files if k.startswith("grid_")}
        self.grid["dim"] = int(self.grid["dim"])
        self.grid["name"] = str(self.grid["name"])
        self.is_dir, self.is_neu = z["bc_is_dir"], z["bc_is_neu"]
        self.mu, self.lam, self.mu_new, self.lam_new = z["mu"], z["lam"], z["mu_new"], z["lam_new"]
        self.modified_cells = z["modified_cells"]

        def mats(prefix, keys):
            out = {}
            for k in keys:
                shape = tuple(int(v) for v in z[f"{prefix}_{k}_shape"])
                out[k] = sps.csr_matrix((z[f"{prefix}_{k}_data"], z[f"{prefix}_{k}_indices"],
                                         z[f"{prefix}_{k}_indptr"]), shape=shape)
            return out

        self.partial = []
        for i in range(int(z["num_partial"])):
            spec = {}
            for kind in ("cells", "faces", "nodes"):
                v = z[f"p{i}_spec_{kind}"]
                if not (v.size == 1 and v[0] == -1):
                    spec["specified_" + kind] = v
            self.partial.append({"spec": spec, "active_faces": z[f"p{i}_active_faces"], "mats": mats(f"p{i}", MPSA_KEYS)})
        self.updated = mats("upd", ("stress", "bound_stress"))


BIOT_KEYS = ("scalar_gradient", "displacement_divergence", "boundary_displacement_divergence", "mpsa_consistency",
             "bound_displacement_pressure")


class BiotCase:
    """Poro-elastic coupling fixture made by oracle/gen_golden_biot.py from the reference's pp.Biot."""

    def __init__(self, name: str):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.name = name
        self.grid = {k[5:]: z[k] for k in z.files if k.startswith("grid_")}
        self.grid["dim"] = int(self.grid["dim"])
        self.grid["name"] = str(self.grid["name"])
        self.bc = {"is_dir": z["bc_is_dir"], "is_neu": z["bc_is_neu"], "is_rob": z["bc_is_rob"],
                   "robin_weight": z["bc_robin_weight"]}
        self.stiffness = z["stiffness"]
        self.alphas = {str(k): z[f"alpha_{k}"] for k in z["alpha_keys"]}
        self.eta_sub = z["eta_sub"] if "eta_sub" in z.files else None  # continuity points per sub-face (sorted CSC order)

        def mat(prefix):
            shape = tuple(int(v) for v in z[prefix + "_shape"])
            return sps.csr_matrix((z[prefix + "_data"], z[prefix + "_indices"], z[prefix + "_indptr"]), shape=shape)

        self.ref = {k: {key: mat(f"ref_{k}__{key}") for key in self.alphas} for k in BIOT_KEYS}
        self.ref_mech = {k: mat("ref_" + k) for k in ("stress", "bound_stress")}


class SubfaceCase:
    """MPFA with boundary conditions per sub-face (oracle/gen_golden_subface.py)."""

    def __init__(self, name: str):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.name = name
        self.grid = {k[5:]: z[k] for k in z.files if k.startswith("grid_")}
        self.grid["dim"] = int(self.grid["dim"])
        self.grid["name"] = str(self.grid["name"])
        self.bc = {k[3:]: z[k] for k in z.files if k.startswith("bc_")}
        self.perm = z["perm"]
        self.ref = {}
        for k in ALL_KEYS:
            shape = tuple(int(v) for v in z[f"ref_{k}_shape"])
            self.ref[k] = sps.csr_matrix((z[f"ref_{k}_data"], z[f"ref_{k}_indices"], z[f"ref_{k}_indptr"]), shape=shape)


class PeriodicCase:
    """Grid with periodic faces; Mpfa and Tpfa matrices of the reference (oracle/gen_golden_periodic.py)."""

    def __init__(self, name: str):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.name = name
        self.grid = {k[5:]: z[k] for k in z.files if k.startswith("grid_")}
        self.grid["dim"] = int(self.grid["dim"])
        self.grid["name"] = str(self.grid["name"])
        self.bc = {k[3:]: z[k] for k in z.files if k.startswith("bc_") and k != "bc_values"}
        self.perm, self.bc_values = z["perm"], z["bc_values"]
        self.vector_source_values = z["vector_source_values"]
        self.periodic_face_map = z["periodic_face_map"]
        self.ref, self.tpfa = {}, {}
        for tag, out in (("ref_", self.ref), ("tpfa_", self.tpfa)):
            for k in ALL_KEYS + ("A",):
                shape = tuple(int(v) for v in z[f"{tag}{k}_shape"])
                out[k] = sps.csr_matrix((z[f"{tag}{k}_data"], z[f"{tag}{k}_indices"], z[f"{tag}{k}_indptr"]), shape=shape)
            out["rhs"] = z[tag + "rhs"]


def periodic_case_names():
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.startswith("periodic_") and f.endswith(".npz"))
