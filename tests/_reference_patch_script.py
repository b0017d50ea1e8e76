"""Run the REFERENCE's ``pp.Mpfa`` (or ``pp.Mpsa``: patches that carry a ``stiffness`` array) on grid patches handed
over as .npz files and save its matrices.

Executed in a subprocess (oracle.ref_env: the reference importable from the live tree or from the byte-compiled
archive oracle/_ref/porepy_ref.zip) by tests/_parity.grid_patch_parity: ``python _reference_patch_script.py DIR``
reads every ``DIR/patch_*.npz`` (raw grid arrays, permeability (3,3,Nc), condition flags, eta) and writes
``DIR/ref_*.npz``.  The patch grid is a ``pp.Grid`` over the same topology arrays with the SAME geometry arrays the
device saw (no second compute_geometry: identical inputs on both sides)."""
import glob
import os
import sys

import numpy as np
import scipy.sparse as sps

import porepy as pp

KEYS = ("flux", "bound_flux", "bound_pressure_cell", "bound_pressure_face", "vector_source",
        "bound_pressure_vector_source")


def grid_of(z):
    nd = int(z["dim"])
    nn = z["nodes"].shape[1]
    nf = z["face_centers"].shape[1]
    nc = z["cell_centers"].shape[1]
    fn = sps.csc_matrix((np.ones(z["fn_indices"].size, dtype=bool), z["fn_indices"], z["fn_indptr"]), shape=(nn, nf))
    cf = sps.csc_matrix((z["cf_sign"].astype(float), z["cf_indices"], z["cf_indptr"]), shape=(nf, nc))
    g = pp.Grid(nd, np.asarray(z["nodes"], dtype=float), fn, cf, str(z["name"]))
    g.face_normals = np.asarray(z["face_normals"], dtype=float)
    g.face_centers = np.asarray(z["face_centers"], dtype=float)
    g.cell_centers = np.asarray(z["cell_centers"], dtype=float)
    g.face_areas = np.asarray(z["face_areas"], dtype=float)
    g.cell_volumes = np.asarray(z["cell_volumes"], dtype=float)
    return g


MPSA_KEYS = ("stress", "bound_stress", "bound_displacement_cell", "bound_displacement_face")


def save(d, path, md, keys):
    out = {}
    for k in keys:
        m = sps.csr_matrix(md[k])
        m.sort_indices()
        out[k + "_data"], out[k + "_indices"], out[k + "_indptr"] = m.data, m.indices, m.indptr
        out[k + "_shape"] = np.array(m.shape)
    np.savez(os.path.join(d, "ref_" + os.path.basename(path)[6:]), **out)


def mpsa_patch(d, path, z, g):
    """numerics/fv/mpsa.py:121-529 on the patch: conditions per face and component ((nd, Nf) flags), stiffness (9, 9, Nc)."""
    bc = pp.BoundaryConditionVectorial(g)
    bc.is_dir = z["is_dir"].copy()
    bc.is_neu = z["is_neu"].copy()
    bc.is_rob = np.zeros_like(bc.is_dir)
    bc.is_internal = np.zeros(g.num_faces, bool)
    C = pp.FourthOrderTensor(np.ones(g.num_cells), np.ones(g.num_cells))
    C.values = np.asarray(z["stiffness"], dtype=float)
    data = pp.initialize_data({}, "mechanics", {"fourth_order_tensor": C, "bc": bc, "mpsa_eta": float(z["eta"]),
                                                  "inverter": "python"})
    pp.Mpsa("mechanics").discretize(g, data)
    save(d, path, data[pp.DISCRETIZATION_MATRICES]["mechanics"], MPSA_KEYS)


def main(d):
    n = 0
    for path in sorted(glob.glob(os.path.join(d, "patch_*.npz"))):
        z = np.load(path)
        g = grid_of(z)
        if "stiffness" in z.files:
            mpsa_patch(d, path, z, g)
            n += 1
            continue
        bc = pp.BoundaryCondition(g)
        bc.is_dir = z["is_dir"].copy()
        bc.is_neu = z["is_neu"].copy()
        bc.is_rob = np.zeros(g.num_faces, bool)
        bc.is_internal = np.zeros(g.num_faces, bool)
        K = pp.SecondOrderTensor(np.ones(g.num_cells))
        K.values = np.asarray(z["K"], dtype=float)
        data = pp.initialize_data({}, "flow", {"second_order_tensor": K, "bc": bc, "mpfa_eta": float(z["eta"]),
                                                  "mpfa_inverter": "python"})
        pp.Mpfa("flow").discretize(g, data)
        save(d, path, data[pp.DISCRETIZATION_MATRICES]["flow"], KEYS)
        n += 1
    print("RESULT", n, os.path.dirname(pp.__file__))


if __name__ == "__main__":
    main(sys.argv[1])
