"""Parameter containers with the field names of the reference (params/tensor.py:68-157,
params/bc.py:68-190, params/data.py:116-204) — plain data holders, no compute."""
from __future__ import annotations

import numpy as np

PARAMETERS = "parameters"
DISCRETIZATION_MATRICES = "discretization_matrices"


class SecondOrderTensor:
    """Cell-wise symmetric permeability; ``values`` has shape (3, 3, Nc).

    Same defaults as the reference: kyy, kzz default to kxx; off-diagonals to zero.
    """

    def __init__(self, kxx, kyy=None, kzz=None, kxy=None, kxz=None, kyz=None):
        kxx = np.asarray(kxx, dtype=float)
        nc = kxx.size
        z = np.zeros(nc)
        kyy = kxx if kyy is None else np.asarray(kyy, dtype=float)
        kzz = kxx if kzz is None else np.asarray(kzz, dtype=float)
        kxy = z if kxy is None else np.asarray(kxy, dtype=float)
        kxz = z if kxz is None else np.asarray(kxz, dtype=float)
        kyz = z if kyz is None else np.asarray(kyz, dtype=float)
        if np.any(kxx < 0) or np.any(kxx * kyy - kxy * kxy < 0):
            raise ValueError("Tensor is not positive definite")
        v = np.zeros((3, 3, nc))
        v[0, 0], v[1, 1], v[2, 2] = kxx, kyy, kzz
        v[0, 1] = v[1, 0] = kxy
        v[0, 2] = v[2, 0] = kxz
        v[1, 2] = v[2, 1] = kyz
        self.values = v

    def copy(self):
        t = SecondOrderTensor.__new__(SecondOrderTensor)
        t.values = self.values.copy()
        return t

    def restrict_to_cells(self, cells):
        t = SecondOrderTensor.__new__(SecondOrderTensor)
        t.values = self.values[:, :, cells].copy()
        return t


class BoundaryCondition:
    """Scalar boundary condition per face: is_dir / is_neu / is_rob / is_internal,
    robin_weight.  Default: every boundary face Neumann; fracture faces are internal."""

    def __init__(self, sd, faces=None, cond=None):
        nf = sd.num_faces
        self.num_faces = nf
        self.dim = sd.dim - 1
        self.is_neu = np.zeros(nf, dtype=bool)
        self.is_dir = np.zeros(nf, dtype=bool)
        self.is_rob = np.zeros(nf, dtype=bool)
        self.is_internal = np.zeros(nf, dtype=bool)
        self.robin_weight = np.ones(nf)
        self.basis = np.ones(nf)
        bnd = sd.get_all_boundary_faces()
        self.is_neu[bnd] = True
        tags = getattr(sd, "tags", {})
        if "fracture_faces" in tags:
            self.is_internal = np.asarray(tags["fracture_faces"], dtype=bool).copy()
        if faces is not None:
            faces = np.asarray(faces)
            if faces.dtype == bool:
                if faces.size != nf:
                    raise ValueError("Wrong size of boolean face array")
                faces = np.flatnonzero(faces)
            if not np.all(np.isin(faces, bnd)):
                raise ValueError("Give boundary condition only on the boundary")
            if cond is None:
                raise ValueError("Boundary condition type must be given with the faces")
            if isinstance(cond, str):
                cond = [cond] * faces.size
            if faces.size != len(cond):
                raise ValueError("One BC per face")
            for f, c in zip(faces, cond):
                s = c.lower()
                if s == "neu":
                    pass
                elif s == "dir":
                    self.is_neu[f], self.is_dir[f] = False, True
                elif s == "rob":
                    self.is_neu[f], self.is_rob[f] = False, True
                else:
                    raise ValueError(f"Boundary should be Dirichlet, Neumann or Robin, not {c}")


def bc_to_raw(bc) -> dict:
    return {
        "is_dir": np.asarray(bc.is_dir, bool).copy(),
        "is_neu": np.asarray(bc.is_neu, bool).copy(),
        "is_rob": np.asarray(bc.is_rob, bool).copy(),
        "is_internal": np.asarray(getattr(bc, "is_internal", np.zeros_like(bc.is_dir)), bool).copy(),
        "robin_weight": np.asarray(bc.robin_weight, float).copy(),
    }


def bc_flags(bc) -> np.ndarray:
    """Pack a BoundaryCondition into the per-face flag byte of the C ABI."""
    r = bc_to_raw(bc)
    return (r["is_dir"] * 1 + r["is_neu"] * 2 + r["is_rob"] * 4 + r["is_internal"] * 8).astype(np.uint8)


def initialize_data(data: dict, keyword: str, specified_parameters: dict | None = None) -> dict:
    """data[PARAMETERS][keyword].update(specified_parameters); ensure the matrix dict exists."""
    data.setdefault(PARAMETERS, {}).setdefault(keyword, {}).update(specified_parameters or {})
    data.setdefault(DISCRETIZATION_MATRICES, {}).setdefault(keyword, {})
    return data


class FourthOrderTensor:
    """Cell-wise stiffness, ``values`` of shape (9, 9, Nc) with
    ``values[3*i + k, 3*a + l, c] = C_{ik,al}`` (sigma_ik = sum C_{ik,al} d_l u_a), isotropic from
    Lame parameters like the reference's constructor (params/tensor.py:251-349)."""

    def __init__(self, mu, lmbda, other_fields=None):
        mu = np.asarray(mu, dtype=float)
        lmbda = np.asarray(lmbda, dtype=float)
        if mu.ndim != 1 or lmbda.shape != mu.shape:
            raise ValueError("mu and lmbda must be 1-D arrays of equal length")
        nc = mu.size
        v = np.zeros((9, 9, nc))
        for i in range(3):
            for j in range(3):
                v[3 * i + i, 3 * j + j] += lmbda
                v[3 * i + j, 3 * i + j] += mu
                v[3 * i + j, 3 * j + i] += mu
        self.values = v
        self.mu = mu
        self.lmbda = lmbda

    def copy(self):
        t = FourthOrderTensor(self.mu.copy(), self.lmbda.copy())
        t.values = self.values.copy()
        return t


class BoundaryConditionVectorial:
    """Component-wise boundary condition: is_dir / is_neu / is_rob of shape (nd, Nf)
    (params/bc.py:222-322).  Default: every boundary face Neumann in every component."""

    def __init__(self, sd, faces=None, cond=None):
        nf, nd = sd.num_faces, sd.dim
        self.num_faces = nf
        self.dim = nd
        self.is_neu = np.zeros((nd, nf), dtype=bool)
        self.is_dir = np.zeros((nd, nf), dtype=bool)
        self.is_rob = np.zeros((nd, nf), dtype=bool)
        self.is_internal = np.zeros(nf, dtype=bool)
        self.robin_weight = np.tile(np.eye(nd)[:, :, None], (1, 1, nf))
        self.basis = np.tile(np.eye(nd)[:, :, None], (1, 1, nf))
        bnd = sd.get_all_boundary_faces()
        self.is_neu[:, bnd] = True
        if faces is not None:
            faces = np.asarray(faces)
            if faces.dtype == bool:
                faces = np.flatnonzero(faces)
            if cond is None:
                raise ValueError("Boundary condition type must be given with the faces")
            if isinstance(cond, str):
                cond = [cond] * faces.size
            for f, c in zip(faces, cond):
                s = c.lower()
                if s == "dir":
                    self.is_dir[:, f], self.is_neu[:, f] = True, False
                elif s == "rob":
                    self.is_rob[:, f], self.is_neu[:, f] = True, False
                elif s != "neu":
                    raise ValueError(f"Boundary should be Dirichlet, Neumann or Robin, not {c}")
