"""porepy_amd — MI355X-native MPFA-O assembly + sparse solve behind PorePy's
``Discretization.discretize()`` / ``assemble_matrix_rhs()`` operator API.

Only the hot path lives here: grids, parameters and the operator class are the
host-side mirror of the reference interface; the arithmetic is in ``csrc/`` (HIP, gfx950),
reached through the C ABI of ``include/porefv.h``.
"""
from . import _lib
from ._lib import Context, PorefvError
from .grid import (CartGrid, Grid, StructuredTetrahedralGrid, StructuredTriangleGrid, TetrahedralGrid, grid_from_raw,
                   grid_to_raw, perturb_interior_nodes)
from .mpfa import Mpfa, as_porepy_discretization, determine_eta
from .mpsa import Mpsa, as_porepy_mpsa
from .biot import Biot, as_porepy_biot
from .partial import active_indices
from .solvers import DeviceAssembly, HipLinearSolver, solve_block_system, solve_csr
from .device_csr import DeviceCsr, block_diag, bmat, merged_matrix, vstack
from . import ad
from . import md_sharding
from .tpfa import DifferentiableTpfa, Tpfa, as_porepy_ad_tpfa_flux
from .params import (DISCRETIZATION_MATRICES, PARAMETERS, BoundaryCondition, BoundaryConditionVectorial,
                     FourthOrderTensor, SecondOrderTensor, bc_flags, bc_to_raw, initialize_data)

__all__ = [
    "Context", "PorefvError", "Grid", "CartGrid", "StructuredTriangleGrid",
    "StructuredTetrahedralGrid", "TetrahedralGrid", "perturb_interior_nodes", "grid_to_raw", "grid_from_raw", "Mpfa",
    "as_porepy_discretization", "determine_eta", "SecondOrderTensor", "BoundaryCondition", "Mpsa",
    "FourthOrderTensor", "BoundaryConditionVectorial",
    "initialize_data", "bc_to_raw", "bc_flags", "PARAMETERS", "DISCRETIZATION_MATRICES", "_lib", "active_indices", "HipLinearSolver", "solve_csr", "Tpfa", "DifferentiableTpfa", "as_porepy_ad_tpfa_flux", "Biot", "as_porepy_mpsa", "as_porepy_biot", "DeviceCsr", "block_diag", "bmat", "merged_matrix", "vstack", "ad", "solve_block_system", "md_sharding", "DeviceAssembly",
]
