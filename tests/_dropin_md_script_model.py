"""The reference's mixed-dimensional single-phase flow model used by the drop-in scripts (imports the REFERENCE)."""
import numpy as np

import porepy as pp
from porepy.applications.md_grids.domains import nd_cube_domain
from porepy.models.fluid_mass_balance import SinglePhaseFlow


class Geometry:
    def set_domain(self):
        self._domain = nd_cube_domain(3, 1.0)

    def set_fractures(self):
        f1 = pp.PlaneFracture(np.array([[0.5, 0.5, 0.5, 0.5], [0.0, 1.0, 1.0, 0.0], [0.0, 0.0, 1.0, 1.0]]))
        f2 = pp.PlaneFracture(np.array([[0.0, 1.0, 1.0, 0.0], [0.5, 0.5, 0.5, 0.5], [0.0, 0.0, 1.0, 1.0]]))
        self._fractures = [f1, f2]

    def grid_type(self):
        return "cartesian"

    def meshing_arguments(self):
        return {"cell_size": float(__import__("os").environ.get("MD_CELL_SIZE", "0.25"))}


class BCs:
    def bc_type_darcy_flux(self, sd):
        sides = self.domain_boundary_sides(sd)
        return pp.BoundaryCondition(sd, sides.west + sides.east, "dir")

    def bc_values_pressure(self, bg):
        sides = self.domain_boundary_sides(bg)
        v = np.zeros(bg.num_cells)
        v[sides.west] = 3.0
        v[sides.east] = 1.0
        return v


class Permeability:
    def permeability(self, subdomains):
        # fractures 100 x more permeable than the matrix (isotropic)
        vals = np.concatenate([np.full(sd.num_cells, 1.0 if sd.dim == 3 else 100.0) for sd in subdomains]) if subdomains else np.zeros(0)
        return self.isotropic_second_order_tensor(subdomains, pp.wrap_as_dense_ad_array(vals, name="k"))


class Model(Geometry, BCs, Permeability, SinglePhaseFlow):
    pass
