/* porefv.h — C ABI of libporefv_hip.so: MI355X-native MPFA-O assembly + sparse solve.
 *
 * This is the drop-in boundary for PorePy's finite-volume hot path.  PorePy is pure
 * Python, so the binding a maintainer adds is a ctypes stub; each
 * entry point names the reference code it stands in for (paths relative to
 * /root/reference/src/porepy).  Plain pointers and sizes only; no torch / numpy types.
 *
 * Conventions
 *   - every function returns a pfv_status (0 = ok); pfv_last_error() gives the text
 *   - all floating point is FP64, all indices int32 (CSR outputs as scipy stores them)
 *   - geometry arrays are SoA with 3 rows, row-major: a[3][N]   (Grid.nodes etc.)
 *   - "host" pointers are read/written synchronously; the handle owns all device memory
 *   - one handle = one HIP device + one stream; calls on a handle are blocking and
 *     must not be issued concurrently (the reference is single-threaded too)
 */
#ifndef POREFV_H
#define POREFV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pfv_ctx pfv_ctx;

typedef enum {
  PFV_OK = 0,
  PFV_ERR_SINGULAR = 1,      /* ValueError("Error in inversion of local linear systems"),
                                numerics/linalg/matrix_operations.py:1487-1490 */
  PFV_ERR_CELL_SHAPE = 2,    /* AssertionError: != nd faces of a cell meet in a node,
                                numerics/fv/_fvutils.py:735-736 */
  PFV_ERR_HIP = 3,           /* HIP runtime error; text in pfv_last_error */
  PFV_ERR_ARGUMENT = 4,      /* bad argument / call order */
  PFV_ERR_UNSUPPORTED = 5,   /* size or feature outside what the kernels cover */
  PFV_ERR_NOT_CONVERGED = 6  /* Krylov solve hit maxit (solution still returned) */
} pfv_status;

/* matrix selectors; 0..5 are the six keys FVElliptic stores
 * (numerics/fv/fv_elliptic.py:30-53), 6 is A = div @ flux (fv_elliptic.py:90-96) */
enum {
  PFV_MAT_FLUX = 0,
  PFV_MAT_BOUND_FLUX = 1,
  PFV_MAT_BOUND_PRESSURE_CELL = 2,
  PFV_MAT_BOUND_PRESSURE_FACE = 3,
  PFV_MAT_VECTOR_SOURCE = 4,
  PFV_MAT_BOUND_PRESSURE_VECTOR_SOURCE = 5,
  PFV_MAT_SYSTEM = 6,
  /* MPSA: the four keys Mpsa stores (numerics/fv/mpsa.py:82-95) and A = div_nd @ stress
   * (mpsa.py:515-529); vector unknowns are cell-major, component-minor (u[nd*c + a]) */
  PFV_MAT_STRESS = 7,
  PFV_MAT_BOUND_STRESS = 8,
  PFV_MAT_BOUND_DISPLACEMENT_CELL = 9,
  PFV_MAT_BOUND_DISPLACEMENT_FACE = 10,
  PFV_MAT_MECH_SYSTEM = 11,
  PFV_MAT_USER_SYSTEM = 12, /* matrix handed over by pfv_set_system */
  PFV_MAT_FLUX_JACOBIAN = 13, /* d flux / d p of pfv_mpfa_ad_flux_system, on the pattern of flux */
  PFV_NUM_MATS = 14
};

/* boundary-condition flag bits per face (params/bc.py:68-190: is_dir/is_neu/is_rob/
 * is_internal of BoundaryCondition) */
enum { PFV_BC_DIR = 1, PFV_BC_NEU = 2, PFV_BC_ROB = 4, PFV_BC_INTERNAL = 8 };

/* Krylov methods for pfv_solve (the reference only has direct solvers,
 * models/solution_strategy.py:830-884; results are judged against its solution) */
enum { PFV_SOLVE_CG = 0, PFV_SOLVE_BICGSTAB = 1, PFV_SOLVE_GMRES = 2 };

/* preconditioners of pfv_solve: Jacobi (default), or one V(1,1) cycle of a plain-aggregation
 * algebraic multigrid built on the device from the assembled matrix (pairwise matching on the
 * strength graph, piecewise-constant prolongation, Galerkin coarse matrices) */
enum { PFV_PRECOND_JACOBI = 0, PFV_PRECOND_AMG = 1, PFV_PRECOND_BLOCK = 2 };

/* flags for pfv_mpfa_discretize */
enum {
  PFV_DISCR_REBUILD_TOPOLOGY = 1, /* redo sub-cell topology + CSR symbolic phase even if
                                     cached (what every Mpfa.discretize call does) */
  PFV_DISCR_SKIP_VECTOR_SOURCE = 2 /* do not fill matrices 4 and 5 */
};

typedef struct {
  int32_t iterations;
  int32_t converged;
  double rel_residual; /* ||b - A x|| / ||b|| as tracked by the recurrence */
  double solve_ms;     /* device time of the solve (HIP events) */
} pfv_solve_info;

/* phase timings of the last calls, milliseconds, measured with HIP events on the
 * handle's stream (the reference logs wall-clock per phase:
 * models/solution_strategy.py:435-442,807-828,846-884) */
typedef struct {
  double topology_ms;     /* SubcellTopology equivalent          (_fvutils.py:51-172)   */
  double symbolic_ms;     /* CSR patterns of the 6 matrices + A                            */
  double node_ms;         /* interaction-region kernel            (mpfa.py:997-1045)     */
  double face_ms;         /* stencil scatter into CSR             (mpfa.py:1088-1147)    */
  double assemble_ms;     /* div @ flux, rhs                      (fv_elliptic.py:67-112) */
  double solve_ms;
  double bytes_written_outputs; /* 8*nnz (+4*nnz per distinct pattern) of what was filled */
  int64_t num_nodes, num_sub_half_faces, sum_block_sq, max_block;
  double amg_setup_ms;            /* last AMG hierarchy build */
  double amg_operator_complexity; /* sum of nnz over the levels / nnz of the system */
  int64_t amg_levels, amg_coarsest_rows;
  double discretize_ms;           /* whole pfv_mpfa_discretize call; less than the sum of its phases when the
                                     interaction-region kernel ran beside the symbolic phase on the handle's
                                     second stream (symbolic_ms and node_ms then are overlapping spans) */
  int64_t solve_renumbered;       /* 1: the last pfv_solve worked on the copy renumbered along the Morton curve,
                                     0: in place (grid already numbered that way, or a user system) */
  double node_flops;              /* FP64 operations the interaction-region kernel executes per launch on this grid:
                                     sum over the nodes of 2 n^3 (Gauss-Jordan) + 2 nd n nh (response table)
                                     + 2 nd nh (3 n_b) (boundary columns) + ~60 nd^2 per sub-cell (nK, D^-1, omega) */
  int64_t node_table_doubles;     /* doubles in the per-node response tables (what the node kernel writes and the
                                     face kernel reads) */
  int64_t amg_maps_reused;        /* 1: the last AMG setup kept the aggregates of the previous one (same pattern,
                                     new values: only the Galerkin products were redone), 0: full setup */
  int64_t amg_level0_nnz;         /* entries of the matrix the finest level of the cycle smooths with: nnz of the system,
                                     or of its strength-filtered copy (PFV_AMG_FILTER_PERMIL, scalar systems: default) */
  double amg_filter_theta;        /* threshold of that filter (0: off) */
  int64_t win_reused;             /* 1: the last discretize kept the SpMV windows of A -- the symbolic phase proved A's
                                     pattern equal (sizes + checksum of the index arrays) to the one they were built for */
  int64_t amg_filter_layout;      /* last AMG setup's strength filter: 0 counted and scanned the kept entries, 1 wrote into
                                     the row layout of the previous filtering of the same pattern (every row kept as many
                                     entries as its slot held), 2 tried that, found a row that did not, and ran again */
  int64_t solve_launches;         /* kernel dispatches of the last pfv_solve's Krylov loop, preconditioner applications
                                     included, its setup excluded (this library's own kernels; rocPRIM primitives, memsets
                                     and copies are not counted) */
  int64_t amg_setup_launches;     /* ... of the last preconditioner setup */
  int64_t node_redo;              /* interaction regions the first launch of the last discretization handed to the
                                     pivoted full body: those whose unpivoted elimination failed its a-posteriori check
                                     (PFV_NODE_GJ=5), or whose condition asked for refinement in a lean launch */
  int64_t symbolic_reused;        /* 1: the last discretize with PFV_DISCR_REBUILD_TOPOLOGY rebuilt the topology, proved it
                                     equal (64-bit digest of what the symbolic phase reads + sizes) to the one the CSR
                                     patterns on the handle were built from, and kept them; symbolic_ms then is the time of
                                     the proof.  0: the patterns were rebuilt (first call, new grid, PFV_SYMB_REUSE=0) */
  int64_t amg_stale_rematches;    /* times a solve on KEPT aggregate maps needed more than 1.3 x the iterations of the first
                                     solve after their matching (same tolerance and method): the maps were dropped and the
                                     next setup matched again */
  int64_t mpsa_contrast_regions;  /* MPSA: interaction regions of the last discretization whose sub-cells' stiffness scales
                                     differ by more than PFV_MPSA_CONTRAST_LIMIT (1e6) -- the regions that were assembled and
                                     eliminated in double-double arithmetic (mpsa_dd.inc) */
  double mpsa_max_contrast;       /* MPSA: the largest such ratio over all interaction regions (1: homogeneous) */
  int64_t assemble_positions_kept; /* 1: the last div @ flux replayed the positions of its entries recorded under the same
                                      kept patterns (no column indices read, no searches); 0: searched */
  int64_t pipeline_runs;          /* > 0: the last pfv_mpfa_discretize ran the interaction-region kernel in this many runs on the
                                     second stream with the face kernel following run by run on the first (node_ms is then the
                                     span of the whole pipeline, face_ms what came after it); 0: one after the other */
} pfv_stats;

pfv_status pfv_create(int device, pfv_ctx** out);
void pfv_destroy(pfv_ctx* h);
const char* pfv_last_error(pfv_ctx* h);
/* 1 if the library was built as the gfx950 HIP product, 0 for the host emulation build
 * that tests use to exercise the kernel logic without a GPU */
int pfv_is_device_build(void);

/* Grid arrays, the fields of pp.Grid the path reads (grids/grid.py:78-272):
 * cell_faces (CSC, Nf x Nc, data +-1, sorted indices), face_nodes (CSC, Nn x Nf),
 * nodes, face_normals, face_centers, cell_centers (3 x N), face_areas. */
pfv_status pfv_set_grid(pfv_ctx* h, int nd, int64_t nc, int64_t nf, int64_t nn,
                        const double* nodes, const int32_t* cf_indptr,
                        const int32_t* cf_indices, const int8_t* cf_sign,
                        const int32_t* fn_indptr, const int32_t* fn_indices,
                        const double* face_normals, const double* face_centers,
                        const double* cell_centers, const double* face_areas);

/* Discretization parameters (numerics/fv/mpfa.py:119-167): permeability as
 * SecondOrderTensor.values, shape (3,3,Nc) C-order; bc flags; Robin weight per face
 * (may be NULL = 1); eta scalar, or per-subface array (length = nnz(face_nodes), in
 * face_nodes CSC order) when eta_subface != NULL. */
pfv_status pfv_mpfa_set_params(pfv_ctx* h, const double* perm_33n, const uint8_t* bc_flags,
                               const double* robin_weight, double eta,
                               const double* eta_subface);

/* New permeability values only (same shape and order as above), everything else of pfv_mpfa_set_params kept: what
 * the step of a nonlinear / time-dependent model changes between two Mpfa.discretize calls (the reference reads
 * data[PARAMETERS][kw]["second_order_tensor"] afresh in every call, numerics/fv/mpfa.py:121-122).  perm_33n is host
 * memory, or device memory after pfv_set_vectors_on_device(h, 1) (copied device-to-device: the coefficient field of
 * a model that evaluates K on the GPU never crosses PCIe). */
pfv_status pfv_mpfa_set_permeability(pfv_ctx* h, const double* perm_33n);

/* Residual and Jacobian of the flow equation with a pressure-dependent permeability, on the device
 * (csrc/ad_flux.inc; the reference: AdTpfaFlux.diffusive_flux with an Mpfa base discretization,
 * models/constitutive_laws.py:1195-1336, 1580-1721): q = T_MPFA p + t_bnd bc + VS_MPFA g with the
 * matrices of the last pfv_mpfa_discretize (computed for K = K(p)), and the product rule
 * dq = T_MPFA dp + diag(w) dT_TPFA with the two-point derivative of the transmissibility.
 *   p         cell pressures (Nc)
 *   dk_dp     d K_rs(c) / d p_c, layout (3,3,Nc) like the permeability (NULL: K does not depend on p)
 *   bc_values per face (Dirichlet value / Neumann flux); vector_source nd per cell or NULL; source Nc or NULL
 *   flux_out  Nf values of q, or NULL
 * Leaves J = d(div q)/dp (pattern of PFV_MAT_SYSTEM) and -(div q - source) as the active system: pfv_solve
 * then returns the Newton increment; pfv_get_matrix(PFV_MAT_SYSTEM) / pfv_get_rhs copy them out.  With
 * PFV_AD_WANT_FLUX_JACOBIAN also dq/dp as PFV_MAT_FLUX_JACOBIAN (pattern of flux).  Vectors are host or
 * device memory as selected by pfv_set_vectors_on_device. */
#define PFV_AD_WANT_FLUX_JACOBIAN 1u
pfv_status pfv_mpfa_ad_flux_system(pfv_ctx* h, const double* p, const double* dk_dp, const double* bc_values,
                                   const double* vector_source, const double* source, double* flux_out,
                                   uint32_t flags);

/* Boundary conditions given per SUB-FACE (numerics/fv/mpfa.py:761-768): flags and Robin weights with
 * one entry per (face, node) pair in face_nodes CSC order (sorted indices), replacing the per-face
 * arrays of pfv_mpfa_set_params.  The following pfv_mpfa_discretize then keeps sub-face rows (and
 * sub-face columns of the boundary matrices) in matrices 0-3 -- no collapse to faces, traces not
 * averaged, Neumann data not divided by the number of face nodes (mpfa.py:1117-1125, 1516-1523) --
 * while matrices 4-5 keep face rows.  flags = NULL returns to per-face conditions. */
pfv_status pfv_mpfa_set_subface_bc(pfv_ctx* h, const uint8_t* bc_flags_sub, const double* robin_weight_sub);

/* Mpfa._flux_discretization (numerics/fv/mpfa.py:592-1156) on the device. */
pfv_status pfv_mpfa_discretize(pfv_ctx* h, uint32_t flags);

/* Two-point flux approximation, Tpfa.discretize (numerics/fv/tpfa.py:84-279): what the
 * reference's Mpfa delegates 1-D grids to (mpfa.py:690-712).  Uses the grid and the parameters
 * of pfv_mpfa_set_params (eta and Robin weights are not used); works for nd = 1, 2, 3.  Fills
 * matrices 0-5 with the patterns the reference stores (cell_faces pattern, diagonals);
 * vector_source_dim = ``ambient_dimension`` (number of vector-source components per cell). */
pfv_status pfv_tpfa_discretize(pfv_ctx* h, int vector_source_dim);

/* Differentiable two-point transmissibilities: the numerical core of
 * AdTpfaFlux.__transmissibility_matrix (models/constitutive_laws.py:1504-1578) on the half-face
 * geometry of DifferentiableTpfa (numerics/fv/tpfa.py:546-620), value and derivative in one pass
 * instead of two sparse products through the forward-AD machinery.  For every half-face
 * e = (cell c, face f, sign), in the order of the cell_faces entries given to pfv_set_grid (cell by
 * cell; the reference numbers its half-faces face by face, sps.find(sd.cell_faces), which only matters
 * for vectors indexed by half-face - the Jacobian below is the same matrix either way):
 *     t_e = d^T K_c n_f / |d|^2,  d = x_f - x_c        t_f = 1 / sum_{e of f} sgn_e / t_e
 *     t_face[f] = t_f                                   (Nf values)
 *     dt_dk[9 e + 3 r + s] = d t_f / d K_c[r][s] = t_f^2 sgn_e d_r n_s / (t_e^2 |d|^2)   (9 nnz(cell_faces))
 * i.e. row f(e), column 9 c(e) + 3 r + s of the Jacobian of t_f_full with respect to the reference's
 * k_c vector.  perm_33n: (3,3,Nc) C-order as in pfv_mpfa_set_params; needs only the grid.  The three
 * arrays follow pfv_set_vectors_on_device (host by default). */
pfv_status pfv_tpfa_transmissibility_ad(pfv_ctx* h, const double* perm_33n, double* t_face, double* dt_dk);

/* Partial (re)discretization: the node-list launch behind ``specified_cells / _faces /
 * _nodes`` (numerics/fv/mpfa.py:178-204, 466-508) and ``update_discretization``
 * (mpfa.py:510-590, _fvutils.py:1090-1257).  ``faces`` = the faces whose rows are to be
 * computed (the reference's ``active_faces``, _fvutils.py:1260-1462, computed by the host);
 * the interaction regions of their nodes are re-solved with the current parameters and
 * exactly these rows of the six matrices are rewritten.  keep_other_rows = 0: every other
 * row is zeroed (``specified_*`` semantics); 1: other rows keep the previous discretization
 * (update semantics; requires an earlier discretize on this handle). */
pfv_status pfv_mpfa_discretize_faces(pfv_ctx* h, uint32_t flags, int64_t n_faces,
                                     const int32_t* faces, int keep_other_rows);

/* shape and nnz of a produced matrix */
pfv_status pfv_matrix_info(pfv_ctx* h, int which, int64_t* nrows, int64_t* ncols,
                           int64_t* nnz);
/* copy a matrix to caller-allocated CSR arrays (indptr nrows+1, indices nnz, data nnz);
 * any pointer may be NULL to skip that array */
pfv_status pfv_get_matrix(pfv_ctx* h, int which, int32_t* indptr, int32_t* indices,
                          double* data);
/* the same for a list of rows only (gathered on the device, one copy): what a caller that slices
 * `data[DISCRETIZATION_MATRICES][kw]["flux"][faces]` needs instead of the whole matrix (21.6 GB for the
 * six matrices of a 2 M-cell grid).  out_indptr has n_rows + 1 entries (row i of the output = row rows[i]);
 * call once with out_indices = out_data = NULL to learn the sizes, then again with the arrays. */
/* free / total bytes of the handle's device (free includes the blocks parked in the handle's own cache);
 * -1 / -1 from the host-emulation build.  The host mirror checks its footprint estimate against it before
 * the first discretization of a grid (the reference's peak-memory estimate: mpfa.py:1315-1355). */
pfv_status pfv_device_memory(pfv_ctx* h, int64_t* free_bytes, int64_t* total_bytes);
/* Page-locked host memory for the arrays the copy-out calls fill (pfv_get_matrix, pfv_get_rhs, pfv_solve): into such a
 * buffer the device writes over PCIe directly (~50 GB/s), into pageable memory the runtime stages the copy (23 GB/s
 * measured) and the freshly allocated pages fault in first (the "eager" operator path moved 23 GB at 6 GB/s that
 * way).  The host mirror keeps the blocks in a pool and hands them to numpy (porepy_amd/_lib.py: PinnedPool), so that
 * the matrices of the next time step land in the blocks the previous ones gave back.  Host emulation: malloc / free.
 * (The reference's matrices are ordinary scipy arrays: fv_elliptic.py:67-112 -- so are these, only their memory
 * comes from here.) */
pfv_status pfv_host_alloc(size_t bytes, void** out);
void pfv_host_free(void* p);
/* number of unknowns of the system pfv_solve / pfv_get_rhs operate on (0: nothing assembled): the length of
 * the arrays those calls write */
pfv_status pfv_active_size(pfv_ctx* h, int64_t* n);
pfv_status pfv_get_matrix_rows(pfv_ctx* h, int which, int64_t n_rows, const int32_t* rows, int32_t* out_indptr,
                               int32_t* out_indices, double* out_data);

/* FVElliptic.assemble_matrix_rhs (numerics/fv/fv_elliptic.py:67-112):
 * A = div @ flux, b = -div @ bound_flux @ bc_values - div @ vector_source @ g + source.
 * vector_source (Nc*nd) and source (Nc) may be NULL.  Results stay on the device. */
pfv_status pfv_mpfa_assemble(pfv_ctx* h, const double* bc_values, const double* vector_source,
                             const double* source);
pfv_status pfv_get_rhs(pfv_ctx* h, double* b);

/* y = M x for a produced matrix; host vectors (testing / flux post-processing) */
pfv_status pfv_spmv(pfv_ctx* h, int which, const double* x, double* y);

/* ---- Biot: coupling terms of the poro-elastic discretization (numerics/fv/biot.py:247-1135),
 * computed from the same interaction-region inverse as MPSA.  One set of five matrices per
 * coupling tensor ("scalar_vector_mappings" of the reference: Biot's alpha, thermal expansion...). */
enum {
  PFV_BIOT_SCALAR_GRADIENT = 0,                 /* (nd Nf x Nc)  */
  PFV_BIOT_DISPLACEMENT_DIVERGENCE = 1,         /* (Nc x nd Nc)  */
  PFV_BIOT_BOUNDARY_DISPLACEMENT_DIVERGENCE = 2, /* (Nc x nd Nf)  */
  PFV_BIOT_CONSISTENCY = 3,                     /* (Nc x Nc), "mpsa_consistency" */
  PFV_BIOT_BOUND_DISPLACEMENT_PRESSURE = 4,     /* (nd Nf x Nc)  */
  PFV_BIOT_NUM_TERMS = 5
};
/* coupling tensors as SecondOrderTensor.values, shape (nalpha, 3, 3, Nc) C-order; nalpha = 0
 * switches the coupling terms off.  Needs the grid and pfv_mpsa_set_params. */
pfv_status pfv_biot_set_alphas(pfv_ctx* h, int nalpha, const double* alpha_k33n);
/* Partial discretization / update of the coupling terms (biot.py:151-245 update_discretization,
 * :326-345 specified_cells / faces / nodes): the interaction regions of the nodes of `faces` are
 * recomputed; face rows (stress, bound_stress, traces, scalar_gradient, bound_displacement_pressure)
 * of `faces` and cell rows (displacement_divergence, boundary_displacement_divergence, consistency) of
 * `cells` are rewritten - the caller passes cells all of whose nodes are among the nodes of `faces`,
 * so that their rows are complete.  keep_other_rows = 0: every other row becomes zero (partial
 * discretization); 1: they keep their values (update). */
pfv_status pfv_biot_discretize_faces(pfv_ctx* h, uint32_t flags, int64_t n_faces, const int32_t* faces,
                                     int64_t n_cells, const int32_t* cells, int keep_other_rows);

/* Biot._local_discretization (biot.py:714-878): pfv_mpsa_discretize plus the coupling terms */
pfv_status pfv_biot_discretize(pfv_ctx* h, uint32_t flags);
pfv_status pfv_biot_matrix_info(pfv_ctx* h, int term, int64_t* nrows, int64_t* ncols, int64_t* nnz);
pfv_status pfv_biot_get_matrix(pfv_ctx* h, int term, int key, int32_t* indptr, int32_t* indices,
                               double* data);

/* Hand an assembled system to the device solver: the caller of the hot path one level up,
 * SolutionStrategy.solve_linear_system (models/solution_strategy.py:830-884), holds the global
 * Jacobian as a scipy CSR matrix and the residual as a numpy vector.  CSR arrays (int32, any
 * column order inside a row) and rhs are copied to the device; every row needs a non-zero
 * diagonal entry (Jacobi preconditioner) or PFV_ERR_UNSUPPORTED is returned.  pfv_solve then
 * operates on this system; needs no grid. */
pfv_status pfv_set_system(pfv_ctx* h, int64_t n, const int32_t* indptr, const int32_t* indices,
                          const double* data, const double* rhs);

/* Periodic faces (Grid.set_periodic_map, grids/grid.py:879-911; SubcellTopology merges the right
 * sub-faces and nodes into the left ones, numerics/fv/_fvutils.py:91-137; Tpfa pairs the cells,
 * numerics/fv/tpfa.py:114-262).  The host passes the *merged* grid to pfv_set_grid (the right cell lists
 * the left face, right nodes renamed to left nodes, right faces left without cells and nodes;
 * porepy_amd/periodic.py) and tells here which side of a merged face is displaced: for every face f,
 * native_cell[f] = the cell that sees the face where it is (-1: not a periodic face) and
 * shift[3][Nf] (SoA) = x_f(left) - x_f(right); every other cell of f sees the face centre at
 * face_centers[f] - shift[f].  Call after pfv_set_grid (which clears it); NULL, NULL clears. */
pfv_status pfv_set_periodic(pfv_ctx* h, const int32_t* native_cell, const double* shift);

/* Where the vector arguments of pfv_mpfa_assemble (bc_values, vector_source, source),
 * pfv_mpsa_assemble (bc_values, source) and pfv_solve (x0, x) live: 0 (default) host memory, as the
 * reference's numpy arrays (fv_elliptic.py:67-112); 1 device memory of this handle's GPU - models
 * that keep iterating on the device skip PCIe altogether.  Device buffers must be complete on the
 * handle's stream (see pfv_set_stream) when the call is made; pfv_solve returns after x is written. */
pfv_status pfv_set_vectors_on_device(pfv_ctx* h, int on);

/* Select the preconditioner of the following pfv_solve calls on this handle. */
pfv_status pfv_set_preconditioner(pfv_ctx* h, int kind);

/* Block preconditioner (PFV_PRECOND_BLOCK) for the coupled Jacobians of mixed-dimensional / multi-physics models
 * that the reference solves directly (models/solution_strategy.py:830-884; mortar coupling
 * models/constitutive_laws.py:987-1000): the unknowns of the system given to pfv_set_system are grouped in
 * n_blocks contiguous blocks [block_ptr[k], block_ptr[k+1]) -- one per (variable, subdomain or interface) -- and
 * the preconditioner is the block lower-triangular part of A, every diagonal block solved by one cycle of its own
 * aggregation-AMG hierarchy (blocks of at most 1024 rows: exactly, by their dense inverse).  gauss_seidel = 0: block
 * Jacobi.  Every diagonal entry must be non-zero: systems whose equations are ordered differently from their
 * unknowns are row-permuted first (host side: porepy_amd.solvers.match_rows).  Selects PFV_PRECOND_BLOCK; the
 * blocks are (re)built by the next pfv_solve. */
pfv_status pfv_set_block_preconditioner(pfv_ctx* h, int64_t n_blocks, const int64_t* block_ptr, int gauss_seidel);

/* Jacobi-preconditioned Krylov solve of A x = b on the device (stand-in for
 * SolutionStrategy.solve_linear_system, models/solution_strategy.py:830-884).
 * x0 may be NULL (zero start); x receives Nc values. */
pfv_status pfv_solve(pfv_ctx* h, int method, double rtol, int maxit, int restart,
                     const double* x0, double* x, pfv_solve_info* info);

/* ---- MPSA-W (numerics/fv/mpsa.py): same grid, vector unknowns ------------------------------
 * stiffness: FourthOrderTensor.values, shape (9,9,Nc) C-order (params/tensor.py:300-349);
 * bc_dir_bits / bc_neu_bits: per face, bit a set if component a is Dirichlet / Neumann
 * (BoundaryConditionVectorial.is_dir / is_neu, params/bc.py:222-322; Cartesian basis, no Robin);
 * cell_volumes: Grid.cell_volumes (Nc), the weights of the node average (mpsa.py:1619-1640);
 * eta as in pfv_mpfa_set_params (scalar). */
pfv_status pfv_mpsa_set_params(pfv_ctx* h, const double* stiffness_99n, const double* cell_volumes,
                               const uint8_t* bc_dir_bits, const uint8_t* bc_neu_bits, double eta);
/* Continuity points per sub-face for MPSA (`mpsa_eta` given as an array of SubcellTopology.num_subfno_unique values,
 * numerics/fv/mpsa.py:293-303, 647-652; _fvutils.py:222-277): eta_subface[s] for sub-face s = position of the (face,
 * node) pair in the face_nodes CSC arrays, used as given also on the boundary.  Call after pfv_mpsa_set_params (which
 * clears it); NULL removes it. */
pfv_status pfv_mpsa_set_subface_eta(pfv_ctx* h, const double* eta_subface);
/* `reconstruction_eta` (numerics/fv/mpsa.py:185, 757-761, _reconstruct_displacement :1187-1266): the displacement
 * traces (bound_displacement_cell / bound_displacement_face) are reconstructed at x_f + hf_eta (x_v - x_f) from the
 * sub-cell gradients, averaged over the two sides of the sub-face, instead of at the continuity points (a scalar: 0
 * on boundary faces, as the reference's distance routine does).  on = 0 switches back.  After pfv_mpsa_set_params
 * (which clears it).  Not combined with the Biot coupling terms (PFV_ERR_UNSUPPORTED). */
pfv_status pfv_mpsa_set_reconstruction_eta(pfv_ctx* h, int on, double hf_eta);
/* the same with one value per sub-face (compute_dist_face_cell with an array, numerics/fv/_fvutils.py:222-277: used as
 * given, also on the boundary), sub-faces in the order of the sorted face_nodes CSC arrays; NULL switches the
 * reconstruction points back to the continuity points. */
pfv_status pfv_mpsa_set_reconstruction_eta_subface(pfv_ctx* h, const double* hf_eta_subface);
/* Robin conditions of the vectorial boundary condition (BoundaryConditionVectorial.is_rob,
 * .robin_weight, params/bc.py:222-322; rows of numerics/fv/mpsa.py:1381-1459): bit c of
 * bc_rob_bits[f] = component c of face f is Robin; robin_weight_ddn = weights W[i][a][f], shape
 * (nd, nd, Nf) C-order (NULL = identity).  Call after pfv_mpsa_set_params (which clears them);
 * bc_rob_bits = NULL removes them. */
pfv_status pfv_mpsa_set_robin(pfv_ctx* h, const uint8_t* bc_rob_bits, const double* robin_weight_ddn);
/* face-wise basis in which the boundary conditions are given (BoundaryConditionVectorial.basis,
 * shape (nd, nd, Nf) C-order; row k = the k-th direction); NULL = Cartesian.  Call after
 * pfv_mpsa_set_params (which resets it). */
pfv_status pfv_mpsa_set_basis(pfv_ctx* h, const double* basis_ddn);

/* Conditions per SUB-FACE (numerics/fv/mpsa.py:712-720: a BoundaryConditionVectorial with one entry per sub-face;
 * :752-754, 780-781 the outputs; :1127-1138 Neumann / Robin data integrated over the sub-face, not divided by
 * #nodes).  Arrays with one entry per sub-face in the order of the SORTED face_nodes CSC arrays: component bit
 * masks as in pfv_mpsa_set_params, optional Robin flags and weights (nd, nd, Nsf) C-order (NULL = identity).
 * After this call pfv_mpsa_discretize fills
 *   PFV_MAT_STRESS                   (nd Nsf x nd Nc)    one row block per sub-face
 *   PFV_MAT_BOUND_STRESS             (nd Nsf x nd Nsf)
 *   PFV_MAT_BOUND_DISPLACEMENT_CELL  (nd Nf  x nd Nc)    as with conditions per face
 *   PFV_MAT_BOUND_DISPLACEMENT_FACE  (nd Nf  x nd Nsf)
 * and pfv_mpsa_assemble refuses (the caller collapses the sub-face rows first, as with the reference).
 * bc_dir_bits_sub = NULL returns to conditions per face; pfv_mpsa_set_params resets it.  Not combined with
 * Biot coupling terms or partial updates; a basis comes per sub-face too (pfv_mpsa_set_subface_basis). */
pfv_status pfv_mpsa_set_subface_bc(pfv_ctx* h, const uint8_t* bc_dir_bits_sub, const uint8_t* bc_neu_bits_sub,
                                   const uint8_t* bc_rob_bits_sub, const double* robin_weight_dds);
/* the basis of conditions per sub-face (the .basis of a BoundaryConditionVectorial with one entry per sub-face,
 * numerics/fv/_fvutils.py:836-852 through ExcludeBoundaries.basis_matrix): shape (nd, nd, Nsf) C-order, sub-faces in
 * the order of the sorted face_nodes CSC arrays; NULL = Cartesian.  Call after pfv_mpsa_set_subface_bc (which
 * resets it). */
pfv_status pfv_mpsa_set_subface_basis(pfv_ctx* h, const double* basis_dds);

/* Mpsa._stress_discretization (numerics/fv/mpsa.py:531-782) on the device; fills matrices 7-10 */
pfv_status pfv_mpsa_discretize(pfv_ctx* h, uint32_t flags);

/* Partial MPSA (re)discretization, same contract as pfv_mpfa_discretize_faces
 * (numerics/fv/mpsa.py:196-216, 383-416 and Mpsa.update_discretization :418-487): the
 * interaction regions around the listed faces are re-solved, the nd rows of each listed face
 * are rewritten in the four matrices, the other rows are zeroed (keep_other_rows = 0) or kept. */
pfv_status pfv_mpsa_discretize_faces(pfv_ctx* h, uint32_t flags, int64_t n_faces,
                                     const int32_t* faces, int keep_other_rows);
/* Mpsa.assemble_matrix_rhs (mpsa.py:486-529): A = div_nd @ stress,
 * b = -div_nd @ bound_stress @ bc_values + source; bc_values has nd*Nf entries ((nd,Nf) raveled
 * column-major), source nd*Nc or NULL.  Makes the mechanics system the one pfv_solve works on
 * (pfv_mpfa_assemble switches back to the flow system). */
pfv_status pfv_mpsa_assemble(pfv_ctx* h, const double* bc_values, const double* source);

/* Device-pointer variants for multi-GPU drivers that keep vectors in HBM
 * (torch tensors): y = A x on the handle's stream; d_x has num_cols entries. */
pfv_status pfv_spmv_device(pfv_ctx* h, int which, const double* d_x, double* d_y);
/* same, but only rows [0, nrows): a sharded solve multiplies the rows of the cells a rank owns
 * (owned cells are numbered first) against owned + halo entries of x */
pfv_status pfv_spmv_device_rows(pfv_ctx* h, int which, int64_t nrows, const double* d_x, double* d_y);
pfv_status pfv_get_device_rhs(pfv_ctx* h, double** d_b, double** d_diag);
/* device-to-device copy of an internal vector into caller memory (e.g. a torch tensor):
 * which = 0 right-hand side, 1 diagonal of the system assembled last (Nc entries for flow,
 * nd Nc for mechanics) */
pfv_status pfv_copy_device_vector(pfv_ctx* h, int which, double* d_dst, int64_t count);
/* Block preconditioner of a sharded solve: aggregation-AMG hierarchy of the leading
 * n_own x n_own block of the active system (a rank's owned cells; couplings to halo columns are
 * dropped), then one V-cycle per call on device vectors of length n_own -- no communication
 * inside the preconditioner.  n_own = 0 means the whole active system. */
pfv_status pfv_amg_setup(pfv_ctx* h, int64_t n_own);
pfv_status pfv_amg_apply_device(pfv_ctx* h, const double* d_r, double* d_z);

/* Sharded Krylov solve: the fused BiCGStab / CG loop of pfv_solve on the rows of the cells this rank
 * owns (the leading n_own rows of the active system; owned cells are numbered first, the remaining
 * columns are halo cells owned by other ranks), with the two data exchanges of the distributed
 * method handed to the caller.  The reference has no distributed path; the split follows its
 * sub-problem machinery (numerics/fv/_fvutils.py:414-539), the solve stands in for
 * SolutionStrategy.solve_linear_system (models/solution_strategy.py:830-884).
 *   exchange_halo(user, d_x, stream): entries [n_own, n_local) of the SpMV input d_x are to be
 *     filled from their owners (RCCL ncclSend / ncclRecv over xGMI); the owned entries are current;
 *   allreduce_sum(user, d_vals, count, stream): in-place sum over the ranks of count (<= 8) doubles
 *     in device memory (ncclAllReduce) -- one call per fused group of dot products (BiCGStab: two calls per
 *     iteration, of 2 and 5 sums: the sums of the vector update ride with those of omega).
 * Both run on the calling thread between kernel launches and must only enqueue work ordered with
 * `stream` (the handle's hipStream_t as void*, see pfv_set_stream); a non-zero return aborts the
 * solve with PFV_ERR_ARGUMENT.  d_work: 2 * n_local + 8 doubles of caller device memory (the two
 * SpMV inputs d_x the hooks see are d_work and d_work + n_local, d_vals is d_work + 2 * n_local).
 * Preconditioner as selected by pfv_set_preconditioner: Jacobi, or one cycle of the block hierarchy
 * of pfv_amg_setup(n_own) (block Jacobi across ranks, no communication inside).  Every rank takes
 * the same branches: the convergence test reads the reduced residual.  d_x_owned receives n_own
 * values (device memory, zero start). */
typedef struct {
  int (*exchange_halo)(void* user, double* d_x, void* stream);
  int (*allreduce_sum)(void* user, double* d_vals, int count, void* stream);
  void* user;
  /* Transport of the coupled hierarchy (pfv_amg_setup_sharded; may be NULL for every other call).  The library
   * owns the halo plans of all levels and packs / unpacks itself; the transport only moves packed buffers:
   *   sendrecv(user, n_peers, peers, d_send, send_ptr, d_recv, recv_ptr, stream): to peer p go the doubles
   *     d_send[send_ptr[p] .. send_ptr[p+1]), from it arrive d_recv[recv_ptr[p] .. recv_ptr[p+1])
   *     (ncclGroupStart / ncclSend + ncclRecv per peer / ncclGroupEnd); the offset arrays are host memory, valid
   *     during the call only;
   *   allgather(user, d_send, d_recv, bytes_per_rank, stream): rank r's bytes_per_rank bytes land at
   *     d_recv + r * bytes_per_rank on every rank (ncclAllGather). */
  int (*sendrecv)(void* user, int n_peers, const int32_t* peers, const double* d_send, const int64_t* send_ptr,
                  double* d_recv, const int64_t* recv_ptr, void* stream);
  int (*allgather)(void* user, const void* d_send, void* d_recv, int64_t bytes_per_rank, void* stream);
} pfv_shard_hooks;
pfv_status pfv_solve_sharded(pfv_ctx* h, int method, double rtol, int maxit, int64_t n_own,
                             const pfv_shard_hooks* hooks, double* d_work, double* d_x_owned,
                             pfv_solve_info* info);

/* Coupled hierarchy of a sharded solve (instead of pfv_amg_setup(n_own), whose block hierarchy ignores the
 * couplings between ranks and pays for it in iterations): every level keeps the columns of the unknowns other ranks
 * own (aggregates never cross a rank boundary; the Galerkin product of the owned rows carries the halo columns
 * along, renamed to the owner's aggregates), smoothing and residual products see current halo values (one
 * point-to-point exchange before each), and from the first level with at most PFV_AMG_GATHER_ROWS (default 32768)
 * rows in total the rows of all ranks are gathered once (allgather) and every rank continues with the same
 * replicated hierarchy -- per cycle one allgather of that level's right-hand side, no further communication.
 * The plan of the finest level is given in CELLS (bs unknowns each travel): send_idx[send_ptr[p] ..) are the owned
 * cells peer p needs, in the order it expects them; recv_pos[recv_ptr[p] ..) the local cells (>= n_own / bs) its
 * values land in, in the order it sends them.  Every rank calls this (it is a collective of the transport);
 * the hooks (all four entries) must stay valid until the next setup or the handle's end.  The reference has no
 * distributed path (models/solution_strategy.py:830-884 solves on one process). */
pfv_status pfv_amg_setup_sharded(pfv_ctx* h, int64_t n_own, const pfv_shard_hooks* hooks, int rank, int world,
                                 int n_peers, const int32_t* peers, const int64_t* send_ptr,
                                 const int32_t* send_idx, const int64_t* recv_ptr, const int32_t* recv_pos);

/* The two hooks served natively over RCCL (xGMI), no Python in the iteration (csrc/rccl_hooks.inc):
 * pack kernel -> ncclGroupStart / ncclSend + ncclRecv per neighbour / ncclGroupEnd -> unpack kernel, and
 * ncclAllReduce of the fused pair of sums, all enqueued on the handle's stream.  librccl is bound with
 * dlopen at the first call.  Rank 0 creates the id, the caller distributes its 128 bytes (torch.distributed,
 * MPI, a file), every rank creates its communicator on its handle's device and describes its halo plan;
 * pfv_rccl_hooks then fills the struct pfv_solve_sharded takes.  PFV_ERR_UNSUPPORTED from the host-emulation
 * build or when librccl cannot be loaded. */
typedef struct pfv_rccl_comm pfv_rccl_comm;
pfv_status pfv_rccl_unique_id(char* id128);
pfv_status pfv_rccl_comm_create(pfv_ctx* h, const char* id128, int rank, int world, pfv_rccl_comm** out);
pfv_status pfv_rccl_set_halo_plan(pfv_rccl_comm* c, int n_peers, const int32_t* peers, const int64_t* send_ptr,
                                  const int32_t* send_idx, const int64_t* recv_ptr, const int32_t* recv_pos);
pfv_status pfv_rccl_hooks(pfv_rccl_comm* c, pfv_shard_hooks* out);
pfv_status pfv_rccl_stats(pfv_rccl_comm* c, int64_t* exchanges, int64_t* allreduces, int64_t* bytes_per_exchange);
const char* pfv_rccl_last_error(pfv_rccl_comm* c);
void pfv_rccl_comm_destroy(pfv_rccl_comm* c);

/* ---- Device-resident CSR matrices and the sparse algebra of the step after discretize --------
 * What MergedOperator.parse (numerics/ad/ad_utils.py:597-663 -> matrix_operations.csr_matrix_from_sparse_blocks) and the
 * scipy products / sums of the operator tree (assembled by EquationSystem.assemble, numerics/ad/equation_system.py:1579)
 * do on the host, kept in HBM: a pfv_csr is created from a discretization matrix of a handle without a host copy
 * (pfv_csr_from_matrix) or uploaded once (projections, divergences: pfv_csr_from_host), combined block-diagonally, by
 * products and by sums, and handed to pfv_solve as the active system (pfv_csr_set_system).  scipy's conventions, so
 * that results compare entry by entry: rows sorted by column without duplicates (required of inputs, guaranteed of
 * outputs), accumulation in the order of csr_matmat / csr_binop_csr with every product rounded on its own (same bits),
 * entries that come out exactly zero are not stored.  A pfv_csr belongs to the handle it was created on (same
 * device for all operands) and must be freed before that handle is destroyed. */
typedef struct pfv_csr pfv_csr;
pfv_status pfv_csr_from_host(pfv_ctx* h, int64_t nrows, int64_t ncols, const int32_t* indptr, const int32_t* indices,
                             const double* values, pfv_csr** out);
/* device-to-device copy of matrix `which` (pfv_matrix_id) of handle src */
pfv_status pfv_csr_from_matrix(pfv_ctx* h, pfv_ctx* src, int which, pfv_csr** out);
pfv_status pfv_csr_block_diag(pfv_ctx* h, int n, const pfv_csr* const* blocks, pfv_csr** out);
pfv_status pfv_csr_matmul(pfv_ctx* h, const pfv_csr* A, const pfv_csr* B, pfv_csr** out);   /* A B (rows fed by more than
                                                     4096 products take a slower path through global sorts) */
pfv_status pfv_csr_axpby(pfv_ctx* h, double alpha, const pfv_csr* A, double beta, const pfv_csr* B, pfv_csr** out);
pfv_status pfv_csr_transpose(pfv_ctx* h, const pfv_csr* A, pfv_csr** out);
/* scipy.sparse.bmat: nbr x nbc blocks, row-major, NULL = zero block of row_sizes[i] x col_sizes[j] */
pfv_status pfv_csr_bmat(pfv_ctx* h, int nbr, int nbc, const pfv_csr* const* blocks, const int64_t* row_sizes,
                        const int64_t* col_sizes, pfv_csr** out);
pfv_status pfv_csr_scale(pfv_csr* A, const double* row_scale, const double* col_scale);   /* in place; host arrays or NULL */
pfv_status pfv_csr_divide(pfv_csr* A, double s);   /* in place: every value divided by s (scipy's A / s) */
pfv_status pfv_csr_spmv(const pfv_csr* A, const double* x, double* y);                     /* host vectors */
pfv_status pfv_csr_spmv_device(const pfv_csr* A, const double* d_x, double* d_y);
pfv_status pfv_csr_info(const pfv_csr* A, int64_t* nrows, int64_t* ncols, int64_t* nnz);
pfv_status pfv_csr_get(const pfv_csr* A, int32_t* indptr, int32_t* indices, double* values);  /* any may be NULL */
/* (A, rhs) become the active system of h -- what pfv_solve / pfv_amg_setup work on -- without leaving the device */
pfv_status pfv_csr_set_system(pfv_ctx* h, const pfv_csr* A, const double* rhs, int rhs_on_device);
void pfv_csr_free(pfv_csr* A);

/* run this handle's work on an externally owned HIP stream (hipStream_t passed as void*, e.g.
 * torch.cuda.current_stream().cuda_stream) so that it is ordered with the caller's kernels and
 * RCCL collectives.  NULL is the legacy default stream (what torch's default stream is) -- the
 * handle's own stream is non-blocking and is NOT ordered with it; pfv_reset_stream goes back to
 * the handle's own stream. */
pfv_status pfv_set_stream(pfv_ctx* h, void* hip_stream);
pfv_status pfv_reset_stream(pfv_ctx* h);
pfv_status pfv_sync(pfv_ctx* h);

pfv_status pfv_get_stats(pfv_ctx* h, pfv_stats* out);
/* the same for a caller compiled against another revision of this header: writes at most `struct_size` bytes
 * (= the caller's sizeof(pfv_stats); fields are only ever appended), so a shorter struct is never overrun */
pfv_status pfv_get_stats_n(pfv_ctx* h, void* out, size_t struct_size);

/* Measurement hook for bench.py: average duration (ms, HIP events on the handle's stream)
 * of `reps` back-to-back launches of one kernel on the data currently in the handle.
 * kernel: 0 = CSR SpMV with A, 1 = interaction-region (node) kernel, 2 = face kernel,
 * 3 = AMG smoothing product, 4 = stream triad (no discretization needed). */
enum { PFV_KERNEL_SPMV_A = 0, PFV_KERNEL_NODE = 1, PFV_KERNEL_FACE = 2,
       PFV_KERNEL_AMG_SMOOTH = 3, /* finest-level smoothing product of the AMG cycle (needs a built hierarchy) */
       PFV_KERNEL_TRIAD = 4,      /* a = b + s c on 3 x 2^27 doubles (3.2 GB moved per launch): the measured
                                     device bandwidth next to the data-sheet 8 TB/s (SURVEY 8(d) "Metric") */
       PFV_KERNEL_READ = 5        /* read-only stream over 2^28 doubles (2.1 GB): the ceiling of the read-dominated
                                     SpMV kernels */ };
pfv_status pfv_time_kernel(pfv_ctx* h, int kernel, int reps, double* avg_ms);

/* Test hook: copy the leading `count` entries of an internal FP64 device array to the host (0 = the
 * per-node response tables, 1 = the boundary columns of the nodes on the boundary). */
pfv_status pfv_debug_copy(pfv_ctx* h, int which, double* dst, int64_t count);

#ifdef __cplusplus
}
#endif
#endif /* POREFV_H */
