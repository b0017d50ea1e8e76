// ctx.h — device-resident state behind a pfv_ctx handle.
#pragma once
#include "../../include/porefv.h"
#include "backend.h"

#include <algorithm>
#include <functional>
#include <memory>
#include <vector>

namespace pfv {

constexpr int kMaxFaceNodes = 8;   // nodes per face supported by the merge kernels
constexpr int kMaxCellFaces = 16;  // faces per cell supported by the A-pattern merge
constexpr int kMaxBlock = 255;     // local sub-face / sub-cell indices are stored as uint8
// block-size classes of the interaction-region kernel (one launch, one LDS size per class)
static const int kClassBounds[] = {4, 8, 12, 16, 24, 32, 40, 48, 64, 96, 128, 192, 255};
constexpr int kNumClasses = 13;
// PFV_PIPE_CHUNKS: runs of the node || face pipeline (0 / 1: off).  OFF -- measured on the MI355X at 2 M cells
// (profiles/r06_ab_runs.txt): the pipeline's span 14.9 ms (K = 4 / 8 / 16 / 30: 14.8 / 14.9 / 15.2 / 16.0) against
// 6.6 + 7.1 = 13.7 ms back to back: the two kernels share the device work-conservingly (the interaction-region kernel
// holds 154 of the 160 KB of LDS of a CU, the face kernel's wavefronts wait for whole workgroups of it to retire), and the
// ready-run-major face order the pipeline needs costs the face kernel 1.4 ms of table locality (8.5 against 7.1 ms).
constexpr int kPipeChunksDefault = 0;

// Per sub-face (face, node) record consumed by the face kernel: one 64-byte line, written whole by the
// interaction-region kernel (by the lane of the sub-cell the flux is evaluated from).  It replaces the
// chain face_nodes -> node -> {node_fptr, node_tptr, node_hptr} and carries the nd weights that turn rows
// of the node's response table into the flux row (mpfa_numeric.inc).
struct alignas(16) SfRec {
  int64_t toff;    // offset (doubles) of the node's response table in `tab`
  uint8_t n;       // sub-faces of the node = rows of the table before the e-row
  uint8_t ls;      // local index of this sub-face = its row
  uint8_t jstar;   // local sub-cell the flux is evaluated from
  uint8_t deg;     // cells of the node
  uint8_t r[3];    // local sub-faces of sub-cell jstar (one of them is ls)
  uint8_t pad;
  double om[3];    // omega_{h*} = nK_{h*} D^-1: flux row = direct term - sum_k om[k] * table row r[k]
  double nk[3];    // nK_{h*}: direct vector-source term
};
static_assert(sizeof(SfRec) == 64, "SfRec is one 64-byte record");

// Per face, in processing order (face_order): what the face kernel needs before it can issue its first
// dependent load -- one 32-byte record instead of the chain order -> {fn_ptr, flux indptr, bound indptr}.
struct alignas(16) FaceRec {  // dwords only: the kernel reads it with scalar loads
  int32_t f;        // face (-1: none)
  int32_t a;        // first sub-face (= fn_ptr[f])
  int32_t p0;       // first entry of the flux row
  int32_t q0;       // first entry of the bound_flux row
  uint32_t lens;    // entries of the flux row | entries of the bound_flux row << 16
  uint32_t nnf;     // nodes of the face
  int32_t pad[2];
};
static_assert(sizeof(FaceRec) == 32, "FaceRec is one 32-byte record");

struct CsrPattern {
  int64_t nrows = 0, ncols = 0, nnz = 0;
  Buf<int32_t> indptr, indices;
  int max_row = 0;
};

// "Windowed" companion of a CSR pattern for the SpMV kernels (spmv_win.inc): per block of 64 rows the
// sorted list of distinct columns the block touches (its x window, staged in LDS by the kernel) and
// per entry the 16-bit position of its column in that list.
struct WinCsr {
  Buf<int64_t> wptr64buf;
  const int64_t* wptr64 = nullptr;  // [nblk + 1] offsets into wcol
  const int32_t* wlen = nullptr;    // window lengths when the windows sit at a fixed stride (one-pass build); else wptr64 differences
  Buf<int32_t> wcol, cnt;
  Buf<uint16_t> lidx;
  int64_t nblk = 0, nrows = 0, nnz = 0;
  int wmax = 0;      // largest window
  bool ok = false;   // false: not built / some block exceeds the LDS budget -> plain CSR kernels
};

// a square system the Krylov solver can work on (flow: A = div flux; mechanics: A = div_nd stress)
struct LinSys {
  const CsrPattern* P = nullptr;
  double* val = nullptr;
  double* diag = nullptr;
  double* rhs = nullptr;
  int64_t n = 0;
  bool valid = false;
  const WinCsr* win = nullptr;  // optional (pfv_solve builds it)
};

struct Amg;      // amg.inc
struct BlockPc;  // amg.inc

struct pfv_ctx_impl {
  MemPool pool;     // first member: destroyed last, after every buffer has been handed back
  pfv_ctx_impl();
  ~pfv_ctx_impl();  // defined where Amg is complete (porefv.hip)
  int device = 0;
  stream_t stream{};      // stream all work of this handle is issued on
  stream_t own_stream{};  // the stream created with the handle (stream may point elsewhere, pfv_set_stream)
  Buf<double> sys_rowmax;            // largest off-diagonal |a_ij| of every row of A, left by assemble_system
  const double* sys_rowmax_for = nullptr;  // ... for this value array (nullptr: none)
  bool face_order_cell_major = false;  // face_order: faces of one first-side cell back to back (PFV_FACE_ORDER / PFV_FACE_CACHE)
  stream_t low_stream{};  // a stream of the LOWEST priority the device offers (PFV_NODE_LOWPRIO: the interaction-region
                          // kernel goes there, so that the short symbolic kernels beside it are dispatched first)
  stream_t aux_stream{};  // second stream of the handle: the interaction-region kernel runs there while the
                          // symbolic phase runs on `stream` (both only need the sub-cell topology)
  std::string err;
  Scratch scratch;

  // ---- grid (HBM, SoA [3][N]) ------------------------------------------------------
  bool have_grid = false;
  int nd = 0;
  int64_t nc = 0, nf = 0, nn = 0, ncf = 0, nsf = 0;  // ncf = nnz(cell_faces), nsf = nnz(face_nodes)
  Buf<double> nodes, fnorm, fcen, ccen, farea;
  // periodic faces (pfv_set_periodic): the side of face f that is not its native cell sees the face
  // centre fcen[f] - face_shift[f]
  bool periodic = false;
  Buf<double> face_shift;    // SoA [3][Nf]
  Buf<int32_t> face_native;  // [Nf] cell, or -1
  Buf<int32_t> cf_ptr, cf_idx, fn_ptr, fn_idx;
  Buf<int8_t> cf_sgn;

  // ---- parameters ---------------------------------------------------------------
  bool have_params = false;
  Buf<double> perm;   // [9][Nc]: perm[(3*i+j)*Nc + c] = K_ij(c)
  Buf<uint8_t> bcflag;
  Buf<double> robin;  // per face
  double eta = 0.0;
  bool have_eta_sub = false;
  Buf<double> eta_sub;  // per subface (face_nodes CSC order)

  // ---- sub-cell topology (SubcellTopology of the reference, node-major) ---------
  bool have_topology = false;
  int64_t nh = 0;             // sub-half-faces (cell, face, node)
  Buf<int32_t> node_hptr;     // [nn+1] segment of each node in the h arrays
  Buf<int32_t> h_cell;        // [nh] cell of h; h sorted by (node, cell, face)
  Buf<uint8_t> h_lf;          // [nh] local index of h's face among the node's faces
  Buf<int8_t> h_sgn;          // [nh] cell_faces sign
  Buf<uint8_t> h_first;       // [nh] 1 if h is the side fluxes are evaluated from
  Buf<int32_t> node_fptr;     // [nn+1] segment of each node in node_sf
  Buf<int32_t> node_sf;       // [nsf] global subface ids (= face_nodes CSC positions), by (node, face)
  Buf<int32_t> node_face;     // [nsf] face id of every entry of node_sf (= sf_face[node_sf[.]]): one load instead of two
  Buf<int32_t> sf_face;       // [nsf] face of a subface
  Buf<uint8_t> sf_ls;         // [nsf] local index of the subface within its node
  Buf<int32_t> face_nsides;   // [nf] 1 = boundary face
  Buf<int32_t> node_cells;    // [nh/nd] cells around each node, ascending (segment = node_hptr/nd)
  Buf<int32_t> node_bptr;     // [nn+1] boundary faces around each node
  Buf<int32_t> node_bfaces;   //   face ids, ascending
  Buf<uint8_t> node_bls;      //   their local subface index
  Buf<int64_t> node_tptr;     // [nn+1] offset of the node's response table in `tab`: (n + 1) rows of ldt doubles
  Buf<int64_t> node_tbptr;    // [nn+1] offset of the node's boundary columns in `tabb`: 2 x n x nb doubles
  Buf<uint8_t> flux_colpairs; // for every flux column: the (node of face, cell) pair per node of the face, 0xff = none;
                              // row of face f at f * cp_stride (cp_stride > 0: fixed stride, as the one-pass symbolic
                              // phase leaves it) or at indptr[f] * max_face_nodes (cp_stride = 0: compact)
  int64_t cp_stride = 0;
  Buf<uint8_t> node_active;   // [nn] partial discretization: nodes of the requested faces
  Buf<int32_t> face_subset;   // partial discretization: the requested faces
  bool biot_rows_complete = false;  // every row of the coupling terms holds a current value (updates need it)
  Buf<int32_t> cell_subset;   // pfv_biot_discretize_faces: cells whose rows are recomputed
  Buf<int32_t> face_order;    // [nf] faces along a Morton curve of their centres: processing order of the
                              //      face kernels, so that faces sharing nodes run close in time (L2 reuse)
  double bbox_lo[3] = {0, 0, 0}, bbox_hi[3] = {1, 1, 1};
  Buf<int32_t> node_order;    // [nn] nodes sorted by block-size class
  std::vector<int64_t> class_begin;  // host: first position of each size class in node_order
  // node || face pipeline (topology.inc): runs of the largest size class, the faces that are ready after each of them
  int pipe_chunks = 0, pipe_class = -1;     // 0: no pipeline on this grid
  std::vector<int64_t> pipe_node_begin;     // [K + 1] positions in node_order: run q = [begin[q], begin[q + 1]) , q = 0 .. K - 1
  std::vector<int32_t> pipe_face_begin;     // [K + 2] positions in face_order: faces ready after run q (0: the other classes) = [begin[q], begin[q + 1])
  Buf<uint8_t> pipe_node_chunk;             // [nn] 0: not in the pipelined class, q + 1: run q of it
  int max_block = 0, max_deg = 0, max_face_nodes = 0, max_cell_faces = 0, max_bnd_per_node = 0;
  int64_t sum_block_sq = 0;   // sum of n(v)^2 (statistics)
  int64_t tab_len = 0, tabb_len = 0;

  // ---- per-node numeric results consumed by the face kernel ---------------------
  // Response table of node v (n sub-faces, nh = nd * deg sub-half-faces, row stride ldt = nh rounded up
  // to even): row r < n holds TV[r][nd*j + b] = d lambda_r / d (vector source component b of cell j)
  // = (A^-1 G)[r][.]; row n holds e[nd*j + b] = (D_j^-1 1)[b], which turns a vector-source entry into the
  // cell-pressure entry (flux = vector_source . blockdiag(e)).
  Buf<double> tab;            // [tab_len]
  Buf<double> tabb;           // [tabb_len] boundary nodes: T[:, lb] beta_lb (n x nb), then A^-1[:, lb] beta_lb
  Buf<SfRec> sf_rec;          // [nsf] see SfRec
  Buf<int32_t> face_ctr;      // work counters of the face kernel, one per XCD at 64-byte spacing (PFV_FACE_DYN)
  unsigned face_ctr_next = 0;          // counter set of the next face-kernel launch (32 sets, cycled)
  Buf<FaceRec> face_rec;      // [nf] see FaceRec (written at the end of the symbolic phase)
  Buf<int32_t> status;        // [4] device status words: singular node, ...

  // ---- outputs --------------------------------------------------------------------
  bool have_symbolic = false, have_numeric = false, have_system = false;
  bool rows_complete = false;  // every row of the six MPFA matrices holds a discretization (maybe of older parameters)
  CsrPattern pat_flux, pat_bound, pat_vs, pat_A;  // bound_pressure_* share flux / bound patterns
  bool vs_indices_pending = false;  // pat_vs.indices not written yet (topology.inc: ensure_vs_indices)
  bool vs_implicit = false;         // vector_source / bound_pressure_vector_source have >= 2^31 entries (nd x the flux pattern:
                                    // beyond ~6.3 M tetrahedra): their int32 row pointers and indices are never formed -- the
                                    // values are addressed through the flux pattern (entry p, component k -> nd p + k, 64-bit),
                                    // products take spmv_vs_implicit, exports go by rows (pfv_get_matrix_rows)
  Buf<double> val[PFV_NUM_MATS];
  bool filled[PFV_NUM_MATS] = {};
  Buf<double> rhs, diag, xsol, face_tmp, vec_in;
  Buf<double> kry[10];
  Buf<double> red;  // reduction partials
  Buf<double> gmres_basis, gmres_small;  // Arnoldi basis [(m+1) n]; Hessenberg / rotations / scalars

  // ---- MPSA -------------------------------------------------------------------------
  bool have_mpsa_params = false, have_mpsa_numeric = false, have_mpsa_symbolic = false, have_mech_system = false;
  Buf<double> stiff;                 // [81][Nc]: stiff[(9*q + r)*Nc + c] = C_qr(c)
  Buf<double> cvol;                  // [nc] cell volumes (node-volume weights of the asymmetric part)
  Buf<uint8_t> bc_dirbits, bc_neubits, bc_robbits;
  Buf<double> mpsa_robw;             // [nd*nd][Nf] Robin weights
  bool have_mpsa_robin = false;
  bool have_mpsa_eta_sub = false;  // continuity points per sub-face (pfv_mpsa_set_subface_eta)
  bool mpsa_hf_on = false;         // reconstruction_eta given (pfv_mpsa_set_reconstruction_eta)
  double mpsa_hf_eta = 0.0;
  Buf<double> mpsa_hf_eta_sub;     // [Nsf] reconstruction_eta per sub-face (used as given, also on the boundary)
  bool have_mpsa_hf_eta_sub = false;
  Buf<double> Et2, Etb2;           // traces reconstructed at the hf_eta points (layout of Et / Etb)
  Buf<double> mpsa_eta_sub;
  Buf<double> mpsa_basis;            // [nd*nd][Nf] boundary basis (BoundaryConditionVectorial.basis)
  bool have_mpsa_basis = false;
  Buf<double> mpsa_basis_sub;        // [nd*nd][Nsf] the same per sub-face (conditions per sub-face, mpsa.py:712-720)
  bool have_mpsa_basis_sub = false;
  Buf<char> mpsa_scratch;            // global-memory work space of interaction regions too large for the LDS
  Buf<long long> mpsa_clk;           // timing lab (PFV_MPSA_CLOCK): s_memtime stamps of one workgroup
  // conditions per sub-face (mpsa.py:712-720): flags / weights per sub-face replace the per-face ones, stress and
  // bound_stress keep sub-face rows, the boundary matrices sub-face columns
  bool mpsa_subface_bc = false, have_mpsa_sub_symbolic = false;
  Buf<uint8_t> bc_dirbits_sub, bc_neubits_sub, bc_robbits_sub;
  Buf<double> mpsa_robw_sub;         // [nd*nd][Nsf]
  bool have_mpsa_robin_sub = false;
  CsrPattern pat_sstress, pat_sbstress, pat_bdf_s, pat_sbdface;  // expansions of pat_sflux / pat_sbound; (Nf x Nsf) and its expansion
  Buf<uint8_t> bdf_src;              // [2 x nnz(pat_bdf_s)] (node position in the face, local boundary sub-face) of every entry
  double mpsa_eta = 0.0;
  Buf<int32_t> cell_nnodes;          // [nc] distinct nodes of a cell (node-volume weights)
  Buf<int64_t> node_eptr, node_ebptr;  // [nn+1] offsets of the per-node expanded rows (cells / boundary faces)
  Buf<double> Es, Et, Esb, Etb;      // per node: (nd*nsf) x (nd*deg) stress / trace rows; x (nd*nb) boundary
  CsrPattern pat_stress, pat_bstress, pat_Am;
  // ---- Biot coupling terms (biot.inc) ------------------------------------------------
  int biot_nalpha = 0;               // coupling tensors set by pfv_biot_set_alphas
  Buf<double> biot_alpha;            // [nalpha][9][Nc]
  bool have_biot_symbolic = false, have_biot_numeric = false;
  Buf<int32_t> cn_ptr, cn_idx;       // cell -> distinct nodes (CSR)
  Buf<int64_t> node_dptr, node_dbptr, node_cptr;  // per-node offsets of the divergence rows
  int64_t biot_tot_p = 0, biot_tot_d = 0, biot_tot_db = 0, biot_tot_c = 0;
  Buf<double> bEsp, bEtp, bDD, bBDD, bCONS;       // per node and key, see mpsa.inc
  CsrPattern pat_sg, pat_dd, pat_bdd;  // (nd Nf x Nc), (Nc x nd Nc), (Nc x nd Nf); consistency uses pat_A
  Buf<double> biot_val[5 * 8];         // values: term-major (PFV_BIOT_* order), then key (<= 8 keys)
  CsrPattern pat_user;               // pfv_set_system
  CsrPattern pat_bpf;                // TPFA: bound_pressure_face (diagonal of the Dirichlet / Neumann faces)
  bool rows_complete_m = false;      // every row of the four MPSA matrices holds a discretization
  bool subface_bc = false;           // conditions per sub-face: matrices 0-3 have sub-face rows
  Buf<uint8_t> bcflag_s;
  Buf<double> robin_s;
  CsrPattern pat_sflux, pat_sbound;
  bool have_sub_symbolic = false;
  bool tpfa_mode = false;            // matrices 0-5 hold a TPFA discretization
  Buf<double> rhs_u, diag_u;
  std::vector<int32_t> mpsa_class_lds;  // LDS bytes of the largest node per block-size class
  Buf<double> rhs_m, diag_m;
  bool vectors_on_device = false;    // pfv_set_vectors_on_device
  LinSys active;                     // what pfv_solve / pfv_get_rhs operate on
  WinCsr win_rows;                   // window of the leading rows of a system matrix (pfv_spmv_device_rows)
  const int32_t* win_rows_for = nullptr;
  int64_t win_rows_n = 0;
  WinCsr win_sys, win_block;         // SpMV windows of the active system / of the leading block (pfv_amg_setup)
  // The solve runs on a copy of the grid systems renumbered along a space-filling curve of the cell
  // centres (reorder.inc): the numbering of the grid generator decides how local the SpMV gathers are.
  bool active_is_grid = false;       // active rows = cells (x active_bs) in the grid's numbering
  bool have_cell_order = false;
  bool cell_order_identity = false;  // the grid's own numbering already follows the curve: solve in place
  Buf<int32_t> cell_perm, cell_iperm;  // new position -> cell, cell -> new position
  CsrPattern pat_perm;
  Buf<double> val_perm, diag_perm, rhs_perm, x_perm;
  const double* perm_for_val = nullptr;  // the values the renumbered copy was made from (nullptr: stale)
  const int32_t* win_for = nullptr;      // the index array win_sys was built for (nullptr: stale)
  bool win_sys_prebuilt = false;         // win_sys was built for pat_A by the discretize call (beside the face kernel)
  bool win_rows_prebuilt = false;        // same for win_rows (sharded solve: the rows of the owned cells)
  int active_bs = 1;                 // unknowns per cell of the active system (AMG block size)
  int precond = 0;                   // PFV_PRECOND_*
  std::unique_ptr<Amg> amg;          // hierarchy of the active system (rebuilt when the system changes)
  const double* amg_for_val = nullptr;  // the matrix values the hierarchy was built from
  Buf<unsigned long long> csum_work;       // partial sums of launch_pattern_checksum
  unsigned long long pat_A_checksum = 0;   // checksum of pat_A's index arrays left by the symbolic phase (0: none)
  unsigned long long win_sys_checksum = 0; // ... of the pattern win_sys was built for (0: unknown)
  unsigned long long win_rows_checksum = 0; // ... of the pattern whose leading win_rows_n rows win_rows covers
  unsigned long long symbolic_epoch = 0;  // bumped by every symbolic phase: saved AMG aggregates die with it
  unsigned long long topo_key = 0;   // topology_digest of the topology on the handle (0: not taken)
  Buf<uint8_t> asm_pos, asm_diagpos;       // assemble_system: position in A's row of every flux entry of every (cell, face); diagonal
  unsigned long long asm_pos_key = 0;      // ... the symb_key they were recorded under (0: none)
  int64_t asm_pos_cells = 0;
  Buf<uint8_t> asm_pos_m, asm_diagpos_m;   // the same for the mechanics system (mpsa.inc: mpsa_assemble_system)
  unsigned long long asm_pos_m_key = 0;
  int64_t asm_pos_m_rows = 0;
  unsigned long long symb_key = 0;   // ... of the topology the symbolic outputs on the handle were built from (0: none / replaced)
  std::unique_ptr<BlockPc> block_pc;  // pfv_set_block_preconditioner
  std::unique_ptr<Amg> amg_block;    // pfv_amg_setup: hierarchy of the leading block (sharded solves)
  CsrPattern pat_block;
  Buf<double> val_block;
  // set only while pfv_solve_sharded runs: the caller's exchange hooks and work space
  const pfv_shard_hooks* shard = nullptr;
  bool shard_overlap = false;        // sharded SpMV: halo exchange on aux_stream beside the interior row blocks
  Buf<int32_t> shard_blocks;         // [interior row blocks | boundary row blocks] of win_rows
  int64_t shard_n_interior = 0, shard_n_boundary = 0;
  LaggedScalar lag_rr;               // residual norm of the Krylov loop, read half an iteration late (linalg.inc)
  Buf<int32_t> node_redo;            // nodes the lean MPFA launches hand to the full body (mpfa_numeric.inc)
  std::function<void(stream_t, int64_t)> node_redo_launch;  // ... and the launch that takes them (set by launch_node_kernel)
  int64_t stats_node_redo = 0;
  Buf<int32_t> mpsa_redo;            // nodes the lean MPSA launches hand to the full body (mpsa.inc)
  Buf<int32_t> mpsa_cell_exp;        // binary exponent of every cell's stiffness scale (mpsa.inc: mpsa_contrast_scan)
  Buf<uint8_t> mpsa_wide_flag;       // [nn] 1: interaction region of high stiffness contrast -> double-double body
  Buf<int32_t> mpsa_wide_list;       // the flagged nodes (order of arrival: every region writes its own rows only)
  Buf<int32_t> mpsa_wide_stat;       // [0] their number, [1] largest exponent difference over all regions
  int64_t stats_mpsa_redo = 0;
  Buf<double> red5;                  // block partials of the sharded BiCGStab's five merged sums
  double* shard_work = nullptr;      // [2 * shard_nloc + 8]: the two SpMV inputs (owned + halo entries), reduction scratch
  int64_t shard_nloc = 0;

  pfv_stats stats{};

  const CsrPattern& pattern_of(int which) const {
    switch (which) {
      case PFV_MAT_FLUX:
      case PFV_MAT_BOUND_PRESSURE_CELL:
        return subface_bc ? pat_sflux : pat_flux;
      case PFV_MAT_FLUX_JACOBIAN:
        return pat_flux;
      case PFV_MAT_BOUND_PRESSURE_FACE:
        return subface_bc ? pat_sbound : (tpfa_mode ? pat_bpf : pat_bound);
      case PFV_MAT_BOUND_FLUX:
        return subface_bc ? pat_sbound : pat_bound;
      case PFV_MAT_VECTOR_SOURCE:
      case PFV_MAT_BOUND_PRESSURE_VECTOR_SOURCE:
        return pat_vs;
      case PFV_MAT_STRESS:
        return mpsa_subface_bc ? pat_sstress : pat_stress;
      case PFV_MAT_BOUND_DISPLACEMENT_CELL:
        return pat_stress;
      case PFV_MAT_BOUND_STRESS:
        return mpsa_subface_bc ? pat_sbstress : pat_bstress;
      case PFV_MAT_BOUND_DISPLACEMENT_FACE:
        return mpsa_subface_bc ? pat_sbdface : pat_bstress;
      case PFV_MAT_MECH_SYSTEM:
        return pat_Am;
      case PFV_MAT_USER_SYSTEM:
        return pat_user;
      default:
        return pat_A;
    }
  }
};

}  // namespace pfv

struct pfv_ctx : pfv::pfv_ctx_impl {};
