// porefv.hip — C ABI (include/porefv.h) of the MI355X-native MPFA-O assembly + solve.
//
// Build (product):   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC porefv.hip
// Build (emulation): g++ -x c++ -DPFV_EMULATE -O2 -std=c++17 -shared -fPIC porefv.hip
//                    (test infrastructure; see backend.h)
#include <cmath>
#include <cstring>

#include "ctx.h"
namespace pfv {
static int rccl_exchange_halo(void* user, double* d_x, void* stream);  // rccl_hooks.inc
}
#include "topology.inc"
#ifndef PFV_EMULATE
#include "gj_mfma.inc"
#endif
#include "mpfa_numeric.inc"
#include "linalg.inc"
#include "reorder.inc"
#include "dd.h"
#include "mpsa.inc"
#include "tpfa.inc"
#include "biot.inc"
#include "ad_flux.inc"

namespace pfv {
#ifndef PFV_EMULATE
// stream triad a = b + s c (PFV_KERNEL_TRIAD): the measured device bandwidth next to the data-sheet figure
__global__ void __launch_bounds__(256) k_triad(int64_t n2, D2* __restrict__ a, const D2* __restrict__ b,
                                               const D2* __restrict__ c) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n2; i += 4 * stride) {
    D2 x[4], y[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      x[u] = b[i + u * stride];
      y[u] = c[i + u * stride];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      D2 r;
      r.x = x[u].x + 0.5 * y[u].x;
      r.y = x[u].y + 0.5 * y[u].y;
      a[i + u * stride] = r;
    }
  }
  for (; i < n2; i += stride) {
    D2 r;
    r.x = b[i].x + 0.5 * c[i].x;
    r.y = b[i].y + 0.5 * c[i].y;
    a[i] = r;
  }
}
// read-only stream (PFV_KERNEL_READ): sum of 2^28 doubles, 16 bytes per lane, four loads in flight; the sums go
// to a small array so that nothing is optimised away
__global__ void __launch_bounds__(256) k_read_stream(int64_t n2, const D2* __restrict__ b, double* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double acc = 0.0;
  for (; i + 3 * stride < n2; i += 4 * stride) {
    D2 x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) x[u] = b[i + u * stride];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += x[u].x + x[u].y;
  }
  for (; i < n2; i += stride) acc += b[i].x + b[i].y;
  if (acc == 1.2345e300) out[0] = acc;  // (never true for the zero-filled buffer: keeps the loads alive)
}
#endif
pfv_ctx_impl::pfv_ctx_impl() = default;
pfv_ctx_impl::~pfv_ctx_impl() = default;
}  // namespace pfv

using pfv::be_d2h;
using pfv::be_h2d;
using pfv::Error;

namespace {

template <class F>
pfv_status guarded(pfv_ctx* h, F&& body) {
  if (!h) return PFV_ERR_ARGUMENT;
  pfv::PoolScope pool_scope(&h->pool);
  try {
#ifndef PFV_EMULATE
    PFV_HIP_CHECK(hipSetDevice(h->device));
#endif
    body();
    return PFV_OK;
  } catch (const Error& e) {
    h->err = e.what();
    return (pfv_status)e.status;
  } catch (const std::exception& e) {
    h->err = e.what();
    return PFV_ERR_HIP;
  }
}

void require(bool ok, const char* msg) {
  if (!ok) throw Error(PFV_ERR_ARGUMENT, msg);
}

bool pattern_ready(const pfv_ctx* h, int which) {
  if (which == PFV_MAT_USER_SYSTEM) return h->filled[which];
  if (which == PFV_MAT_FLUX_JACOBIAN) return h->have_symbolic;
  return which >= PFV_MAT_STRESS ? h->have_mpsa_symbolic : h->have_symbolic;
}

template <class T>
void upload(pfv::Buf<T>& buf, const T* host, size_t n, pfv::stream_t s) {
  buf.ensure(n);
  be_h2d(buf.p, host, n * sizeof(T), s);
}

}  // namespace

namespace {
// the leading rows of a pattern as a pattern of its own (shares the index arrays, frees nothing)
struct RowsView {
  pfv::CsrPattern V;
  RowsView(const pfv::CsrPattern& P, int64_t nrows) {
    V.nrows = nrows;
    V.ncols = P.ncols;
    V.nnz = P.nnz;  // (the window's entry positions index the full arrays)
    V.max_row = P.max_row;
    V.indptr.p = P.indptr.p;
    V.indices.p = P.indices.p;
  }
  ~RowsView() {
    V.indptr.p = nullptr;
    V.indices.p = nullptr;
  }
};
}  // namespace

extern "C" {

int pfv_is_device_build(void) {
#ifdef PFV_EMULATE
  return 0;
#else
  return 1;
#endif
}

pfv_status pfv_create(int device, pfv_ctx** out) {
  if (!out) return PFV_ERR_ARGUMENT;
  *out = nullptr;
  pfv_ctx* h = nullptr;
  try {
    h = new pfv_ctx();
    h->device = device;
#ifndef PFV_EMULATE
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
      delete h;
      return PFV_ERR_HIP;  // no GPU: the product library never falls back to the host
    }
    if (device < 0 || device >= count) {
      delete h;
      return PFV_ERR_ARGUMENT;
    }
    PFV_HIP_CHECK(hipSetDevice(device));
    PFV_HIP_CHECK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->own_stream = h->stream;
    PFV_HIP_CHECK(hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking));
    {
      int least = 0, greatest = 0;
      if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest) {
        if (hipStreamCreateWithPriority(&h->low_stream, hipStreamNonBlocking, least) != hipSuccess) h->low_stream = nullptr;
      }
      (void)hipGetLastError();
    }
#endif
  } catch (...) {
    delete h;
    return PFV_ERR_HIP;
  }
  *out = h;
  return PFV_OK;
}

void pfv_destroy(pfv_ctx* h) {
  if (!h) return;
#ifndef PFV_EMULATE
  (void)hipSetDevice(h->device);
  if (h->stream) {
    (void)hipStreamSynchronize(h->stream);
  }
  hipStream_t s = h->own_stream, s2 = h->aux_stream, s3 = h->low_stream;
  if (s2) (void)hipStreamSynchronize(s2);
  if (s3) (void)hipStreamSynchronize(s3);
  {
    pfv::PoolScope pool_scope(&h->pool);
    delete h;
  }
  if (s) (void)hipStreamDestroy(s);
  if (s2) (void)hipStreamDestroy(s2);
  if (s3) (void)hipStreamDestroy(s3);
#else
  {
    pfv::PoolScope pool_scope(&h->pool);
    delete h;
  }
#endif
}

const char* pfv_last_error(pfv_ctx* h) { return h ? h->err.c_str() : "null handle"; }

// right-hand-side / solution vectors of the assemble and solve calls: host memory by default, device
// memory after pfv_set_vectors_on_device(h, 1)
static void vec_in(pfv_ctx* h, double* dst, const double* src, size_t n) {
  if (h->vectors_on_device) pfv::be_d2d(dst, src, n * sizeof(double), h->stream);
  else be_h2d(dst, src, n * sizeof(double), h->stream);
}

pfv_status pfv_set_vectors_on_device(pfv_ctx* h, int on) {
  return guarded(h, [&] { h->vectors_on_device = on != 0; });
}

pfv_status pfv_set_grid(pfv_ctx* h, int nd, int64_t nc, int64_t nf, int64_t nn, const double* nodes,
                        const int32_t* cf_indptr, const int32_t* cf_indices, const int8_t* cf_sign,
                        const int32_t* fn_indptr, const int32_t* fn_indices,
                        const double* face_normals, const double* face_centers,
                        const double* cell_centers, const double* face_areas) {
  return guarded(h, [&] {
    require(nd >= 1 && nd <= 3, "nd must be 1, 2 or 3 (1-D grids: TPFA only, as in the reference, mpfa.py:690-712)");
    require(nc > 0 && nf > 0 && nn > 0, "empty grid");
    require(nodes && cf_indptr && cf_indices && cf_sign && fn_indptr && fn_indices && face_normals &&
                face_centers && cell_centers && face_areas,
            "null grid array");
    require(nc < (int64_t(1) << 31) && nf < (int64_t(1) << 31) && nn < (int64_t(1) << 31), "grid too large for int32 ids");
    auto s = h->stream;
    pfv::be_sync(s);
    h->pool.trim();  // a new grid: the cached block sizes are of no use any more
    h->nd = nd;
    h->nc = nc;
    h->nf = nf;
    h->nn = nn;
    h->ncf = cf_indptr[nc];
    h->nsf = fn_indptr[nf];
    require(cf_indptr[0] == 0 && fn_indptr[0] == 0, "indptr must start at 0");
    for (int d = 0; d < 3; ++d) {
      double lo = face_centers[(size_t)d * nf], hi = lo;
      for (int64_t f = 1; f < nf; ++f) {
        const double x = face_centers[(size_t)d * nf + f];
        lo = x < lo ? x : lo;
        hi = x > hi ? x : hi;
      }
      h->bbox_lo[d] = lo;
      h->bbox_hi[d] = hi;
    }
    upload(h->nodes, nodes, 3 * (size_t)nn, s);
    upload(h->fnorm, face_normals, 3 * (size_t)nf, s);
    upload(h->fcen, face_centers, 3 * (size_t)nf, s);
    upload(h->ccen, cell_centers, 3 * (size_t)nc, s);
    upload(h->farea, face_areas, (size_t)nf, s);
    upload(h->cf_ptr, cf_indptr, (size_t)nc + 1, s);
    upload(h->cf_idx, cf_indices, (size_t)h->ncf, s);
    upload(h->cf_sgn, cf_sign, (size_t)h->ncf, s);
    upload(h->fn_ptr, fn_indptr, (size_t)nf + 1, s);
    upload(h->fn_idx, fn_indices, (size_t)h->nsf, s);
    h->have_grid = true;
    h->periodic = false;
    h->have_cell_order = false;
    h->perm_for_val = nullptr;
    h->win_for = h->win_rows_for = nullptr;
    h->have_topology = h->have_symbolic = h->have_numeric = h->have_system = false;
    h->topo_key = h->symb_key = 0;
    h->win_sys_prebuilt = h->win_rows_prebuilt = false;
    h->biot_rows_complete = false;
    h->rows_complete = false;
    h->rows_complete_m = false;
    h->have_sub_symbolic = h->have_mpsa_sub_symbolic = false;
    h->subface_bc = h->mpsa_subface_bc = false;
    h->tpfa_mode = false;
    h->have_mpsa_numeric = h->have_mpsa_symbolic = h->have_mech_system = false;
    h->active.valid = false;
    for (bool& f : h->filled) f = false;
  });
}

pfv_status pfv_set_periodic(pfv_ctx* h, const int32_t* native_cell, const double* shift) {
  return guarded(h, [&] {
    require(h->have_grid, "pfv_set_grid first");
    require((native_cell == nullptr) == (shift == nullptr), "give both arrays, or neither to clear");
    h->have_numeric = h->have_system = false;
    h->win_sys_prebuilt = h->win_rows_prebuilt = false;
    h->rows_complete = false;
    if (!native_cell) {
      h->periodic = false;
      return;
    }
    upload(h->face_native, native_cell, (size_t)h->nf, h->stream);
    upload(h->face_shift, shift, 3 * (size_t)h->nf, h->stream);
    h->periodic = true;
  });
}

pfv_status pfv_mpfa_set_params(pfv_ctx* h, const double* perm_33n, const uint8_t* bc_flags,
                               const double* robin_weight, double eta, const double* eta_subface) {
  return guarded(h, [&] {
    require(h->have_grid, "pfv_set_grid must be called first");
    require(perm_33n && bc_flags, "null parameter array");
    auto s = h->stream;
    upload(h->perm, perm_33n, 9 * (size_t)h->nc, s);
    upload(h->bcflag, bc_flags, (size_t)h->nf, s);
    if (robin_weight) {
      upload(h->robin, robin_weight, (size_t)h->nf, s);
    } else {
      std::vector<double> ones((size_t)h->nf, 1.0);
      upload(h->robin, ones.data(), ones.size(), s);
    }
    h->eta = eta;
    h->have_eta_sub = eta_subface != nullptr;
    if (eta_subface) upload(h->eta_sub, eta_subface, (size_t)h->nsf, s);
    h->have_params = true;
    h->subface_bc = false;
    h->have_numeric = h->have_system = false;
    h->win_sys_prebuilt = h->win_rows_prebuilt = false;
  });
}

pfv_status pfv_mpfa_set_permeability(pfv_ctx* h, const double* perm_33n) {
  return guarded(h, [&] {
    require(h->have_grid && h->have_params, "pfv_mpfa_set_params first");
    require(perm_33n != nullptr, "null parameter array");
    h->perm.ensure(9 * (size_t)h->nc);
    vec_in(h, h->perm.p, perm_33n, 9 * (size_t)h->nc);
    h->have_numeric = h->have_system = false;
    h->win_sys_prebuilt = h->win_rows_prebuilt = false;
  });
}

pfv_status pfv_mpfa_set_subface_bc(pfv_ctx* h, const uint8_t* bc_flags_sub, const double* robin_weight_sub) {
  return guarded(h, [&] {
    require(h->have_grid && h->have_params, "pfv_mpfa_set_params first");
    h->subface_bc = false;
    h->have_numeric = h->have_system = false;
    h->win_sys_prebuilt = h->win_rows_prebuilt = false;
    h->rows_complete = false;
    if (!bc_flags_sub) return;
    const size_t nsf = (size_t)h->nsf;
    upload(h->bcflag_s, bc_flags_sub, nsf, h->stream);
    if (robin_weight_sub) {
      upload(h->robin_s, robin_weight_sub, nsf, h->stream);
    } else {
      std::vector<double> ones(nsf, 1.0);
      upload(h->robin_s, ones.data(), nsf, h->stream);
    }
    h->subface_bc = true;
  });
}

namespace {
// A rebuilt topology (PFV_DISCR_REBUILD_TOPOLOGY, the timed step of bench.py) against the symbolic outputs the handle
// still holds: the patterns of the six matrices and of A, the column -> pair maps and the face records are functions of
// the topology alone.  When the digest of the topology just built equals the one they were built from (topology.inc:
// topology_digest; sizes are part of it) they are kept: nothing build_symbolic would write differs from what is there.
// The topology itself is ALWAYS rebuilt -- that is what the flag asks for, and what the digest is taken of; a miss (new
// grid, other boundary, first call) runs the symbolic phase as before.  PFV_SYMB_REUSE=0: never keep (the cold step).
// Ref: the phase this stands for is SubcellTopology + the index work of the SpGEMM chain, _fvutils.py:51-172.
bool symbolic_outputs_still_valid(pfv_ctx* h) {
  h->stats.symbolic_reused = 0;
  if (pfv::env_int("PFV_SYMB_REUSE", 1) == 0) {
    h->topo_key = 0;
    return false;
  }
  h->topo_key = pfv::topology_digest(*h);
  const bool keep = h->have_symbolic && !h->tpfa_mode && h->symb_key != 0 && h->symb_key == h->topo_key &&
                    h->pat_flux.nrows == h->nf && h->pat_A.nrows == h->nc;
  if (keep) {
    // what build_symbolic's prologue does for the VALUES: they belong to the previous discretization
    for (int m = PFV_MAT_FLUX; m <= PFV_MAT_SYSTEM; ++m) h->filled[m] = false;
    h->stats.symbolic_reused = 1;
  }
  return keep;
}
}  // namespace

pfv_status pfv_mpfa_discretize(pfv_ctx* h, uint32_t flags) {
  return guarded(h, [&] {
    require(h->have_grid && h->have_params, "grid and parameters must be set before discretize");
    require(h->nd >= 2, "MPFA needs a 2-D or 3-D grid (1-D: pfv_tpfa_discretize)");
    auto s = h->stream;
    pfv::Timer tm, tall;
    tall.start(s);
    bool node_done = false;
    bool face_done = false;       // the face kernel ran inside the node || face pipeline
    const bool with_vs = !(flags & PFV_DISCR_SKIP_VECTOR_SOURCE);
    bool cells_deferred = false;  // part 2 of the symbolic phase (pattern of A) still to be built
    if (!h->have_topology || !h->have_symbolic || (flags & PFV_DISCR_REBUILD_TOPOLOGY)) {
      tm.start(s);
      pfv::build_topology(*h);
      h->stats.topology_ms = tm.stop(s);
      tm.start(s);
      const bool keep_symbolic = symbolic_outputs_still_valid(h);
      if (keep_symbolic) {
        h->stats.symbolic_ms = tm.stop(s);  // (the digest and its read-back)
        node_done = false;                  // the interaction-region kernel has the device to itself, below
      } else {
#ifndef PFV_EMULATE
      // The symbolic phase (CSR patterns) and the interaction-region kernel (local inverses) both
      // depend on the sub-cell topology only, and both are bound by latency at low occupancy, not by
      // bandwidth: the node kernel goes to the handle's second stream and shares the CUs with the
      // symbolic kernels.  Its buffers live in the handle (nothing it touches passes through the block
      // cache while the other stream runs); the status words of the two phases are disjoint.
      if (h->aux_stream && pfv::env_int("PFV_OVERLAP_NODE", 1) != 0) {
        // (PFV_NODE_LOWPRIO=1: on the handle's lowest-priority stream -- the symbolic kernels, short and many, then win
        // the dispatch slots the long kernel frees instead of queueing behind its ~300 k wavefronts)
        pfv::stream_t ns = (h->low_stream && pfv::env_int("PFV_NODE_LOWPRIO", 0) != 0) ? h->low_stream : h->aux_stream;
        pfv::StreamFork fork(s, ns);   // ns waits for everything enqueued on s so far
        pfv::Timer tn;
        tn.start(ns);
        pfv::launch_node_kernel(*h, nullptr, ns);
        tn.mark(ns);
        tm.start(s);
        // (the pattern of A is only needed by the assembly: PFV_SYMB_DEFER_CELLS=1 builds it beside the face kernel,
        // below -- measured: interaction-region span 14.2 -> 13.2 ms, face span 7.2 -> 9.1 ms, step +0.5 ms: off)
        cells_deferred = !h->subface_bc && pfv::env_int("PFV_SYMB_DEFER_CELLS", 0) != 0;
        pfv::build_symbolic(*h, cells_deferred ? 1 : 3);
        h->stats.symbolic_ms = tm.stop(s);
        fork.join();                              // s waits for the node kernel
        h->stats.node_ms = tn.elapsed_after_sync();
        pfv::check_node_status(*h, s);
        node_done = true;
      }
#endif
      if (!node_done) {
        tm.start(s);
        pfv::build_symbolic(*h);
        h->stats.symbolic_ms = tm.stop(s);
      }
      h->tpfa_mode = false;
      h->have_sub_symbolic = h->have_mpsa_sub_symbolic = false;
      }  // !keep_symbolic
    }
#ifndef PFV_EMULATE
    h->stats.pipeline_runs = 0;
    if (!node_done && !h->subface_bc && h->aux_stream && h->pipe_chunks > 1 && h->have_symbolic &&
        pfv::env_int("PFV_PIPE", 1) != 0) {
      // ---- node || face pipeline (patterns in place: kept after a proved-equal topology, or a values-only call).  The
      // interaction-region kernel is bound by instruction issue, the face kernel by memory traffic; back to back each
      // leaves the other's resource idle.  The node kernel goes to the second stream in K runs of its largest size class
      // (topology.inc); after run q an event releases, on the main stream, the face kernel over the faces whose nodes are
      // all through -- one contiguous range of the ready-run-major face order.  The host enqueues node run, event, face
      // range alternately; the device overlaps face range q with node run q + 1.  Same kernels on the same data: the
      // matrices are bit for bit those of the sequential order (tests: PFV_PIPE=0 against 1).
      // Interaction regions the unpivoted elimination hands to the redo list are only known once every run is through:
      // they are redone then, and the faces around them recomputed (a handful per million nodes).
      tm.start(s);
      const int K = h->pipe_chunks;
      std::vector<hipEvent_t> ev((size_t)K + 1, nullptr);
      for (auto& e : ev) PFV_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      pfv::stream_t ns = h->aux_stream;
      try {
        pfv::StreamFork fork(s, ns);
        const std::function<void(int)> after = [&](int q) {
          PFV_HIP_CHECK(hipEventRecord(ev[q], ns));
          PFV_HIP_CHECK(hipStreamWaitEvent(s, ev[q], 0));
          const int64_t f0 = h->pipe_face_begin[q], f1 = h->pipe_face_begin[q + 1];
          pfv::run_face_kernel(*h, with_vs, nullptr, 0, f0, f1 - f0);
        };
        pfv::launch_node_kernel(*h, nullptr, ns, &after);
        fork.join();
      } catch (...) {
        for (auto& e : ev) if (e) (void)hipEventDestroy(e);
        throw;
      }
      for (auto& e : ev) (void)hipEventDestroy(e);
      pfv::check_node_status(*h, s);  // (reads the flags: everything above is through; runs the redo list, if any)
      if (h->stats.node_redo > 0) pfv::redo_faces_around_nodes(*h, with_vs, h->stats.node_redo);
      h->stats.node_ms = tm.stop(s);
      h->stats.pipeline_runs = K;
      node_done = face_done = true;
    }
#endif
    if (!node_done) {
      tm.start(s);
      pfv::run_node_kernel(*h);
      h->stats.node_ms = tm.stop(s);
    }
    tm.start(s);
    if (h->subface_bc) {
      if (!h->have_sub_symbolic || (flags & PFV_DISCR_REBUILD_TOPOLOGY)) pfv::build_subface_symbolic(*h);
      pfv::run_subface_kernel(*h);
      if (with_vs) pfv::run_face_kernel(*h, true);
    } else {
#ifndef PFV_EMULATE
      // The SpMV window of A (spmv_win.inc) only needs A's pattern, which the symbolic phase has left:
      // when the solve is going to work on the system in place (grid numbered along the Morton curve),
      // it is built now on the second stream, beside the face kernel, instead of inside pfv_solve.
      h->win_sys_prebuilt = h->win_rows_prebuilt = false;
      // (sizes of A's pattern: known now, or -- pattern deferred -- an upper estimate that only decides whether the
      // second stream is used at all; the exact test follows once the pattern exists)
      const int64_t nnzA_guess = cells_deferred ? h->pat_flux.nnz : h->pat_A.nnz;
      bool prebuild = h->aux_stream && h->have_cell_order && h->cell_order_identity &&
                      pfv::env_int("PFV_OVERLAP_WINDOW", 1) != 0 && pfv::env_int("PFV_REORDER", 1) != 0 &&
                      nnzA_guess >= pfv::env_int("PFV_SPMV_WINDOW_MIN_NNZ", 20000);
      // (a sharded solve multiplies the rows of the owned cells: the window of those rows, for the
      // number of owned rows of the previous solve on this handle)
      h->win_rows_prebuilt = false;
      bool prebuild_rows = !prebuild && h->aux_stream && h->win_rows_n > 0 && h->win_rows_n <= h->nc &&
                           pfv::env_int("PFV_OVERLAP_WINDOW", 1) != 0 &&
                           nnzA_guess >= pfv::env_int("PFV_SPMV_WINDOW_MIN_NNZ", 20000);
      if (prebuild || prebuild_rows || cells_deferred) {
        pfv::StreamFork fork(s, h->aux_stream);
        if (!face_done) pfv::run_face_kernel(*h, with_vs);  // (face_done: it ran inside the node || face pipeline)
        h->stream = h->aux_stream;
        try {
          if (cells_deferred) {
            pfv::build_symbolic(*h, 2);
            cells_deferred = false;
            const bool big = h->pat_A.nnz >= pfv::env_int("PFV_SPMV_WINDOW_MIN_NNZ", 20000);
            prebuild = prebuild && big;
            prebuild_rows = prebuild_rows && big;
          }
          if (!prebuild && !prebuild_rows) {
          } else if (prebuild) {
            // the windows are a function of A's pattern alone: kept when the symbolic phase has just proved the
            // pattern equal to the one they were built for (sizes + checksum of the index arrays)
            const bool keep = h->win_sys.ok && h->pat_A_checksum != 0 && h->win_sys_checksum == h->pat_A_checksum &&
                              h->win_sys.nrows == h->pat_A.nrows && h->win_sys.nnz == h->pat_A.nnz &&
                              pfv::env_int("PFV_WIN_REUSE", 1) != 0;
            h->stats.win_reused = keep ? 1 : 0;
            if (!keep) {
              h->win_sys_checksum = 0;
              pfv::win_build(*h, h->pat_A, h->win_sys);
              if (h->win_sys.ok) h->win_sys_checksum = h->pat_A_checksum;
            }
          } else {
            // (the window of the owned rows of a sharded solve: kept like win_sys when A's pattern is proved equal)
            const bool keep = h->win_rows.ok && h->pat_A_checksum != 0 && h->win_rows_checksum == h->pat_A_checksum &&
                              h->win_rows.nrows == h->win_rows_n && pfv::env_int("PFV_WIN_REUSE", 1) != 0;
            h->stats.win_reused = keep ? 1 : 0;
            if (!keep) {
              h->win_rows_checksum = 0;
              RowsView rows(h->pat_A, h->win_rows_n);
              pfv::win_build(*h, rows.V, h->win_rows);
              if (h->win_rows.ok) h->win_rows_checksum = h->pat_A_checksum;
            }
          }
        } catch (...) {
          h->stream = s;
          throw;
        }
        h->stream = s;
        fork.join();
        if (prebuild) {
          h->win_for = h->pat_A.indices.p;
          h->win_sys_prebuilt = true;
        } else if (prebuild_rows) {
          h->win_rows_for = h->pat_A.indices.p;
          h->win_rows_prebuilt = true;
        }
      } else
#endif
      if (!face_done) pfv::run_face_kernel(*h, with_vs);
    }
    if (cells_deferred) pfv::build_symbolic(*h, 2);  // (no second stream was used)
    h->stats.face_ms = tm.stop(s);
    h->stats.discretize_ms = tall.stop(s);
    h->have_numeric = true;
    h->rows_complete = !h->subface_bc;
    h->have_system = false;
    h->filled[PFV_MAT_SYSTEM] = false;
    double bytes = 0.0;
    bytes += 2.0 * 8.0 * (double)h->pat_flux.nnz + 2.0 * 8.0 * (double)h->pat_bound.nnz;
    if (with_vs) bytes += 2.0 * 8.0 * (double)h->pat_vs.nnz;
    h->stats.bytes_written_outputs = bytes;
  });
}

pfv_status pfv_tpfa_discretize(pfv_ctx* h, int vector_source_dim) {
  return guarded(h, [&] {
    require(h->have_grid && h->have_params, "grid and parameters must be set before discretize");
    require(vector_source_dim >= 1 && vector_source_dim <= 3, "vector_source_dim must be 1, 2 or 3");
    auto s = h->stream;
    pfv::Timer tm;
    tm.start(s);
    h->have_symbolic = false;  // the MPFA patterns (if any) are replaced
    h->symb_key = 0;
    h->rows_complete = false;
    for (int m = PFV_MAT_FLUX; m <= PFV_MAT_BOUND_PRESSURE_VECTOR_SOURCE; ++m) h->filled[m] = false;
    h->filled[PFV_MAT_SYSTEM] = false;
    pfv::tpfa_discretize(*h, vector_source_dim);
    h->stats.face_ms = tm.stop(s);
    h->tpfa_mode = true;
    h->have_symbolic = true;
    h->have_topology = false;  // an MPFA call on this handle rebuilds its own topology + patterns
    h->have_numeric = true;
    h->have_system = false;
    h->win_sys_prebuilt = h->win_rows_prebuilt = false;
  });
}

pfv_status pfv_tpfa_transmissibility_ad(pfv_ctx* h, const double* perm_33n, double* t_face, double* dt_dk) {
  return guarded(h, [&] {
    require(h->have_grid, "the grid must be set first");
    require(perm_33n && t_face && dt_dk, "null argument");
    auto s = h->stream;
    const size_t nc = (size_t)h->nc, nf = (size_t)h->nf, ncf = (size_t)h->ncf;
    pfv::Buf<double> perm, t, jac;
    double* dp = perm.ensure(9 * nc);
    vec_in(h, dp, perm_33n, 9 * nc);
    double* dt = h->vectors_on_device ? t_face : t.ensure(std::max<size_t>(nf, 1));
    double* dj = h->vectors_on_device ? dt_dk : jac.ensure(std::max<size_t>(9 * ncf, 1));
    pfv::tpfa_transmissibility_ad(*h, dp, dt, dj);
    if (!h->vectors_on_device) {
      be_d2h(t_face, dt, sizeof(double) * nf, s);
      be_d2h(dt_dk, dj, sizeof(double) * 9 * ncf, s);
    } else {
      pfv::be_sync(s);
    }
  });
}

pfv_status pfv_mpfa_discretize_faces(pfv_ctx* h, uint32_t flags, int64_t n_faces, const int32_t* faces,
                                     int keep_other_rows) {
  return guarded(h, [&] {
    require(h->have_grid && h->have_params, "grid and parameters must be set before discretize");
    require(n_faces >= 0 && (n_faces == 0 || faces), "bad face list");
    require(h->nd >= 2, "MPFA needs a 2-D or 3-D grid");
    require(!h->subface_bc, "partial discretization with conditions per sub-face is not covered");
    require(!keep_other_rows || h->rows_complete,
            "update of a discretization that was never computed on this handle");
    for (int64_t i = 0; i < n_faces; ++i)
      require(faces[i] >= 0 && faces[i] < h->nf, "face index out of range");
    auto s = h->stream;
    pfv::Timer tm;
    if (!h->have_topology || !h->have_symbolic || (flags & PFV_DISCR_REBUILD_TOPOLOGY)) {
      require(!keep_other_rows, "the topology cannot be rebuilt under an update");
      h->tpfa_mode = false;
      tm.start(s);
      pfv::build_topology(*h);
      h->stats.topology_ms = tm.stop(s);
      tm.start(s);
      pfv::build_symbolic(*h);
      h->stats.symbolic_ms = tm.stop(s);
    }
    const bool with_vs = !(flags & PFV_DISCR_SKIP_VECTOR_SOURCE);
    int32_t* sub = h->face_subset.ensure(std::max<int64_t>(n_faces, 1));
    uint8_t* act = h->node_active.ensure(h->nn);
    pfv::be_h2d(sub, faces, sizeof(int32_t) * (size_t)n_faces, s);
    pfv::be_memset(act, 0, (size_t)h->nn, s);
    const int32_t* fn_ptr = h->fn_ptr;
    const int32_t* fn_idx = h->fn_idx;
    pfv::parallel_for(s, n_faces, PFV_LAMBDA(int64_t i) {
      const int f = sub[i];
      for (int e = fn_ptr[f]; e < fn_ptr[f + 1]; ++e) act[fn_idx[e]] = 1;
    });
    tm.start(s);
    pfv::run_node_kernel(*h, act);
    h->stats.node_ms = tm.stop(s);
    if (!keep_other_rows) {
      h->rows_complete = false;
      const int mats[6] = {PFV_MAT_FLUX, PFV_MAT_BOUND_FLUX, PFV_MAT_BOUND_PRESSURE_CELL,
                           PFV_MAT_BOUND_PRESSURE_FACE, PFV_MAT_VECTOR_SOURCE,
                           PFV_MAT_BOUND_PRESSURE_VECTOR_SOURCE};
      for (int m : mats) {
        if (m >= PFV_MAT_VECTOR_SOURCE && !with_vs) continue;
        const int64_t nnz = h->pattern_of(m).nnz;
        double* v = h->val[m].ensure(std::max<int64_t>(nnz, 1));
        pfv::be_memset(v, 0, sizeof(double) * (size_t)nnz, s);
      }
    }
    tm.start(s);
    pfv::run_face_kernel(*h, with_vs, sub, n_faces);
    h->stats.face_ms = tm.stop(s);
    h->have_numeric = true;
    h->have_system = false;
    h->win_sys_prebuilt = h->win_rows_prebuilt = false;
    h->filled[PFV_MAT_SYSTEM] = false;
  });
}

// index arrays that are written on first use (topology.inc: ensure_vs_indices)
static void materialize_pattern(pfv_ctx* h, int which) {
  if (which == PFV_MAT_VECTOR_SOURCE || which == PFV_MAT_BOUND_PRESSURE_VECTOR_SOURCE) pfv::ensure_vs_indices(*h);
}

pfv_status pfv_matrix_info(pfv_ctx* h, int which, int64_t* nrows, int64_t* ncols, int64_t* nnz) {
  return guarded(h, [&] {
    require(which >= 0 && which < PFV_NUM_MATS, "bad matrix selector");
    require(pattern_ready(h, which), "discretize first");
    const pfv::CsrPattern& P = h->pattern_of(which);
    if (nrows) *nrows = P.nrows;
    if (ncols) *ncols = P.ncols;
    if (nnz) *nnz = P.nnz;
  });
}

pfv_status pfv_get_matrix(pfv_ctx* h, int which, int32_t* indptr, int32_t* indices, double* data) {
  return guarded(h, [&] {
    require(which >= 0 && which < PFV_NUM_MATS, "bad matrix selector");
    require(pattern_ready(h, which), "discretize first");
    if ((which == PFV_MAT_VECTOR_SOURCE || which == PFV_MAT_BOUND_PRESSURE_VECTOR_SOURCE) && h->vs_implicit && !h->tpfa_mode)
      throw pfv::Error(PFV_ERR_UNSUPPORTED, "vector_source has more than 2^31 entries on this handle: int32 CSR arrays "
                                            "cannot hold it -- fetch it by rows (pfv_get_matrix_rows)");
    if (indices) materialize_pattern(h, which);
    const pfv::CsrPattern& P = h->pattern_of(which);
    auto s = h->stream;
    if (indptr) be_d2h(indptr, P.indptr.p, sizeof(int32_t) * (size_t)(P.nrows + 1), s);
    if (indices) be_d2h(indices, P.indices.p, sizeof(int32_t) * (size_t)P.nnz, s);
    if (data) {
      require(h->filled[which], "matrix values have not been computed");
      be_d2h(data, h->val[which].p, sizeof(double) * (size_t)P.nnz, s);
    }
  });
}

pfv_status pfv_get_matrix_rows(pfv_ctx* h, int which, int64_t n_rows, const int32_t* rows, int32_t* out_indptr,
                               int32_t* out_indices, double* out_data) {
  return guarded(h, [&] {
    require(which >= 0 && which < PFV_NUM_MATS, "bad matrix selector");
    require(pattern_ready(h, which), "discretize first");
    require(n_rows >= 0 && (n_rows == 0 || rows) && out_indptr, "bad row list");
    // (vector-source matrices beyond 2^31 entries: rows are expanded from the flux pattern, nd entries per flux entry)
    const bool imp = (which == PFV_MAT_VECTOR_SOURCE || which == PFV_MAT_BOUND_PRESSURE_VECTOR_SOURCE) && h->vs_implicit &&
                     !h->tpfa_mode;
    const int mult = imp ? h->nd : 1;
    if (!imp && (out_indices || out_data)) materialize_pattern(h, which);
    const pfv::CsrPattern& P = imp ? h->pat_flux : h->pattern_of(which);
    for (int64_t i = 0; i < n_rows; ++i) require(rows[i] >= 0 && rows[i] < P.nrows, "row index out of range");
    auto s = h->stream;
    pfv::Buf<int32_t> d_rows, d_len, d_ix;
    pfv::Buf<int64_t> d_ptr;
    pfv::Buf<double> d_val;
    int32_t* dr = d_rows.ensure(std::max<int64_t>(n_rows, 1));
    int32_t* dl = d_len.ensure(n_rows + 1);
    int64_t* dp = d_ptr.ensure(n_rows + 1);
    be_h2d(dr, rows, sizeof(int32_t) * (size_t)n_rows, s);
    const int32_t* ip = P.indptr;
    pfv::parallel_for(s, n_rows, PFV_LAMBDA(int64_t i) { dl[i] = (ip[dr[i] + 1] - ip[dr[i]]) * mult; });
    pfv::exclusive_scan<int32_t, int64_t>(s, h->scratch, dl, dp, (size_t)n_rows);
    std::vector<int64_t> hp((size_t)n_rows + 1);
    be_d2h(hp.data(), dp, sizeof(int64_t) * (size_t)(n_rows + 1), s);
    require(hp[(size_t)n_rows] < (int64_t(1) << 31), "more than 2^31 entries requested");
    for (int64_t i = 0; i <= n_rows; ++i) out_indptr[i] = (int32_t)hp[(size_t)i];
    if (!out_indices && !out_data) return;
    require(!out_data || h->filled[which], "matrix values have not been computed");
    const int64_t tot = hp[(size_t)n_rows];
    const int32_t* ix = P.indices;
    const double* val = out_data ? h->val[which].p : nullptr;
    int32_t* tix = d_ix.ensure(std::max<int64_t>(tot, 1));
    double* tv = d_val.ensure(std::max<int64_t>(out_data ? tot : 1, 1));
    pfv::wave_for<16>(s, n_rows, 0, PFV_LAMBDA(const pfv::WaveCtx& w) {
      const int64_t i = w.item;
      const int p0 = ip[dr[i]], len = dl[i];
      const int64_t o = dp[i];
      if (mult == 1) {
        PFV_LANES(k, len) {
          tix[o + k] = ix[p0 + k];
          if (val) tv[o + k] = val[p0 + k];
        }
      } else {
        PFV_LANES(k, len) {
          const int e = k / mult, cpt = k - e * mult;
          tix[o + k] = ix[p0 + e] * mult + cpt;
          if (val) tv[o + k] = val[(int64_t)p0 * mult + k];
        }
      }
    });
    if (out_indices) be_d2h(out_indices, tix, sizeof(int32_t) * (size_t)tot, s);
    if (out_data) be_d2h(out_data, tv, sizeof(double) * (size_t)tot, s);
  });
}

pfv_status pfv_mpfa_assemble(pfv_ctx* h, const double* bc_values, const double* vector_source,
                             const double* source) {
  return guarded(h, [&] {
    require(h->have_numeric, "discretize first");
    require(!h->subface_bc, "flux has sub-face rows (conditions per sub-face): collapse it before assembling");
    require(bc_values != nullptr, "bc_values is required");
    require(!vector_source || h->filled[PFV_MAT_VECTOR_SOURCE], "vector_source matrix was skipped");
    auto s = h->stream;
    pfv::Timer tm;
    const size_t nf = (size_t)h->nf, nc = (size_t)h->nc;
    const size_t nvs = (size_t)h->pat_vs.ncols;  // nd (MPFA) or ambient-dimension (TPFA) entries per cell
    double* in = h->vec_in.ensure(nf + nvs + nc);
    double* d_bc = in;
    double* d_vs = vector_source ? in + nf : nullptr;
    double* d_src = source ? in + nf + nvs : nullptr;
    vec_in(h, d_bc, bc_values, nf);
    if (d_vs) vec_in(h, d_vs, vector_source, nvs);
    if (d_src) vec_in(h, d_src, source, nc);
    tm.start(s);
    if (!h->have_system) {
      pfv::assemble_system(*h);
      if (h->amg) h->amg->valid = false;
      if (h->block_pc) h->block_pc->for_val = nullptr;
      if (h->amg_block) h->amg_block->valid = false;
      h->perm_for_val = nullptr;
      // windows built for this very pattern by the discretize call are kept
      if (h->win_rows_prebuilt) h->win_rows_prebuilt = false;
      else h->win_rows_for = nullptr;
      if (h->win_sys_prebuilt) h->win_sys_prebuilt = false;
      else h->win_for = nullptr;
    }
    pfv::assemble_rhs(*h, d_bc, d_vs, d_src);
    h->stats.assemble_ms = tm.stop(s);
    h->have_system = true;
    h->active.P = &h->pat_A;
    h->active.val = h->val[PFV_MAT_SYSTEM].p;
    h->active.diag = h->diag.p;
    h->active.rhs = h->rhs.p;
    h->active.n = h->nc;
    h->active_bs = 1;
    h->active_is_grid = true;
    h->active.valid = true;
  });
}

pfv_status pfv_mpfa_ad_flux_system(pfv_ctx* h, const double* p, const double* dk_dp, const double* bc_values,
                                   const double* vector_source, const double* source, double* flux_out,
                                   uint32_t flags) {
  return guarded(h, [&] {
    require(h->have_numeric && !h->tpfa_mode, "pfv_mpfa_discretize first");
    require(!h->subface_bc, "conditions per sub-face are not covered");
    require(p != nullptr, "p is required");
    require(!vector_source || h->filled[PFV_MAT_VECTOR_SOURCE], "vector_source matrix was skipped");
    auto s = h->stream;
    const size_t nf = (size_t)h->nf, nc = (size_t)h->nc, nvs = (size_t)h->pat_vs.ncols;
    pfv::Buf<double> in_;
    double* in = in_.ensure(nc + 9 * nc + nf + nvs + nc + nf);
    double* d_p = in;
    double* d_dk = dk_dp ? in + nc : nullptr;
    double* d_bc = bc_values ? in + 10 * nc : nullptr;
    double* d_vs = vector_source ? in + 10 * nc + nf : nullptr;
    double* d_src = source ? in + 10 * nc + nf + nvs : nullptr;
    double* d_q = in + 11 * nc + nf + nvs;
    vec_in(h, d_p, p, nc);
    if (d_dk) vec_in(h, d_dk, dk_dp, 9 * nc);
    if (d_bc) vec_in(h, d_bc, bc_values, nf);
    if (d_vs) vec_in(h, d_vs, vector_source, nvs);
    if (d_src) vec_in(h, d_src, source, nc);
    pfv::Timer tm;
    tm.start(s);
    pfv::ad_flux_system(*h, d_p, d_dk, d_bc, d_vs, d_src, flux_out ? d_q : nullptr,
                        (flags & PFV_AD_WANT_FLUX_JACOBIAN) != 0);
    h->stats.assemble_ms = tm.stop(s);
    if (flux_out) {
      if (h->vectors_on_device) pfv::be_d2d(flux_out, d_q, nf * sizeof(double), s);
      else be_d2h(flux_out, d_q, nf * sizeof(double), s);
    }
    pfv::be_sync(s);
    h->have_system = false;  // PFV_MAT_SYSTEM now holds J, not div flux: pfv_mpfa_assemble rebuilds it
    if (h->amg) h->amg->valid = false;
      if (h->block_pc) h->block_pc->for_val = nullptr;
    if (h->amg_block) h->amg_block->valid = false;
    h->perm_for_val = nullptr;
    h->win_for = h->win_rows_for = nullptr;
    h->win_sys_prebuilt = h->win_rows_prebuilt = false;
    h->active.P = &h->pat_A;
    h->active.val = h->val[PFV_MAT_SYSTEM].p;
    h->active.diag = h->diag.p;
    h->active.rhs = h->rhs.p;
    h->active.n = h->nc;
    h->active_bs = 1;
    h->active_is_grid = true;
    h->active.valid = true;
  });
}

pfv_status pfv_mpsa_set_params(pfv_ctx* h, const double* stiffness_99n, const double* cell_volumes,
                               const uint8_t* bc_dir_bits, const uint8_t* bc_neu_bits, double eta) {
  return guarded(h, [&] {
    require(h->have_grid, "pfv_set_grid must be called first");
    require(stiffness_99n && cell_volumes && bc_dir_bits && bc_neu_bits, "null parameter array");
    auto s = h->stream;
    upload(h->stiff, stiffness_99n, 81 * (size_t)h->nc, s);
    upload(h->cvol, cell_volumes, (size_t)h->nc, s);
    upload(h->bc_dirbits, bc_dir_bits, (size_t)h->nf, s);
    upload(h->bc_neubits, bc_neu_bits, (size_t)h->nf, s);
    h->have_mpsa_robin = false;
    h->have_mpsa_eta_sub = false;
    h->mpsa_hf_on = false;
    h->have_mpsa_hf_eta_sub = false;
    h->have_mpsa_basis = false;
    h->have_mpsa_basis_sub = false;
    h->mpsa_subface_bc = false;
    h->mpsa_eta = eta;
    h->have_mpsa_params = true;
    h->have_mpsa_numeric = h->have_mech_system = false;
  });
}

pfv_status pfv_mpsa_set_robin(pfv_ctx* h, const uint8_t* bc_rob_bits, const double* robin_weight_ddn) {
  return guarded(h, [&] {
    require(h->have_grid && h->have_mpsa_params, "pfv_mpsa_set_params first");
    auto s = h->stream;
    h->have_mpsa_robin = false;
    if (!bc_rob_bits) return;
    const size_t nf = (size_t)h->nf, n2 = (size_t)h->nd * h->nd;
    upload(h->bc_robbits, bc_rob_bits, nf, s);
    if (robin_weight_ddn) {
      upload(h->mpsa_robw, robin_weight_ddn, n2 * nf, s);
    } else {
      std::vector<double> eye(n2 * nf, 0.0);
      for (int i = 0; i < h->nd; ++i)
        for (size_t f = 0; f < nf; ++f) eye[((size_t)h->nd * i + i) * nf + f] = 1.0;
      upload(h->mpsa_robw, eye.data(), n2 * nf, s);
    }
    h->have_mpsa_robin = true;
    h->have_mpsa_numeric = false;
    h->have_mech_system = false;
  });
}

pfv_status pfv_mpsa_set_subface_eta(pfv_ctx* h, const double* eta_subface) {
  return guarded(h, [&] {
    require(h->have_grid && h->have_mpsa_params, "pfv_mpsa_set_params first");
    h->have_mpsa_eta_sub = false;
    if (eta_subface) {
      upload(h->mpsa_eta_sub, eta_subface, (size_t)h->nsf, h->stream);
      h->have_mpsa_eta_sub = true;
    }
    h->have_mpsa_numeric = false;
    h->have_mech_system = false;
  });
}

pfv_status pfv_mpsa_set_reconstruction_eta(pfv_ctx* h, int on, double hf_eta) {
  return guarded(h, [&] {
    require(h->have_grid && h->have_mpsa_params, "pfv_mpsa_set_params first");
    require(!on || (hf_eta >= 0.0 && hf_eta < 1.0), "reconstruction_eta must lie in [0, 1)");
    h->mpsa_hf_on = on != 0;
    h->mpsa_hf_eta = hf_eta;
    h->have_mpsa_hf_eta_sub = false;
    h->have_mpsa_numeric = false;
    h->have_mech_system = false;
  });
}

pfv_status pfv_mpsa_set_reconstruction_eta_subface(pfv_ctx* h, const double* hf_eta_subface) {
  return guarded(h, [&] {
    require(h->have_grid && h->have_mpsa_params, "pfv_mpsa_set_params first");
    h->have_mpsa_hf_eta_sub = false;
    h->mpsa_hf_on = hf_eta_subface != nullptr;
    if (hf_eta_subface) {
      if (!h->have_topology) {  // the sub-face count is known from the topology
        pfv::build_topology(*h);
        pfv::build_symbolic(*h);
        h->tpfa_mode = false;
        h->have_mpsa_symbolic = false;
        h->have_sub_symbolic = h->have_mpsa_sub_symbolic = false;
        h->have_numeric = h->have_system = false;
      }
      upload(h->mpsa_hf_eta_sub, hf_eta_subface, (size_t)h->nsf, h->stream);
      h->have_mpsa_hf_eta_sub = true;
    }
    h->have_mpsa_numeric = false;
    h->have_mech_system = false;
  });
}

pfv_status pfv_mpsa_set_basis(pfv_ctx* h, const double* basis_ddn) {
  return guarded(h, [&] {
    require(h->have_grid && h->have_mpsa_params, "pfv_mpsa_set_params first");
    h->have_mpsa_basis = false;
    if (basis_ddn) {
      upload(h->mpsa_basis, basis_ddn, (size_t)h->nd * h->nd * (size_t)h->nf, h->stream);
      h->have_mpsa_basis = true;
    }
    h->have_mpsa_numeric = false;
    h->have_mech_system = false;
  });
}

pfv_status pfv_mpsa_set_subface_bc(pfv_ctx* h, const uint8_t* bc_dir_bits_sub, const uint8_t* bc_neu_bits_sub,
                                   const uint8_t* bc_rob_bits_sub, const double* robin_weight_dds) {
  return guarded(h, [&] {
    require(h->have_grid && h->have_mpsa_params, "pfv_mpsa_set_params first");
    h->mpsa_subface_bc = false;
    h->have_mpsa_basis_sub = false;
    h->have_mpsa_numeric = h->have_mech_system = false;
    if (!bc_dir_bits_sub) return;  // back to conditions per face
    require(bc_neu_bits_sub != nullptr, "null parameter array");
    require(!h->have_mpsa_basis, "conditions per sub-face take their basis per sub-face: pfv_mpsa_set_subface_basis");
    require(h->biot_nalpha == 0, "conditions per sub-face with Biot coupling terms are not covered");
    auto s = h->stream;
    if (!h->have_topology) {  // the sub-face count is known from the topology
      pfv::build_topology(*h);
      pfv::build_symbolic(*h);
      h->tpfa_mode = false;
      h->have_mpsa_symbolic = false;
      h->have_sub_symbolic = h->have_mpsa_sub_symbolic = false;
      h->have_numeric = h->have_system = false;
    }
    const size_t nsf = (size_t)h->nsf, n2 = (size_t)h->nd * h->nd;
    upload(h->bc_dirbits_sub, bc_dir_bits_sub, nsf, s);
    upload(h->bc_neubits_sub, bc_neu_bits_sub, nsf, s);
    h->have_mpsa_robin_sub = bc_rob_bits_sub != nullptr;
    if (bc_rob_bits_sub) {
      upload(h->bc_robbits_sub, bc_rob_bits_sub, nsf, s);
      if (robin_weight_dds) {
        upload(h->mpsa_robw_sub, robin_weight_dds, n2 * nsf, s);
      } else {
        std::vector<double> eye(n2 * nsf, 0.0);
        for (int i = 0; i < h->nd; ++i)
          for (size_t f = 0; f < nsf; ++f) eye[((size_t)h->nd * i + i) * nsf + f] = 1.0;
        upload(h->mpsa_robw_sub, eye.data(), n2 * nsf, s);
      }
    }
    h->mpsa_subface_bc = true;
  });
}

pfv_status pfv_mpsa_set_subface_basis(pfv_ctx* h, const double* basis_dds) {
  return guarded(h, [&] {
    require(h->have_grid && h->have_mpsa_params && h->mpsa_subface_bc, "pfv_mpsa_set_subface_bc first");
    h->have_mpsa_basis_sub = false;
    if (basis_dds) {
      upload(h->mpsa_basis_sub, basis_dds, (size_t)h->nd * h->nd * (size_t)h->nsf, h->stream);
      h->have_mpsa_basis_sub = true;
    }
    h->have_mpsa_numeric = false;
    h->have_mech_system = false;
  });
}

pfv_status pfv_mpsa_discretize(pfv_ctx* h, uint32_t flags) {
  return guarded(h, [&] {
    require(h->have_grid && h->have_mpsa_params, "grid and MPSA parameters must be set before discretize");
    auto s = h->stream;
    pfv::Timer tm;
    if (!h->have_topology || !h->have_symbolic || (flags & PFV_DISCR_REBUILD_TOPOLOGY)) {
      tm.start(s);
      pfv::build_topology(*h);
      h->stats.topology_ms = tm.stop(s);
      tm.start(s);
      // (a rebuilt topology that is proved equal to the one the patterns were built from keeps them -- and with them
      // their block expansions of mpsa_symbolic, functions of those patterns alone; see symbolic_outputs_still_valid)
      const bool had_mpsa_symbolic = h->have_mpsa_symbolic;
      if (symbolic_outputs_still_valid(h)) {
        h->stats.symbolic_ms = tm.stop(s);
        h->have_mpsa_symbolic = had_mpsa_symbolic;
        h->have_numeric = h->have_system = false;
      } else {
      pfv::build_symbolic(*h);
      h->stats.symbolic_ms = tm.stop(s);
      h->tpfa_mode = false;
      h->have_mpsa_symbolic = false;
      h->have_sub_symbolic = h->have_mpsa_sub_symbolic = false;
      h->have_numeric = h->have_system = false;
      }
    }
    if (!h->have_mpsa_symbolic) {
      tm.start(s);
      pfv::mpsa_symbolic(*h);
      h->stats.symbolic_ms += tm.stop(s);
      h->have_biot_symbolic = false;
    }
    if (h->biot_nalpha > 0 && !h->have_biot_symbolic) pfv::biot_symbolic(*h);
    if (h->mpsa_subface_bc && !h->have_mpsa_sub_symbolic) {
      require(h->biot_nalpha == 0, "conditions per sub-face with Biot coupling terms are not covered");
      tm.start(s);
      pfv::mpsa_subface_symbolic(*h);
      h->stats.symbolic_ms += tm.stop(s);
    }
    tm.start(s);
    pfv::mpsa_run_node_kernel(*h);
    h->stats.node_ms = tm.stop(s);
    tm.start(s);
    pfv::mpsa_run_face_kernel(*h);
    if (h->mpsa_subface_bc) pfv::mpsa_run_subface_kernel(*h);
    h->stats.face_ms = tm.stop(s);
    h->have_mpsa_numeric = true;
    h->rows_complete_m = true;
    h->have_mech_system = false;
    h->filled[PFV_MAT_MECH_SYSTEM] = false;
  });
}

pfv_status pfv_mpsa_discretize_faces(pfv_ctx* h, uint32_t flags, int64_t n_faces, const int32_t* faces,
                                     int keep_other_rows) {
  return guarded(h, [&] {
    require(h->have_grid && h->have_mpsa_params, "grid and MPSA parameters must be set before discretize");
    require(n_faces >= 0 && (n_faces == 0 || faces), "bad face list");
    require(!keep_other_rows || h->rows_complete_m,
            "update of a discretization that was never computed on this handle");
    require(!h->mpsa_subface_bc, "partial discretization with conditions per sub-face is not covered");
    for (int64_t i = 0; i < n_faces; ++i)
      require(faces[i] >= 0 && faces[i] < h->nf, "face index out of range");
    auto s = h->stream;
    pfv::Timer tm;
    if (!h->have_topology || !h->have_symbolic || (flags & PFV_DISCR_REBUILD_TOPOLOGY)) {
      require(!keep_other_rows, "the topology cannot be rebuilt under an update");
      pfv::build_topology(*h);
      pfv::build_symbolic(*h);
      h->tpfa_mode = false;
      h->have_mpsa_symbolic = false;
      h->have_sub_symbolic = h->have_mpsa_sub_symbolic = false;
      h->have_numeric = h->have_system = false;
    }
    if (!h->have_mpsa_symbolic) {
      pfv::mpsa_symbolic(*h);
      h->have_biot_symbolic = false;
    }
    if (h->biot_nalpha > 0 && !h->have_biot_symbolic) pfv::biot_symbolic(*h);
    int32_t* sub = h->face_subset.ensure(std::max<int64_t>(n_faces, 1));
    uint8_t* act = h->node_active.ensure(h->nn);
    pfv::be_h2d(sub, faces, sizeof(int32_t) * (size_t)n_faces, s);
    pfv::be_memset(act, 0, (size_t)h->nn, s);
    const int32_t* fn_ptr = h->fn_ptr;
    const int32_t* fn_idx = h->fn_idx;
    pfv::parallel_for(s, n_faces, PFV_LAMBDA(int64_t i) {
      const int f = sub[i];
      for (int e = fn_ptr[f]; e < fn_ptr[f + 1]; ++e) act[fn_idx[e]] = 1;
    });
    tm.start(s);
    pfv::mpsa_run_node_kernel(*h, act);
    h->stats.node_ms = tm.stop(s);
    if (!keep_other_rows) {
      h->rows_complete_m = false;
      for (int m = PFV_MAT_STRESS; m <= PFV_MAT_BOUND_DISPLACEMENT_FACE; ++m) {
        const int64_t nnz = h->pattern_of(m).nnz;
        double* v = h->val[m].ensure(std::max<int64_t>(nnz, 1));
        pfv::be_memset(v, 0, sizeof(double) * (size_t)nnz, s);
      }
    }
    tm.start(s);
    pfv::mpsa_run_face_kernel(*h, sub, n_faces);
    h->stats.face_ms = tm.stop(s);
    h->have_mpsa_numeric = true;
    h->have_mech_system = false;
    h->filled[PFV_MAT_MECH_SYSTEM] = false;
  });
}

pfv_status pfv_biot_set_alphas(pfv_ctx* h, int nalpha, const double* alpha_k33n) {
  return guarded(h, [&] {
    require(h->have_grid, "pfv_set_grid must be called first");
    require(nalpha >= 0 && nalpha <= 8 && (nalpha == 0 || alpha_k33n), "bad coupling tensors");
    const bool layout_change = (nalpha > 0) != (h->biot_nalpha > 0) || nalpha != h->biot_nalpha;
    h->biot_nalpha = nalpha;
    if (nalpha > 0) upload(h->biot_alpha, alpha_k33n, (size_t)nalpha * 9 * (size_t)h->nc, h->stream);
    if (layout_change) {
      h->have_mpsa_symbolic = false;  // LDS sizing of the node kernel depends on it
      h->have_biot_symbolic = false;
      h->biot_rows_complete = false;
    }
    h->have_biot_numeric = false;
  });
}

pfv_status pfv_biot_discretize(pfv_ctx* h, uint32_t flags) {
  return guarded(h, [&] {
    require(h->have_grid && h->have_mpsa_params, "grid and MPSA parameters must be set before discretize");
    require(h->biot_nalpha > 0, "pfv_biot_set_alphas first");
    auto s = h->stream;
    pfv::Timer tm;
    if (!h->have_topology || !h->have_symbolic || (flags & PFV_DISCR_REBUILD_TOPOLOGY)) {
      tm.start(s);
      pfv::build_topology(*h);
      h->stats.topology_ms = tm.stop(s);
      tm.start(s);
      pfv::build_symbolic(*h);
      h->stats.symbolic_ms = tm.stop(s);
      h->tpfa_mode = false;
      h->have_mpsa_symbolic = h->have_biot_symbolic = false;
      h->have_numeric = h->have_system = false;
    }
    if (!h->have_mpsa_symbolic) {
      pfv::mpsa_symbolic(*h);
      h->have_biot_symbolic = false;
    }
    if (!h->have_biot_symbolic) pfv::biot_symbolic(*h);
    tm.start(s);
    pfv::mpsa_run_node_kernel(*h);
    h->stats.node_ms = tm.stop(s);
    tm.start(s);
    pfv::mpsa_run_face_kernel(*h);
    for (int ka = 0; ka < h->biot_nalpha; ++ka) {
      pfv::biot_run_face_kernel(*h, ka);
      pfv::biot_run_cell_kernel(*h, ka);
    }
    h->stats.face_ms = tm.stop(s);
    h->have_mpsa_numeric = true;
    h->rows_complete_m = true;
    h->have_biot_numeric = true;
    h->biot_rows_complete = true;
    h->have_mech_system = false;
    h->filled[PFV_MAT_MECH_SYSTEM] = false;
  });
}

pfv_status pfv_biot_discretize_faces(pfv_ctx* h, uint32_t flags, int64_t n_faces, const int32_t* faces,
                                     int64_t n_cells, const int32_t* cells, int keep_other_rows) {
  if (!h) return PFV_ERR_ARGUMENT;
  if (keep_other_rows && !h->biot_rows_complete) {
    h->err = "update of coupling terms that were never computed on this handle";
    return PFV_ERR_ARGUMENT;
  }
  if (h->biot_nalpha <= 0) {
    h->err = "pfv_biot_set_alphas first";
    return PFV_ERR_ARGUMENT;
  }
  // interaction regions of the faces' nodes (with the Biot tail) and the four MPSA matrices
  pfv_status st = pfv_mpsa_discretize_faces(h, flags, n_faces, faces, keep_other_rows);
  if (st != PFV_OK) return st;
  return guarded(h, [&] {
    require(n_cells >= 0 && (n_cells == 0 || cells), "bad cell list");
    for (int64_t i = 0; i < n_cells; ++i) require(cells[i] >= 0 && cells[i] < h->nc, "cell index out of range");
    auto s = h->stream;
    int32_t* csub = h->cell_subset.ensure(std::max<int64_t>(n_cells, 1));
    pfv::be_h2d(csub, cells, sizeof(int32_t) * (size_t)n_cells, s);
    if (!keep_other_rows) {
      for (int term = 0; term < PFV_BIOT_NUM_TERMS; ++term) {
        const int64_t nnz = pfv::biot_pattern(*h, term).nnz;
        for (int ka = 0; ka < h->biot_nalpha; ++ka) {
          double* v = h->biot_val[(size_t)term * h->biot_nalpha + ka].ensure(std::max<int64_t>(nnz, 1));
          pfv::be_memset(v, 0, sizeof(double) * (size_t)nnz, s);
        }
      }
    }
    for (int ka = 0; ka < h->biot_nalpha; ++ka) {
      if (n_faces > 0) pfv::biot_run_face_kernel(*h, ka, h->face_subset.p, n_faces);
      if (n_cells > 0) pfv::biot_run_cell_kernel(*h, ka, csub, n_cells);
    }
    if (!keep_other_rows) h->biot_rows_complete = false;
    h->have_biot_numeric = true;
  });
}

pfv_status pfv_biot_matrix_info(pfv_ctx* h, int term, int64_t* nrows, int64_t* ncols, int64_t* nnz) {
  return guarded(h, [&] {
    require(term >= 0 && term < PFV_BIOT_NUM_TERMS, "bad term");
    require(h->have_biot_symbolic, "pfv_biot_discretize first");
    const pfv::CsrPattern& P = pfv::biot_pattern(*h, term);
    if (nrows) *nrows = P.nrows;
    if (ncols) *ncols = P.ncols;
    if (nnz) *nnz = P.nnz;
  });
}

pfv_status pfv_biot_get_matrix(pfv_ctx* h, int term, int key, int32_t* indptr, int32_t* indices, double* data) {
  return guarded(h, [&] {
    require(term >= 0 && term < PFV_BIOT_NUM_TERMS, "bad term");
    require(h->have_biot_symbolic, "pfv_biot_discretize first");
    require(key >= 0 && key < h->biot_nalpha, "bad coupling key");
    const pfv::CsrPattern& P = pfv::biot_pattern(*h, term);
    auto s = h->stream;
    if (indptr) be_d2h(indptr, P.indptr.p, sizeof(int32_t) * (size_t)(P.nrows + 1), s);
    if (indices) be_d2h(indices, P.indices.p, sizeof(int32_t) * (size_t)P.nnz, s);
    if (data) {
      require(h->have_biot_numeric, "pfv_biot_discretize first");
      be_d2h(data, h->biot_val[(size_t)term * h->biot_nalpha + key].p, sizeof(double) * (size_t)P.nnz, s);
    }
  });
}

pfv_status pfv_mpsa_assemble(pfv_ctx* h, const double* bc_values, const double* source) {
  return guarded(h, [&] {
    require(h->have_mpsa_numeric, "pfv_mpsa_discretize first");
    require(!h->mpsa_subface_bc, "stress has sub-face rows (conditions per sub-face): collapse it before assembling");
    require(bc_values != nullptr, "bc_values is required");
    auto s = h->stream;
    const size_t nfd = (size_t)h->nf * h->nd, ncd = (size_t)h->nc * h->nd;
    double* in = h->vec_in.ensure(nfd + ncd);
    vec_in(h, in, bc_values, nfd);
    double* d_src = nullptr;
    if (source) {
      d_src = in + nfd;
      vec_in(h, d_src, source, ncd);
    }
    pfv::Timer tm;
    tm.start(s);
    if (!h->have_mech_system) {
      pfv::mpsa_assemble_system(*h);
      if (h->amg) h->amg->valid = false;
      if (h->block_pc) h->block_pc->for_val = nullptr;
      if (h->amg_block) h->amg_block->valid = false;
      h->perm_for_val = nullptr;
      h->win_for = h->win_rows_for = nullptr;
    }
    pfv::mpsa_assemble_rhs(*h, in, d_src);
    h->stats.assemble_ms = tm.stop(s);
    h->have_mech_system = true;
    h->active.P = &h->pat_Am;
    h->active.val = h->val[PFV_MAT_MECH_SYSTEM].p;
    h->active.diag = h->diag_m.p;
    h->active.rhs = h->rhs_m.p;
    h->active.n = h->nc * h->nd;
    h->active_bs = h->nd;
    h->active_is_grid = true;
    h->active.valid = true;
  });
}

pfv_status pfv_device_memory(pfv_ctx* h, int64_t* free_bytes, int64_t* total_bytes) {
  return guarded(h, [&] {
    require(free_bytes && total_bytes, "null output");
#ifdef PFV_EMULATE
    *free_bytes = *total_bytes = -1;  // host emulation: unknown
#else
    size_t f = 0, t = 0;
    PFV_HIP_CHECK(hipMemGetInfo(&f, &t));
    *free_bytes = (int64_t)f + (int64_t)h->pool.cached;  // blocks parked in the handle's cache are reusable
    *total_bytes = (int64_t)t;
#endif
  });
}

pfv_status pfv_host_alloc(size_t bytes, void** out) {
  if (!out) return PFV_ERR_ARGUMENT;
  *out = nullptr;
  if (bytes == 0) bytes = 8;
#ifdef PFV_EMULATE
  *out = std::malloc(bytes);
  return *out ? PFV_OK : PFV_ERR_HIP;
#else
  return hipHostMalloc(out, bytes, hipHostMallocDefault) == hipSuccess ? PFV_OK : PFV_ERR_HIP;
#endif
}

void pfv_host_free(void* p) {
  if (!p) return;
#ifdef PFV_EMULATE
  std::free(p);
#else
  (void)hipHostFree(p);
#endif
}

pfv_status pfv_active_size(pfv_ctx* h, int64_t* n) {
  return guarded(h, [&] {
    require(n != nullptr, "null output");
    *n = h->active.valid ? h->active.n : 0;
  });
}

pfv_status pfv_get_rhs(pfv_ctx* h, double* b) {
  return guarded(h, [&] {
    require(h->active.valid && b, "assemble first");
    be_d2h(b, h->active.rhs, sizeof(double) * (size_t)h->active.n, h->stream);
  });
}

pfv_status pfv_spmv(pfv_ctx* h, int which, const double* x, double* y) {
  return guarded(h, [&] {
    require(which >= 0 && which < PFV_NUM_MATS && x && y, "bad argument");
    require((h->have_symbolic || which == PFV_MAT_USER_SYSTEM) && h->filled[which], "matrix values have not been computed");
    const bool vs = which == PFV_MAT_VECTOR_SOURCE || which == PFV_MAT_BOUND_PRESSURE_VECTOR_SOURCE;
    if (!(vs && h->vs_implicit)) materialize_pattern(h, which);
    const pfv::CsrPattern& P = h->pattern_of(which);
    auto s = h->stream;
    pfv::Buf<double> dx, dy;
    dx.ensure((size_t)P.ncols);
    dy.ensure((size_t)P.nrows);
    be_h2d(dx.p, x, sizeof(double) * (size_t)P.ncols, s);
    if (vs && h->vs_implicit) pfv::spmv_vs_implicit(*h, h->val[which], dx.p, dy.p);
    else pfv::spmv(*h, P, h->val[which], dx.p, dy.p);
    be_d2h(y, dy.p, sizeof(double) * (size_t)P.nrows, s);
  });
}

pfv_status pfv_spmv_device(pfv_ctx* h, int which, const double* d_x, double* d_y) {
  return guarded(h, [&] {
    require(which >= 0 && which < PFV_NUM_MATS && d_x && d_y, "bad argument");
    require((h->have_symbolic || which == PFV_MAT_USER_SYSTEM) && h->filled[which], "matrix values have not been computed");
    if ((which == PFV_MAT_VECTOR_SOURCE || which == PFV_MAT_BOUND_PRESSURE_VECTOR_SOURCE) && h->vs_implicit) {
      pfv::spmv_vs_implicit(*h, h->val[which], d_x, d_y);
      return;
    }
    materialize_pattern(h, which);
    pfv::spmv(*h, h->pattern_of(which), h->val[which], d_x, d_y);
  });
}

pfv_status pfv_spmv_device_rows(pfv_ctx* h, int which, int64_t nrows, const double* d_x, double* d_y) {
  return guarded(h, [&] {
    require(which >= 0 && which < PFV_NUM_MATS && d_x && d_y, "bad argument");
    require((h->have_symbolic || which == PFV_MAT_USER_SYSTEM) && h->filled[which], "matrix values have not been computed");
    materialize_pattern(h, which);
    const pfv::CsrPattern& P = h->pattern_of(which);
    require(nrows >= 0 && nrows <= P.nrows, "nrows out of range");
    pfv::CsrPattern V;  // view of the leading rows (shares the index arrays)
    V.nrows = nrows;
    V.ncols = P.ncols;
    V.nnz = P.nrows ? (int64_t)((double)P.nnz * (double)nrows / (double)P.nrows) : 0;
    V.indptr.p = P.indptr.p;
    V.indices.p = P.indices.p;
    try {
      const bool sysmat = which == PFV_MAT_SYSTEM || which == PFV_MAT_MECH_SYSTEM || which == PFV_MAT_USER_SYSTEM;
      bool done = false;
      if (sysmat && nrows > 0 && P.nnz >= pfv::env_int("PFV_SPMV_WINDOW_MIN_NNZ", 20000)) {
        // the sharded Krylov loop's product with the owned rows: windowed kernel, window kept until
        // the matrix is assembled again
        if (h->win_rows_for != P.indices.p || h->win_rows_n != nrows) {
          V.nnz = P.nnz;  // (lidx is indexed by entry position)
          h->win_rows_checksum = 0;
          pfv::win_build(*h, V, h->win_rows);
          h->win_rows_for = P.indices.p;
          h->win_rows_n = nrows;
        }
        if (h->win_rows.ok) {
          pfv::LinSys sys;
          sys.P = &V;
          sys.val = h->val[which].p;
          sys.win = &h->win_rows;
          pfv::sys_spmv(*h, sys, d_x, d_y);
          done = true;
        }
      }
      if (!done) pfv::spmv(*h, V, h->val[which], d_x, d_y);
    } catch (...) {
      V.indptr.p = nullptr;
      V.indices.p = nullptr;
      throw;
    }
    V.indptr.p = nullptr;  // not owned
    V.indices.p = nullptr;
  });
}

pfv_status pfv_copy_device_vector(pfv_ctx* h, int which, double* d_dst, int64_t count) {
  return guarded(h, [&] {
    require(h->active.valid && d_dst && count >= 0 && count <= h->active.n, "bad argument / assemble first");
    require(which == 0 || which == 1, "which must be 0 (rhs) or 1 (diagonal)");
    pfv::be_d2d(d_dst, which == 0 ? h->active.rhs : h->active.diag, sizeof(double) * (size_t)count, h->stream);
  });
}

pfv_status pfv_set_stream(pfv_ctx* h, void* hip_stream) {
  return guarded(h, [&] {
#ifdef PFV_EMULATE
    (void)hip_stream;
#else
    pfv::be_sync(h->stream);
    h->stream = reinterpret_cast<hipStream_t>(hip_stream);  // NULL = the legacy default stream
#endif
  });
}

pfv_status pfv_reset_stream(pfv_ctx* h) {
  return guarded(h, [&] {
#ifndef PFV_EMULATE
    pfv::be_sync(h->stream);
    h->stream = h->own_stream;
#endif
  });
}

pfv_status pfv_get_device_rhs(pfv_ctx* h, double** d_b, double** d_diag) {
  return guarded(h, [&] {
    require(h->have_system, "assemble first");
    if (d_b) *d_b = h->rhs.p;
    if (d_diag) *d_diag = h->diag.p;
  });
}

pfv_status pfv_sync(pfv_ctx* h) {
  return guarded(h, [&] { pfv::be_sync(h->stream); });
}

pfv_status pfv_set_system(pfv_ctx* h, int64_t n, const int32_t* indptr, const int32_t* indices,
                          const double* data, const double* rhs) {
  return guarded(h, [&] {
    require(n > 0 && indptr && indices && data && rhs, "bad system");
    require(indptr[0] == 0, "indptr must start at 0");
    const int64_t nnz = indptr[n];
    require(nnz >= 0, "bad indptr");
    auto s = h->stream;
    pfv::CsrPattern& P = h->pat_user;
    P.nrows = P.ncols = n;
    P.nnz = nnz;
    int32_t* ip = P.indptr.ensure(n + 1);
    int32_t* ix = P.indices.ensure(std::max<int64_t>(nnz, 1));
    double* v = h->val[PFV_MAT_USER_SYSTEM].ensure(std::max<int64_t>(nnz, 1));
    double* b = h->rhs_u.ensure(n);
    double* dg = h->diag_u.ensure(n);
    be_h2d(ip, indptr, sizeof(int32_t) * (size_t)(n + 1), s);
    be_h2d(ix, indices, sizeof(int32_t) * (size_t)nnz, s);
    be_h2d(v, data, sizeof(double) * (size_t)nnz, s);
    be_h2d(b, rhs, sizeof(double) * (size_t)n, s);
    int32_t* st = h->status.ensure(16);
    pfv::be_memset(st, 0, sizeof(int32_t) * 4, s);
    pfv::parallel_for(s, n, PFV_LAMBDA(int64_t i) {
      double d = 0.0;
      bool bad = false;
      for (int e = ip[i]; e < ip[i + 1]; ++e) {
        const int cidx = ix[e];
        if (cidx < 0 || cidx >= n) bad = true;
        else if (cidx == i) d += v[e];
      }
      dg[i] = d;
      if (bad) pfv::atomic_max_i32(st + 1, 1);
      if (!(d != 0.0) || !(d == d)) pfv::atomic_max_i32(st + 0, (int32_t)(i < 0x7fffffff ? i + 1 : 0x7fffffff));
    });
    int32_t sth[2];
    be_d2h(sth, st, sizeof(sth), s);
    require(!sth[1], "column index out of range");
    if (sth[0])
      throw pfv::Error(PFV_ERR_UNSUPPORTED, "zero diagonal entry in row " + std::to_string(sth[0] - 1) +
                                                ": the Jacobi-preconditioned solver does not apply");
    int mr = 0;
    for (int64_t i = 0; i < n; ++i) mr = std::max(mr, indptr[i + 1] - indptr[i]);
    P.max_row = mr;
    h->filled[PFV_MAT_USER_SYSTEM] = true;
    h->active.P = &P;
    h->active.val = v;
    h->active.diag = dg;
    h->active.rhs = b;
    h->active.n = n;
    h->active_bs = 1;
    if (h->amg) h->amg->valid = false;
      if (h->block_pc) h->block_pc->for_val = nullptr;
    if (h->amg_block) h->amg_block->valid = false;
    h->perm_for_val = nullptr;
    h->win_for = h->win_rows_for = nullptr;
    h->active_is_grid = false;
    h->active.valid = true;
  });
}

pfv_status pfv_amg_setup(pfv_ctx* h, int64_t n_own) {
  return guarded(h, [&] {
    require(h->active.valid, "assemble first");
    const int bs = h->active_bs;
    const int64_t n = h->active.n;
    if (n_own <= 0) n_own = n;
    require(n_own <= n && n_own % bs == 0, "bad block size");
    auto s = h->stream;
    if (!h->amg_block) h->amg_block = std::make_unique<pfv::Amg>();
    h->amg_block->valid = false;
    h->amg_block->dist.reset();  // (a coupled hierarchy of an earlier pfv_amg_setup_sharded)
    const pfv::CsrPattern* P = h->active.P;
    const double* val = h->active.val;
    if (n_own < n) {
      // leading block: drop the columns >= n_own (halo cells of the subdomain)
      const int32_t* ip = P->indptr;
      const int32_t* ix = P->indices;
      pfv::Buf<int32_t> len;
      pfv::Buf<int64_t> pos;
      int32_t* ln = len.ensure(n_own + 1);
      int64_t* ps = pos.ensure(n_own + 1);
      pfv::parallel_for(s, n_own, PFV_LAMBDA(int64_t r) {
        int m = 0;
        for (int e = ip[r]; e < ip[r + 1]; ++e) m += ix[e] < n_own ? 1 : 0;
        ln[r] = m;
      });
      pfv::exclusive_scan<int32_t, int64_t>(s, h->scratch, ln, ps, (size_t)n_own);
      const int64_t nnz = pfv::read_scalar<int64_t>(s, ps + n_own);
      pfv::CsrPattern& B = h->pat_block;
      B.nrows = B.ncols = n_own;
      B.nnz = nnz;
      B.max_row = P->max_row;
      int32_t* bp = B.indptr.ensure(n_own + 1);
      int32_t* bx = B.indices.ensure(std::max<int64_t>(nnz, 1));
      double* bv = h->val_block.ensure(std::max<int64_t>(nnz, 1));
      pfv::parallel_for(s, n_own + 1, PFV_LAMBDA(int64_t r) {
        bp[r] = (int32_t)ps[r];
        if (r < n_own) {
          int64_t o = ps[r];
          for (int e = ip[r]; e < ip[r + 1]; ++e)
            if (ix[e] < n_own) {
              bx[o] = ix[e];
              bv[o] = val[e];
              ++o;
            }
        }
      });
      P = &B;
      val = bv;
    }
    const pfv::WinCsr* bw = nullptr;
    if (P->nnz >= pfv::env_int("PFV_SPMV_WINDOW_MIN_NNZ", 20000)) {
      pfv::win_build(*h, *P, h->win_block);
      bw = &h->win_block;
    }
    pfv::amg_setup(*h, *h->amg_block, *P, val, bs, h->active.diag, bw);  // the leading block keeps its diagonal
    h->stats.amg_setup_ms = h->amg_block->setup_ms;
    h->stats.amg_operator_complexity = h->amg_block->op_complexity;
    h->stats.amg_levels = (int64_t)h->amg_block->nlev;
    h->stats.amg_coarsest_rows = h->amg_block->lev[h->amg_block->nlev - 1]->n;
    h->stats.amg_maps_reused = h->amg_block->reused ? 1 : 0;
  });
}

pfv_status pfv_amg_setup_sharded(pfv_ctx* h, int64_t n_own, const pfv_shard_hooks* hooks, int rank, int world,
                                 int n_peers, const int32_t* peers, const int64_t* send_ptr,
                                 const int32_t* send_idx, const int64_t* recv_ptr, const int32_t* recv_pos) {
  return guarded(h, [&] {
    require(h->active.valid, "assemble first");
    const int bs = h->active_bs;
    const int64_t n = h->active.n;
    require(n_own > 0 && n_own <= n && n_own % bs == 0 && n % bs == 0, "n_own out of range");
    require(hooks && hooks->sendrecv && hooks->allgather, "the sendrecv and allgather hooks are required");
    require(world >= 1 && rank >= 0 && rank < world, "bad rank / world");
    require(n_peers >= 0 && (n_peers == 0 || (peers && send_ptr && recv_ptr)), "bad halo plan");
    const int64_t own_cells = n_own / bs, loc_cells = n / bs;
    auto plan = std::make_unique<pfv::AmgPlan>();
    plan->n_peers = n_peers;
    plan->peers.assign(peers, peers + n_peers);
    plan->send_ptr.assign(1, 0);
    plan->recv_ptr.assign(1, 0);
    if (n_peers > 0) {
      plan->send_ptr.assign(send_ptr, send_ptr + n_peers + 1);
      plan->recv_ptr.assign(recv_ptr, recv_ptr + n_peers + 1);
    }
    require(plan->send_ptr.front() == 0 && plan->recv_ptr.front() == 0, "halo plan: offsets must start at 0");
    for (int p = 0; p < n_peers; ++p) {
      require(plan->send_ptr[p + 1] >= plan->send_ptr[p] && plan->recv_ptr[p + 1] >= plan->recv_ptr[p],
              "halo plan: offsets must be non-decreasing");
      require(peers[p] >= 0 && peers[p] < world && peers[p] != rank, "peer out of range");
    }
    const int64_t ns = plan->send_ptr.back(), nr = plan->recv_ptr.back();
    require((ns == 0 || send_idx) && (nr == 0 || recv_pos), "halo plan: index lists missing");
    for (int64_t k = 0; k < ns; ++k) require(send_idx[k] >= 0 && send_idx[k] < own_cells, "halo plan: send index is not an owned cell");
    for (int64_t k = 0; k < nr; ++k)
      require(recv_pos[k] >= own_cells && recv_pos[k] < loc_cells, "halo plan: receive position is not a halo cell");
    require(nr == loc_cells - own_cells, "halo plan: every halo cell must be received exactly once");
    plan->h_send_idx.assign(send_idx, send_idx + ns);
    plan->h_recv_pos.assign(recv_pos, recv_pos + nr);
    plan->n_own = own_cells;
    plan->n_halo = loc_cells - own_cells;
    pfv::amg_plan_upload(*h, *plan);
    if (!h->amg_block) h->amg_block = std::make_unique<pfv::Amg>();
    pfv::Amg& amg = *h->amg_block;
    amg.valid = false;
    if (!amg.dist) amg.dist = std::make_unique<pfv::AmgDist>();
    amg.dist->hooks = *hooks;
    amg.dist->rank = rank;
    amg.dist->world = world;
    amg.dist->plan.clear();
    amg.dist->plan.push_back(std::move(plan));
    // the rows of the owned unknowns over all local columns (owned + halo)
    const pfv::CsrPattern& P = *h->active.P;
    pfv::CsrPattern& V = amg.dist->rows0;  // (borrows the index arrays: see ~AmgDist)
    V.nrows = n_own;
    V.ncols = n;
    V.nnz = P.nnz;
    V.max_row = P.max_row;
    V.indptr.p = P.indptr.p;
    V.indices.p = P.indices.p;
    // the windows of the owned rows (built beside the face kernel, or by the previous sharded solve): the windows of the
    // strength-filtered operator are derived from them instead of hashed and sorted from scratch (spmv_win.inc: win_derive)
    const pfv::WinCsr* win0 = (bs == 1 && h->win_rows.ok && h->win_rows_for == P.indices.p && h->win_rows_n == n_own &&
                               h->win_rows.nrows == n_own && pfv::env_int("PFV_SHARD_WIN0", 1) != 0)
                                  ? &h->win_rows
                                  : nullptr;
    pfv::amg_setup(*h, amg, V, h->active.val, bs, nullptr, win0);
    h->stats.amg_setup_ms = amg.setup_ms;
    h->stats.amg_operator_complexity = amg.op_complexity;
    h->stats.amg_levels = (int64_t)(amg.nlev + (amg.dist->glob ? amg.dist->glob->nlev - 1 : 0));
    h->stats.amg_coarsest_rows = amg.dist->gN;
    h->stats.amg_maps_reused = amg.reused ? 1 : 0;
  });
}

pfv_status pfv_amg_apply_device(pfv_ctx* h, const double* d_r, double* d_z) {
  return guarded(h, [&] {
    require(h->amg_block && h->amg_block->valid, "pfv_amg_setup first");
    require(d_r && d_z && d_r != d_z, "bad vectors");
    if (h->amg_block->dist) pfv::amg_apply_dist(*h, *h->amg_block, d_r, d_z);
    else pfv::amg_cycle(*h, *h->amg_block, 0, d_r, d_z);
  });
}

pfv_status pfv_set_preconditioner(pfv_ctx* h, int kind) {
  return guarded(h, [&] {
    require(kind == PFV_PRECOND_JACOBI || kind == PFV_PRECOND_AMG || (kind == PFV_PRECOND_BLOCK && h->block_pc),
            "unknown preconditioner (PFV_PRECOND_BLOCK: pfv_set_block_preconditioner first)");
    h->precond = kind;
  });
}

pfv_status pfv_set_block_preconditioner(pfv_ctx* h, int64_t n_blocks, const int64_t* block_ptr, int gauss_seidel) {
  return guarded(h, [&] {
    require(n_blocks >= 1 && block_ptr != nullptr && block_ptr[0] == 0, "bad block layout");
    for (int64_t k = 0; k < n_blocks; ++k) require(block_ptr[k + 1] > block_ptr[k], "blocks must be non-empty and ascending");
    if (!h->block_pc) h->block_pc = std::make_unique<pfv::BlockPc>();
    h->block_pc->ptr.assign(block_ptr, block_ptr + n_blocks + 1);
    h->block_pc->gs = gauss_seidel != 0;
    h->block_pc->for_val = nullptr;
    h->precond = PFV_PRECOND_BLOCK;
  });
}

// The system the Krylov loop works on: the active one, renumbered along the cell order when it is a
// grid system (reorder.inc), with the SpMV windows of its pattern (spmv_win.inc).  Copies and
// windows are kept until the matrix is assembled again.
static pfv::LinSys solver_system(pfv_ctx* h, bool& permuted) {
  pfv::LinSys sys = h->active;
  const int bs = h->active_bs;
  permuted = false;
  if (h->active_is_grid && sys.n == h->nc * bs && pfv::env_int("PFV_REORDER", 1) != 0 &&
      h->nc >= pfv::env_int("PFV_REORDER_MIN_CELLS", 256)) {
    if (!h->have_cell_order) pfv::build_cell_order(*h);
  }
  if (h->active_is_grid && h->have_cell_order && !h->cell_order_identity && sys.n == h->nc * bs &&
      pfv::env_int("PFV_REORDER", 1) != 0 && h->nc >= pfv::env_int("PFV_REORDER_MIN_CELLS", 256)) {
    if (h->perm_for_val != sys.val) {
      pfv::permute_matrix(*h, sys, bs);
      h->perm_for_val = sys.val;
      h->win_for = h->win_rows_for = nullptr;
      if (h->amg) h->amg->valid = false;
      if (h->block_pc) h->block_pc->for_val = nullptr;  // (the copy's buffers are shared by the flow and mechanics systems)
    }
    pfv::permute_vector(*h, sys.n, bs, sys.rhs, h->rhs_perm.ensure(sys.n), true);
    sys.P = &h->pat_perm;
    sys.val = h->val_perm.p;
    sys.diag = h->diag_perm.p;
    sys.rhs = h->rhs_perm.p;
    permuted = true;
  }
  sys.win = nullptr;
  if (sys.P->nnz >= pfv::env_int("PFV_SPMV_WINDOW_MIN_NNZ", 20000)) {
    if (h->win_for != sys.P->indices.p || !h->win_sys.ok || pfv::env_int("PFV_SPMV_WINDOW", 1) == 0) {
      h->win_sys_checksum = 0;
      pfv::win_build(*h, *sys.P, h->win_sys);
      h->win_for = sys.P->indices.p;
      if (h->win_sys.ok && sys.P == &h->pat_A && h->have_symbolic) h->win_sys_checksum = h->pat_A_checksum;
    }
    sys.win = &h->win_sys;
  }
  return sys;
}

pfv_status pfv_solve(pfv_ctx* h, int method, double rtol, int maxit, int restart, const double* x0,
                     double* x, pfv_solve_info* info) {
  pfv::SolveResult res;
  pfv_status st = guarded(h, [&] {
    require(h->active.valid, "assemble first");
    require(x != nullptr, "x is required");
    require(method == PFV_SOLVE_CG || method == PFV_SOLVE_BICGSTAB || method == PFV_SOLVE_GMRES,
            "method must be PFV_SOLVE_CG, PFV_SOLVE_BICGSTAB or PFV_SOLVE_GMRES");
    require(rtol > 0 && maxit > 0, "rtol and maxit must be positive");
    auto s = h->stream;
    const size_t n = (size_t)h->active.n;
    double* dx = h->xsol.ensure(n);
    if (x0) vec_in(h, dx, x0, n); else pfv::be_memset(dx, 0, n * sizeof(double), s);
    pfv::Timer tm;
    tm.start(s);
    bool permuted = false;
    const pfv::LinSys sys = solver_system(h, permuted);
    h->stats.solve_renumbered = permuted ? 1 : 0;
    double* dxs = dx;
    if (permuted) {
      dxs = h->x_perm.ensure(n);
      if (x0) pfv::permute_vector(*h, (int64_t)n, h->active_bs, dx, dxs, true);
      else pfv::be_memset(dxs, 0, n * sizeof(double), s);
    }
    pfv::Precond M;
    const pfv::Precond* Mp = nullptr;
    const long long launches_before_setup = pfv::launch_counter().load(std::memory_order_relaxed);
    if (h->precond == PFV_PRECOND_AMG) {
      if (!h->amg) h->amg = std::make_unique<pfv::Amg>();
      if (!h->amg->valid || h->amg_for_val != sys.val) {
        pfv::amg_setup(*h, *h->amg, *sys.P, sys.val, h->active_bs, sys.diag, sys.win);
        h->amg_for_val = sys.val;
        h->stats.amg_setup_ms = h->amg->setup_ms;
        h->stats.amg_operator_complexity = h->amg->op_complexity;
        h->stats.amg_levels = (int64_t)h->amg->nlev;
        h->stats.amg_coarsest_rows = h->amg->lev[h->amg->nlev - 1]->n;
        h->stats.amg_maps_reused = h->amg->reused ? 1 : 0;
        h->stats.amg_stale_rematches = h->amg->stale_rematches;
        h->stats.amg_level0_nnz = h->amg->lev[0]->P->nnz;
        h->stats.amg_filter_theta = h->amg->filter_level0 ? h->amg->filter_theta : 0.0;
      }
      M.amg = h->amg.get();
      Mp = &M;
    } else if (h->precond == PFV_PRECOND_BLOCK) {
      require(h->block_pc && !permuted, "pfv_set_block_preconditioner first (user systems only)");
      require(h->block_pc->ptr.back() == (int64_t)n, "the block layout does not cover the system");
      if (h->block_pc->for_val != sys.val) {
        pfv::blockpc_setup(*h, *h->block_pc, *sys.P, sys.val);
        h->stats.amg_setup_ms = h->block_pc->setup_ms;
      }
      M.blocks = h->block_pc.get();
      M.P = sys.P;
      M.val = sys.val;
      M.diag = sys.diag;
      Mp = &M;
    }
    const long long launches_before_loop = pfv::launch_counter().load(std::memory_order_relaxed);
    h->stats.amg_setup_launches = (int64_t)(launches_before_loop - launches_before_setup);
    res = method == PFV_SOLVE_GMRES
              ? pfv::gmres_solve(*h, sys, rtol, maxit, restart, dxs, x0 == nullptr, Mp)
              : pfv::krylov_solve(*h, sys, method, rtol, maxit, dxs, x0 == nullptr, Mp);
    if (M.amg && x0 == nullptr) h->amg->note_iterations(res.iterations, rtol, method, res.converged);
    h->stats.solve_launches = (int64_t)(pfv::launch_counter().load(std::memory_order_relaxed) - launches_before_loop);
    if (permuted) pfv::permute_vector(*h, (int64_t)n, h->active_bs, dxs, dx, false);
    h->stats.solve_ms = tm.stop(s);
    if (h->vectors_on_device) pfv::be_d2d(x, dx, n * sizeof(double), s); else be_d2h(x, dx, n * sizeof(double), s);
  });
  if (info) {
    info->iterations = res.iterations;
    info->converged = res.converged ? 1 : 0;
    info->rel_residual = res.relres;
    info->solve_ms = h ? h->stats.solve_ms : 0.0;
  }
  if (st == PFV_OK && !res.converged) {
    h->err = "Krylov solver did not reach the requested tolerance";
    return PFV_ERR_NOT_CONVERGED;
  }
  return st;
}


pfv_status pfv_solve_sharded(pfv_ctx* h, int method, double rtol, int maxit, int64_t n_own,
                             const pfv_shard_hooks* hooks, double* d_work, double* d_x_owned,
                             pfv_solve_info* info) {
  pfv::SolveResult res;
  pfv_status st = guarded(h, [&] {
    require(h->active.valid, "assemble first");
    require(hooks && hooks->exchange_halo && hooks->allreduce_sum, "both hooks are required");
    require(d_work && d_x_owned, "work space and x are required");
    require(method == PFV_SOLVE_CG || method == PFV_SOLVE_BICGSTAB, "a sharded solve runs PFV_SOLVE_CG or PFV_SOLVE_BICGSTAB");
    require(rtol > 0 && maxit > 0, "rtol and maxit must be positive");
    const int64_t n_loc = h->active.n;
    require(n_own > 0 && n_own <= n_loc && n_own % h->active_bs == 0, "n_own out of range");
    auto s = h->stream;
    pfv::Timer tm;
    tm.start(s);
    const pfv::CsrPattern& P = *h->active.P;
    RowsView rows(P, n_own);
    pfv::LinSys sys = h->active;
    sys.P = &rows.V;
    sys.n = n_own;
    sys.win = nullptr;
    if (P.nnz >= pfv::env_int("PFV_SPMV_WINDOW_MIN_NNZ", 20000)) {
      if (h->win_rows_for != P.indices.p || h->win_rows_n != n_own) {
        h->win_rows_checksum = 0;
        pfv::win_build(*h, rows.V, h->win_rows);
        h->win_rows_for = P.indices.p;
        if (h->win_rows.ok && &P == &h->pat_A && h->have_symbolic) h->win_rows_checksum = h->pat_A_checksum;
        h->win_rows_n = n_own;
      }
      if (h->win_rows.ok) sys.win = &h->win_rows;
    }
    pfv::Precond M;
    const pfv::Precond* Mp = nullptr;
    if (h->precond == PFV_PRECOND_AMG) {
      require(h->amg_block && h->amg_block->valid && h->amg_block->lev[0]->n == n_own,
              "pfv_amg_setup(n_own) first: the sharded solve preconditions with the hierarchy of the owned block");
      M.amg = h->amg_block.get();
      Mp = &M;
    } else if (h->precond == PFV_PRECOND_BLOCK) {
      // block lower-triangular sweep over the blocks of the OWNED unknowns (pfv_set_block_preconditioner with a layout
      // that covers [0, n_own)): couplings to other ranks' unknowns are left to the Krylov loop (block Jacobi across
      // ranks, Gauss-Seidel over the (variable, subdomain) blocks inside a rank) -- every diagonal block is extracted
      // from the owned rows and columns, the halo columns lie behind them and never enter a block or its sweep
      require(h->block_pc != nullptr, "pfv_set_block_preconditioner first");
      require(h->block_pc->ptr.back() == n_own, "the block layout must cover exactly the owned unknowns");
      if (h->block_pc->for_val != sys.val) {
        pfv::blockpc_setup(*h, *h->block_pc, rows.V, sys.val);
        h->stats.amg_setup_ms = h->block_pc->setup_ms;
      }
      M.blocks = h->block_pc.get();
      M.P = &rows.V;
      M.val = sys.val;
      M.diag = sys.diag;
      Mp = &M;
    }
    pfv::be_memset(d_x_owned, 0, sizeof(double) * (size_t)n_own, s);
    pfv::be_memset(d_work, 0, sizeof(double) * (size_t)(2 * n_loc + 8), s);
    h->shard_overlap = false;
#ifndef PFV_EMULATE
    {
      // PFV_SHARD_OVERLAP=1: halo exchange of the Krylov products on the second stream beside the row blocks that need
      // no halo entry (linalg.inc: shard_spmv) -- with the native RCCL hooks only (a caller's hooks are not known to
      // honour a foreign stream).  OFF by default: no two-GPU box has run it yet.  =2 (tests): every second block
      // counts as a boundary block.
      const int ov = pfv::env_int("PFV_SHARD_OVERLAP", 0);
      if (ov != 0 && sys.win && sys.win->ok && h->aux_stream && hooks->exchange_halo == &pfv::rccl_exchange_halo) {
        const pfv::WinCsr& W = *sys.win;
        const int64_t nblk = W.nblk;
        int32_t* lst = h->shard_blocks.ensure(2 * nblk + 2);
        int32_t* cnt = h->status.ensure(16);
        pfv::be_memset(cnt + 12, 0, 2 * sizeof(int32_t), s);
        const int64_t* wptr = W.wptr64;
        const int32_t* wlen = W.wlen;
        const int32_t* wcol = W.wcol.p;
        int32_t* tmp_b = lst + nblk;  // boundary blocks collected behind, moved up below
        pfv::parallel_for(s, nblk, PFV_LAMBDA(int64_t b) {
          const int64_t w0 = wptr[b];
          const int wn = wlen ? wlen[b] : (int)(wptr[b + 1] - w0);
          const bool bnd = ov == 2 ? (b & 1) != 0 : (wn > 0 && wcol[w0 + wn - 1] >= n_own);  // (window columns ascend)
          if (bnd) tmp_b[atomicAdd(cnt + 13, 1)] = (int32_t)b;
          else lst[atomicAdd(cnt + 12, 1)] = (int32_t)b;
        });
        int32_t hc[2];
        be_d2h(hc, cnt + 12, sizeof(hc), s);
        h->shard_n_interior = hc[0];
        h->shard_n_boundary = hc[1];
        pfv::be_d2d(lst + hc[0], tmp_b, sizeof(int32_t) * (size_t)hc[1], s);
        h->shard_overlap = true;
      }
    }
#endif
    h->shard = hooks;
    h->shard_work = d_work;
    h->shard_nloc = n_loc;
    try {
      res = pfv::krylov_solve(*h, sys, method, rtol, maxit, d_x_owned, true, Mp);
    } catch (...) {
      h->shard = nullptr;
      throw;
    }
    h->shard = nullptr;
    if (M.amg) h->amg_block->note_iterations(res.iterations, rtol, method, res.converged);
    h->stats.solve_ms = tm.stop(s);
  });
  if (info) {
    info->iterations = res.iterations;
    info->converged = res.converged ? 1 : 0;
    info->rel_residual = res.relres;
    info->solve_ms = h ? h->stats.solve_ms : 0.0;
  }
  if (st == PFV_OK && !res.converged) {
    h->err = "Krylov solver did not reach the requested tolerance";
    return PFV_ERR_NOT_CONVERGED;
  }
  return st;
}

pfv_status pfv_get_stats(pfv_ctx* h, pfv_stats* out) {
  return guarded(h, [&] {
    require(out != nullptr, "null output");
    *out = h->stats;
  });
}

pfv_status pfv_get_stats_n(pfv_ctx* h, void* out, size_t struct_size) {
  return guarded(h, [&] {
    require(out != nullptr, "null output");
    std::memcpy(out, &h->stats, std::min(struct_size, sizeof(pfv_stats)));
  });
}

pfv_status pfv_time_kernel(pfv_ctx* h, int kernel, int reps, double* avg_ms) {
  return guarded(h, [&] {
    require(avg_ms != nullptr && reps > 0, "bad argument");
    auto s = h->stream;
    pfv::Timer tm;
    if (kernel == PFV_KERNEL_SPMV_A) {
      require(h->have_system, "assemble first");
      const size_t n = (size_t)h->nc;
      double* x = h->kry[7].ensure(n);
      double* y = h->kry[8].ensure(n);
      pfv::be_d2d(x, h->rhs.p, n * sizeof(double), s);
      // what the Krylov loop launches (renumbered system, windowed kernel when the pattern allows it)
      require(h->active.valid && h->active.P == &h->pat_A, "the flow system must be the active one");
      bool permuted = false;
      const pfv::LinSys sys = solver_system(h, permuted);
      pfv::sys_spmv(*h, sys, x, y);
      tm.start(s);
      for (int i = 0; i < reps; ++i) pfv::sys_spmv(*h, sys, x, y);
      *avg_ms = tm.stop(s) / reps;
    } else if (kernel == PFV_KERNEL_AMG_SMOOTH) {
      pfv::Amg* amg = (h->amg && h->amg->valid) ? h->amg.get() : h->amg_block.get();  // single-GPU / sharded
      require(amg && amg->valid && amg->nlev > 0, "no AMG hierarchy (solve with PFV_PRECOND_AMG first)");
      pfv::AmgLevel& L = *amg->lev[0];
      const size_t n = (size_t)std::max(L.n, L.m);  // (coupled hierarchy: the input carries the halo entries)
      double* x = h->kry[7].ensure(n);
      double* y = h->kry[8].ensure(n);
      double* b = h->kry[6].ensure(n);
      pfv::be_memset(x, 0, n * sizeof(double), s);
      pfv::be_memset(b, 0, n * sizeof(double), s);
      pfv::amg_spmv(*h, *amg, L, x, y, b);
      tm.start(s);
      for (int i = 0; i < reps; ++i) pfv::amg_spmv(*h, *amg, L, x, y, b);
      *avg_ms = tm.stop(s) / reps;
    } else if (kernel == PFV_KERNEL_TRIAD) {
      const size_t n = size_t(1) << 27;
      pfv::Buf<double> buf;
      double* a = buf.ensure(3 * n);
      double* b = a + n;
      double* cc = b + n;
      pfv::be_memset(a, 0, 3 * n * sizeof(double), s);
      pfv::D2* a2 = reinterpret_cast<pfv::D2*>(a);
      const pfv::D2* b2 = reinterpret_cast<const pfv::D2*>(b);
      const pfv::D2* c2 = reinterpret_cast<const pfv::D2*>(cc);
      auto triad = [&] {  // 16 bytes per lane and stream, four independent loads per stream in flight
#ifdef PFV_EMULATE
        for (size_t i = 0; i < n / 2; ++i) {
          a2[i].x = b2[i].x + 0.5 * c2[i].x;
          a2[i].y = b2[i].y + 0.5 * c2[i].y;
        }
#else
        PFV_LAUNCH(pfv::k_triad, dim3(256 * 16), dim3(256), 0, s, (int64_t)(n / 2), a2, b2, c2);
        PFV_HIP_CHECK(hipGetLastError());
#endif
      };
      triad();
      tm.start(s);
      for (int i = 0; i < reps; ++i) triad();
      *avg_ms = tm.stop(s) / reps;
    } else if (kernel == PFV_KERNEL_READ) {
#ifdef PFV_EMULATE
      *avg_ms = 0.0;
#else
      const size_t n = size_t(1) << 28;
      pfv::Buf<double> buf;
      double* a = buf.ensure(n + 8);
      pfv::be_memset(a, 0, (n + 8) * sizeof(double), s);
      auto rd = [&] {
        PFV_LAUNCH(pfv::k_read_stream, dim3(256 * 16), dim3(256), 0, s, (int64_t)(n / 2),
                           reinterpret_cast<const pfv::D2*>(a), a + n);
        PFV_HIP_CHECK(hipGetLastError());
      };
      rd();
      tm.start(s);
      for (int i = 0; i < reps; ++i) rd();
      *avg_ms = tm.stop(s) / reps;
#endif
    } else if (kernel == PFV_KERNEL_NODE) {
      require(h->have_numeric, "discretize first");
      tm.start(s);
      for (int i = 0; i < reps; ++i) pfv::run_node_kernel(*h);
      *avg_ms = tm.stop(s) / reps;
    } else if (kernel == PFV_KERNEL_FACE) {
      require(h->have_numeric, "discretize first");
      const bool with_vs = h->filled[PFV_MAT_VECTOR_SOURCE];
      tm.start(s);
      for (int i = 0; i < reps; ++i) pfv::run_face_kernel(*h, with_vs);
      *avg_ms = tm.stop(s) / reps;
    } else {
      require(false, "unknown kernel id");
    }
  });
}

pfv_status pfv_debug_copy(pfv_ctx* h, int which, double* dst, int64_t count) {
  return guarded(h, [&] {
    require(h->have_numeric && dst && count >= 0 && count <= (which == 0 ? h->tab_len : h->tabb_len), "bad argument");
    const double* src = which == 0 ? h->tab.p : h->tabb.p;
    be_d2h(dst, src, sizeof(double) * (size_t)count, h->stream);
  });
}

}  // extern "C"

#include "rccl_hooks.inc"
#include "csr_algebra.inc"
