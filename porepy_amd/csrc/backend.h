// backend.h — thin execution layer under the porefv kernels.
//
// Product build (hipcc --offload-arch=gfx950): kernels run on the MI355X; one HIP stream
// per handle; rocprim for radix sort / scans; HIP events for phase timing.
//
// PFV_EMULATE build (plain g++, test infrastructure only, never shipped or loaded by
// porepy_amd/): the SAME kernel bodies run sequentially on the host so that topology,
// indexing and numerics of every kernel can be checked against the oracle in a container
// without a GPU.  To make that possible kernels follow a discipline:
//   * element kernels are lambdas over a flat index          -> pfv::parallel_for
//   * cooperative kernels are written for ONE 64-lane wavefront per work item, as
//     lane-strided loops (PFV_LANES) separated by w.sync(); values that cross a sync
//     live in LDS (w.lds), never in per-lane registers         -> pfv::wave_for
#pragma once

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <vector>

#ifdef PFV_EMULATE
#define PFV_HD
#define PFV_FN inline
#define PFV_LAMBDA [=]
#else
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#define PFV_HD __host__ __device__
// helpers that take LDS pointers must be inlined into the kernel, otherwise the pointers decay
// to the flat address space and every LDS access becomes a (slow, vmcnt-coupled) flat_load/store
#define PFV_FN __host__ __device__ __attribute__((always_inline)) inline
#define PFV_LAMBDA [=] __device__
// (the kernel name is parenthesised here: HIP_KERNEL_NAME(k<a, b>) has lost its protecting parentheses by the time the
// argument is substituted)
#define PFV_LAUNCH(kernel, ...)                  \
  do {                                           \
    ::pfv::launch_counter().fetch_add(1, std::memory_order_relaxed); \
    hipLaunchKernelGGL((kernel), __VA_ARGS__);   \
  } while (0)
#endif
// lane-strided loop of the wavefront (or lane group) that owns work item `w`
#define PFV_LANES(i, n) for (int i = w.lane; i < (int)(n); i += w.width)

namespace pfv {

struct Error : std::runtime_error {
  int status;
  Error(int s, const std::string& m) : std::runtime_error(m), status(s) {}
};

constexpr int kWave = 64;

// kernel dispatches issued by this library (every launch site goes through PFV_LAUNCH): pfv_stats.solve_launches
// (relaxed atomic: handles on different threads, or a discretization on the second stream beside a solve, all count
// here; differences of it are statistics, not synchronisation)
inline std::atomic<long long>& launch_counter() {
  static std::atomic<long long> c{0};
  return c;
}

#ifndef PFV_EMULATE
#define PFV_HIP_CHECK(expr)                                                              \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess)                                                                \
      throw ::pfv::Error(3, std::string(#expr) + ": " + hipGetErrorString(e_));          \
  } while (0)
using stream_t = hipStream_t;
#else
using stream_t = int;
#endif

// ---------------------------------------------------------------- memory
inline void* raw_malloc(size_t bytes) {
#ifdef PFV_EMULATE
  void* p = std::malloc(bytes);
  if (!p) throw Error(3, "host emulation: out of memory");
  return p;
#else
  void* p = nullptr;
  PFV_HIP_CHECK(hipMalloc(&p, bytes));
  return p;
#endif
}
inline void raw_free(void* p) {
#ifdef PFV_EMULATE
  std::free(p);
#else
  (void)hipFree(p);
#endif
}

// Per-handle cache of device blocks.  The setup phases allocate a few dozen temporaries per call;
// hipMalloc / hipFree cost tens of microseconds each on a good day and hipFree synchronises the
// device (on a loaded host the symbolic phase went from 16 to 37 ms that way).  Freed blocks are
// kept and handed out again to requests of a similar size.  All work of a handle is ordered on its
// one stream (pfv_set_stream synchronises before switching), so a block that is reused while
// kernels that read its previous contents are still queued is safe.
struct MemPool {
  struct Blk { void* p; size_t bytes; };
  std::vector<Blk> free_list;
  std::unordered_map<void*, size_t> live;
  size_t cached = 0;
  static constexpr size_t kMaxCached = size_t(48) << 30;
  void* take(size_t bytes) {
    size_t best = (size_t)-1, best_bytes = (size_t)-1;
    for (size_t i = 0; i < free_list.size(); ++i) {
      const size_t b = free_list[i].bytes;
      if (b >= bytes && b <= 2 * bytes + (size_t(1) << 16) && b < best_bytes) { best = i; best_bytes = b; }
    }
    void* p;
    if (best != (size_t)-1) {
      p = free_list[best].p;
      cached -= best_bytes;
      free_list[best] = free_list.back();
      free_list.pop_back();
      live[p] = best_bytes;
    } else {
      p = raw_malloc(bytes);
      live[p] = bytes;
    }
    return p;
  }
  bool give(void* p) {  // false: not one of ours
    auto it = live.find(p);
    if (it == live.end()) return false;
    const size_t b = it->second;
    live.erase(it);
    if (cached + b > kMaxCached) raw_free(p);
    else { free_list.push_back({p, b}); cached += b; }
    return true;
  }
  void trim() {
    for (auto& b : free_list) raw_free(b.p);
    free_list.clear();
    cached = 0;
  }
  ~MemPool() { trim(); }
};
inline thread_local MemPool* tls_pool = nullptr;  // set for the duration of an API call (porefv.hip: guarded)
struct PoolScope {
  MemPool* prev;
  explicit PoolScope(MemPool* p) : prev(tls_pool) { tls_pool = p; }
  ~PoolScope() { tls_pool = prev; }
};

inline void* be_malloc(size_t bytes) {
  if (bytes == 0) bytes = 8;
  return tls_pool ? tls_pool->take(bytes) : raw_malloc(bytes);
}
inline void be_free(void* p) {
  if (!p) return;
  if (tls_pool && tls_pool->give(p)) return;
  raw_free(p);
}
#ifndef PFV_EMULATE
// Waiting for the stream: poll instead of sleeping on the completion interrupt.  A step has ~70
// points where the host needs a number from the device (sizes of the next allocation, convergence);
// with hipStreamSynchronize each costs a wake-up, which on a loaded host went from ~20 us to ~0.5 ms
// (107 -> 178 ms per step on such a box).  PFV_SPIN_WAIT=0 goes back to blocking waits.
inline bool spin_wait_enabled() {
  static const bool on = [] {
    const char* e = std::getenv("PFV_SPIN_WAIT");
    return !(e && std::atoi(e) == 0);
  }();
  return on;
}
inline int env_int_early(const char* name, int dflt) {
  const char* e = std::getenv(name);
  return e ? std::atoi(e) : dflt;
}
inline void wait_stream(stream_t s) {
  if (!spin_wait_enabled()) {
    PFV_HIP_CHECK(hipStreamSynchronize(s));
    return;
  }
  for (;;) {
    const hipError_t e = hipStreamQuery(s);
    if (e == hipSuccess) return;
    if (e != hipErrorNotReady) PFV_HIP_CHECK(e);
  }
}
inline void* pinned_scratch() {  // 4 KB of page-locked host memory per thread: target of the small reads
  thread_local void* p = nullptr;
  if (!p) PFV_HIP_CHECK(hipHostMalloc(&p, 4096, hipHostMallocDefault));
  return p;
}
#endif
inline void be_h2d(void* dst, const void* src, size_t bytes, stream_t s) {
  if (!bytes) return;
#ifdef PFV_EMULATE
  (void)s;
  std::memcpy(dst, src, bytes);
#else
  PFV_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
  wait_stream(s);
#endif
}
inline void be_d2h(void* dst, const void* src, size_t bytes, stream_t s) {
  if (!bytes) return;
#ifdef PFV_EMULATE
  (void)s;
  std::memcpy(dst, src, bytes);
#else
  if (bytes <= 4096 && spin_wait_enabled()) {
    void* p = pinned_scratch();
    PFV_HIP_CHECK(hipMemcpyAsync(p, src, bytes, hipMemcpyDeviceToHost, s));
    wait_stream(s);
    std::memcpy(dst, p, bytes);
    return;
  }
  if (bytes >= (size_t(8) << 20) && env_int_early("PFV_PINNED_COPIES", 0) != 0) {
    // Large results into pageable host memory through two page-locked staging buffers (DMA into one while
    // the other is copied out).  OFF by default: measured on the MI355X box, the runtime's own pageable
    // hipMemcpy moves the 1.66 GB system matrix in 70 ms (23 GB/s), this pipeline in 167 ms (bound by the
    // single-threaded memcpy out of the staging buffer).  Kept as a switch for hosts where that differs.
    constexpr size_t kChunk = size_t(16) << 20;
    thread_local char* stage[2] = {nullptr, nullptr};
    thread_local hipEvent_t ev[2] = {nullptr, nullptr};
    for (int i = 0; i < 2; ++i) {
      if (!stage[i]) {
        PFV_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&stage[i]), kChunk, hipHostMallocDefault));
        PFV_HIP_CHECK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
      }
    }
    const char* sp = static_cast<const char*>(src);
    char* dp = static_cast<char*>(dst);
    const size_t nchunk = (bytes + kChunk - 1) / kChunk;
    auto issue = [&](size_t k) {
      const size_t off = k * kChunk, len = std::min(kChunk, bytes - off);
      PFV_HIP_CHECK(hipMemcpyAsync(stage[k & 1], sp + off, len, hipMemcpyDeviceToHost, s));
      PFV_HIP_CHECK(hipEventRecord(ev[k & 1], s));
    };
    issue(0);
    for (size_t k = 0; k < nchunk; ++k) {
      if (k + 1 < nchunk) issue(k + 1);
      for (;;) {
        const hipError_t e = hipEventQuery(ev[k & 1]);
        if (e == hipSuccess) break;
        if (e != hipErrorNotReady) PFV_HIP_CHECK(e);
      }
      const size_t off = k * kChunk, len = std::min(kChunk, bytes - off);
      std::memcpy(dp + off, stage[k & 1], len);
    }
    return;
  }
  PFV_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s));
  wait_stream(s);
#endif
}
inline void be_d2d(void* dst, const void* src, size_t bytes, stream_t s) {
  if (!bytes) return;
#ifdef PFV_EMULATE
  (void)s;
  std::memmove(dst, src, bytes);
#else
  PFV_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s));
#endif
}
inline void be_memset(void* dst, int value, size_t bytes, stream_t s) {
  if (!bytes) return;
#ifdef PFV_EMULATE
  (void)s;
  std::memset(dst, value, bytes);
#else
  PFV_HIP_CHECK(hipMemsetAsync(dst, value, bytes, s));
#endif
}
inline void be_sync(stream_t s) {
#ifdef PFV_EMULATE
  (void)s;
#else
  wait_stream(s);
#endif
}

// growable device array
template <class T>
struct Buf {
  T* p = nullptr;
  size_t cap = 0;
  Buf() = default;
  Buf(const Buf&) = delete;
  Buf& operator=(const Buf&) = delete;
  ~Buf() { be_free(p); }
  // A buffer that has to grow a SECOND time takes an eighth of head room: sizes that follow the values (entries the
  // strength filter keeps, Galerkin products) move by a few per cent from step to step, and every growth of a
  // few-hundred-MB block is a hipMalloc of milliseconds inside the step (measured on coefficients that change every
  // step: 10 ms per step over the first steps of a run, until every buffer had seen its largest size).
  T* ensure(size_t n) {
    if (n > cap) {
      // (the head room is bounded -- 64 MiB: the sizes that move are those of the solver's level structures, not the
      // multi-GB value arrays of the discretization -- and never the reason for a failure: if the larger request does
      // not fit, the exact one is tried before giving up)
      const size_t room = std::min(cap / 8, (size_t(64) << 20) / sizeof(T));
      size_t want = cap > 0 ? std::max(n, cap + room) : n;
      be_free(p);
      p = nullptr;
      cap = 0;
      if (want > n) {
        try {
          p = static_cast<T*>(be_malloc(want * sizeof(T)));
        } catch (const Error&) {
          p = nullptr;
          want = n;
        }
      }
      if (!p) p = static_cast<T*>(be_malloc(want * sizeof(T)));
      cap = want;
    }
    return p;
  }
  void release() {
    be_free(p);
    p = nullptr;
    cap = 0;
  }
  operator T*() const { return p; }
};

// ---------------------------------------------------------------- timing
struct Timer {
#ifdef PFV_EMULATE
  std::chrono::steady_clock::time_point t0;
  void start(stream_t) { t0 = std::chrono::steady_clock::now(); }
  double stop(stream_t) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
#else
  hipEvent_t a = nullptr, b = nullptr;
  Timer() {
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
  }
  ~Timer() {
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
  }
  Timer(const Timer&) = delete;
  void start(stream_t s) { PFV_HIP_CHECK(hipEventRecord(a, s)); }
  double stop(stream_t s) {
    PFV_HIP_CHECK(hipEventRecord(b, s));
    if (spin_wait_enabled()) {
      for (;;) {
        const hipError_t e = hipEventQuery(b);
        if (e == hipSuccess) break;
        if (e != hipErrorNotReady) PFV_HIP_CHECK(e);
      }
    } else {
      PFV_HIP_CHECK(hipEventSynchronize(b));
    }
    float ms = 0.f;
    PFV_HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    return ms;
  }
  // two-step form for work on a second stream: mark() records the end without waiting,
  // elapsed_after_sync() is read once the host has to wait for that work anyway
  void mark(stream_t s) { PFV_HIP_CHECK(hipEventRecord(b, s)); }
  double elapsed_after_sync() {
    PFV_HIP_CHECK(hipEventSynchronize(b));
    float ms = 0.f;
    PFV_HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    return ms;
  }
#endif
};

#ifndef PFV_EMULATE
// fork / join of two streams with events: `side` first waits for everything enqueued on `main` so
// far; join() makes `main` wait for everything enqueued on `side` since
struct StreamFork {
  stream_t main, side;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  bool joined = false;
  StreamFork(stream_t m, stream_t sd) : main(m), side(sd) {
    PFV_HIP_CHECK(hipEventCreateWithFlags(&e0, hipEventDisableTiming));
    PFV_HIP_CHECK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
    PFV_HIP_CHECK(hipEventRecord(e0, main));
    PFV_HIP_CHECK(hipStreamWaitEvent(side, e0, 0));
  }
  void join() {
    PFV_HIP_CHECK(hipEventRecord(e1, side));
    PFV_HIP_CHECK(hipStreamWaitEvent(main, e1, 0));
    joined = true;
  }
  ~StreamFork() {
    if (!joined) (void)hipStreamSynchronize(side);  // error path: nothing may outlive the caller's buffers
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
  }
  StreamFork(const StreamFork&) = delete;
};
#endif

// ---------------------------------------------------------------- launches
struct WaveCtx {
  int64_t item;  // work item this lane group owns (node, face, cell, ...)
  char* lds;     // LDS scratch of this lane group (16-byte aligned)
  int lane;      // lane index within the group, 0 .. width-1
  int width;     // lanes per work item: 64 (one wavefront), a power of two below it, or a
                 // multi-wavefront workgroup (block_for)
  char* red;     // 256 bytes of LDS scratch for cross-wavefront reductions (block_for only)
#ifdef PFV_EMULATE
  bool lane0() const { return true; }
  void sync() const {}
  void lsync() const {}
  // index in [lo, hi) of the largest |base[i * stride]| (lowest index on ties)
  int argmax_abs(const double* base, int stride, int lo, int hi, double* best) const {
    double b = -1.0;
    int p = lo;
    for (int i = lo; i < hi; ++i) {
      const double a = std::fabs(base[(size_t)i * stride]);
      if (a > b) { b = a; p = i; }
    }
    *best = b;
    return p;
  }
#else
  __device__ bool lane0() const { return lane == 0; }
  __device__ void sync() const { __syncthreads(); }
  // Ordering of LDS traffic only, for work items that live in ONE wavefront (wave_for): LDS operations of
  // a wavefront execute in order, so all that is needed is that the compiler keeps the order -- no
  // s_barrier and, unlike __syncthreads(), no vmcnt(0) that would drain the global loads in flight.
  // Not for data handed from lane to lane through global memory.
  __device__ void lsync() const {
    if (width > 64) {
      __syncthreads();
    } else {
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  __device__ int argmax_abs(const double* base, int stride, int lo, int hi, double* best) const {
    double b = -1.0;
    int p = 0x7fffffff;
    for (int i = lo + lane; i < hi; i += width) {
      const double a = fabs(base[(size_t)i * stride]);
      if (a > b) { b = a; p = i; }
    }
    const int w = width < 64 ? width : 64;
    for (int o = w >> 1; o > 0; o >>= 1) {
      const double ob = __shfl_xor(b, o, w);
      const int op = __shfl_xor(p, o, w);
      if (ob > b || (ob == b && op < p)) { b = ob; p = op; }
    }
    if (width > 64) {  // combine the wavefronts of the workgroup through LDS
      double* rb = reinterpret_cast<double*>(red);
      int* rp = reinterpret_cast<int*>(red + 128);  // (16 wavefronts of a 1024-thread workgroup: 16 doubles, 16 ints)
      const int nw = width >> 6, wv = lane >> 6;
      __syncthreads();
      if ((lane & 63) == 0) { rb[wv] = b; rp[wv] = p; }
      __syncthreads();
      b = rb[0];
      p = rp[0];
      for (int i = 1; i < nw; ++i) {
        if (rb[i] > b || (rb[i] == b && rp[i] < p)) { b = rb[i]; p = rp[i]; }
      }
    }
    *best = b;
    return p;
  }
#endif
};

#ifndef PFV_EMULATE
template <class F>
__global__ void __launch_bounds__(256) k_parallel_for(int64_t n, F f) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) f(i);
}
// G lanes per work item, 64 / G items per single-wavefront workgroup.  A workgroup is one
// wavefront, so __syncthreads() is a scheduling no-op plus the LDS ordering fence we need.
template <int G, class F>
__global__ void __launch_bounds__(64) k_wave_for(int64_t n, size_t lds_per_item, F f) {
  extern __shared__ __attribute__((aligned(16))) char pfv_lds[];
  constexpr int per_block = 64 / G;
  const int grp = (int)threadIdx.x / G;
  for (int64_t b0 = (int64_t)blockIdx.x * per_block; b0 < n; b0 += (int64_t)gridDim.x * per_block) {
    const int64_t b = b0 + grp;
    if (b < n) {
      WaveCtx w{b, pfv_lds + (size_t)grp * lds_per_item, (int)threadIdx.x % G, G, nullptr};
      f(w);
    }
    __syncthreads();
  }
}
// The same with a register budget: at least W wavefronts per SIMD must fit (amdgpu_waves_per_eu), i.e. the compiler
// keeps the kernel within 512 / W VGPRs -- for kernels whose occupancy would otherwise fall below what their LDS
// footprint allows (the interaction-region kernel: 177 VGPRs = 2 per SIMD, its 17 KB of LDS allow 9 per CU).
template <int G, int W, class F>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(W))) k_wave_for_occ(int64_t n, size_t lds_per_item, F f) {
  extern __shared__ __attribute__((aligned(16))) char pfv_lds[];
  constexpr int per_block = 64 / G;
  const int grp = (int)threadIdx.x / G;
  for (int64_t b0 = (int64_t)blockIdx.x * per_block; b0 < n; b0 += (int64_t)gridDim.x * per_block) {
    const int64_t b = b0 + grp;
    if (b < n) {
      WaveCtx w{b, pfv_lds + (size_t)grp * lds_per_item, (int)threadIdx.x % G, G, nullptr};
      f(w);
    }
    __syncthreads();
  }
}
// Same, with the items split into 8 contiguous ranges, one per XCD (workgroup b is observed to run on
// XCD b % 8; a speed assumption only): work items that are neighbours in the item order share one L2.
// gridDim.x is a multiple of 8.
template <int G, class F>
__global__ void __launch_bounds__(64) k_wave_for_xcd(int64_t n, size_t lds_per_item, F f) {
  extern __shared__ __attribute__((aligned(16))) char pfv_lds[];
  constexpr int per_block = 64 / G;
  const int grp = (int)threadIdx.x / G;
  const int64_t nvb = (n + per_block - 1) / per_block;  // virtual blocks
  const int64_t chunk = (nvb + 7) >> 3;
  const int64_t xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
  for (int64_t vb = slot; vb < chunk; vb += per_xcd) {
    const int64_t b = (xcd * chunk + vb) * per_block + grp;
    if (b < n) {
      WaveCtx w{b, pfv_lds + (size_t)grp * lds_per_item, (int)threadIdx.x % G, G, nullptr};
      f(w);
    }
    __syncthreads();
  }
}
#endif

// One thread per index.
template <class F>
inline void parallel_for(stream_t s, int64_t n, F f) {
  if (n <= 0) return;
#ifdef PFV_EMULATE
  (void)s;
  for (int64_t i = 0; i < n; ++i) f(i);
#else
  int64_t blocks = (n + 255) / 256;
  if (blocks > 16384) blocks = 16384;  // grid-stride beyond 64 blocks per CU
  PFV_LAUNCH(HIP_KERNEL_NAME(k_parallel_for<F>), dim3((unsigned)blocks), dim3(256), 0, s, n, f);
  PFV_HIP_CHECK(hipGetLastError());
#endif
}

// G lanes (default: one 64-lane wavefront) per work item, `lds_bytes` of LDS per item.
template <int G = 64, class F>
inline void wave_for(stream_t s, int64_t n, size_t lds_bytes, F f) {
  if (n <= 0) return;
  lds_bytes = (lds_bytes + 15) & ~size_t(15);
#ifdef PFV_EMULATE
  (void)s;
  std::vector<double> lds((lds_bytes + 7) / 8 + 2);
  for (int64_t b = 0; b < n; ++b) {
    WaveCtx w{b, reinterpret_cast<char*>(lds.data()), 0, 1, nullptr};
    f(w);
  }
#else
  constexpr int per_block = 64 / G;
  const size_t block_lds = lds_bytes * per_block;
  if (block_lds > 160 * 1024) throw Error(5, "work item needs more than 160 KiB of LDS");
  int64_t blocks = (n + per_block - 1) / per_block;
  // enough resident wavefronts to fill 256 CUs several times over, then grid-stride
  if (blocks > 256 * 64) blocks = 256 * 64;
  if (block_lds > 48 * 1024) {
    PFV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wave_for<G, F>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)block_lds));
  }
  PFV_LAUNCH(HIP_KERNEL_NAME(k_wave_for<G, F>), dim3((unsigned)blocks), dim3(64), block_lds, s, n,
                     lds_bytes, f);
  PFV_HIP_CHECK(hipGetLastError());
#endif
}

#ifndef PFV_EMULATE
// wave_for with at least W wavefronts per SIMD guaranteed by the register allocation (k_wave_for_occ)
template <int G, int W, class F>
inline void wave_for_occ(stream_t s, int64_t n, size_t lds_bytes, F f) {
  if (n <= 0) return;
  lds_bytes = (lds_bytes + 15) & ~size_t(15);
  constexpr int per_block = 64 / G;
  const size_t block_lds = lds_bytes * per_block;
  if (block_lds > 160 * 1024) throw Error(5, "work item needs more than 160 KiB of LDS");
  int64_t blocks = (n + per_block - 1) / per_block;
  if (blocks > 256 * 64) blocks = 256 * 64;
  if (block_lds > 48 * 1024) {
    PFV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wave_for_occ<G, W, F>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)block_lds));
  }
  PFV_LAUNCH(HIP_KERNEL_NAME(k_wave_for_occ<G, W, F>), dim3((unsigned)blocks), dim3(64), block_lds, s, n,
                     lds_bytes, f);
  PFV_HIP_CHECK(hipGetLastError());
}
#endif

// wave_for with the lane-group width picked at run time (16 / 32 / 64): the one-item-per-wavefront
// kernels are bound by the latency of their dependent index loads, so narrower groups (more items
// in flight per wavefront) pay off whenever an item has fewer than ~64 parallel elements.
template <class F>
inline void wave_for_g(int G, stream_t s, int64_t n, size_t lds_bytes, F f) {
  if (G <= 16) wave_for<16>(s, n, lds_bytes, f);
  else if (G <= 32) wave_for<32>(s, n, lds_bytes, f);
  else wave_for<64>(s, n, lds_bytes, f);
}

// One wavefront per work item, the items dealt to the XCDs in 8 contiguous ranges (k_wave_for_xcd).
template <class F>
inline void wave_for_xcd(stream_t s, int64_t n, size_t lds_bytes, F f) {
#ifdef PFV_EMULATE
  wave_for<64>(s, n, lds_bytes, f);
#else
  if (n < 4096) {
    wave_for<64>(s, n, lds_bytes, f);
    return;
  }
  lds_bytes = (lds_bytes + 15) & ~size_t(15);
  if (lds_bytes > 160 * 1024) throw Error(5, "work item needs more than 160 KiB of LDS");
  int64_t blocks = n > 256 * 64 ? 256 * 64 : ((n + 7) & ~int64_t(7));
  if (lds_bytes > 48 * 1024) {
    PFV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wave_for_xcd<64, F>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  }
  PFV_LAUNCH(HIP_KERNEL_NAME(k_wave_for_xcd<64, F>), dim3((unsigned)blocks), dim3(64), lds_bytes, s, n,
                     lds_bytes, f);
  PFV_HIP_CHECK(hipGetLastError());
#endif
}
inline int env_int(const char* name, int dflt) {
  const char* e = std::getenv(name);
  return e ? std::atoi(e) : dflt;
}

// T threads (several wavefronts) per work item, one item per workgroup: for work items whose LDS
// footprint allows only one of them per CU anyway (MPSA interaction regions, ~140 KB).
#ifndef PFV_EMULATE
template <int T, class F>
__global__ void __launch_bounds__(T) k_block_for(int64_t n, size_t lds_bytes, F f) {
  extern __shared__ __attribute__((aligned(16))) char pfv_lds[];
  for (int64_t b = blockIdx.x; b < n; b += gridDim.x) {
    WaveCtx w{b, pfv_lds, (int)threadIdx.x, T, pfv_lds + lds_bytes};
    f(w);
    __syncthreads();
  }
}
#endif
// The same, for work items whose scratch does not fit the LDS (MPSA interaction regions of nodes where more
// than ~36 faces meet): every resident workgroup works in its own slice of a global-memory buffer instead.
// Slow (every access is an L2 round trip) but it removes the size limit; only the few oversized items of a
// launch take this path.  The 128 bytes for cross-wavefront reductions stay in LDS.
#ifndef PFV_EMULATE
template <int T, class F>
__global__ void __launch_bounds__(T) k_block_for_global(int64_t n, size_t bytes, char* scratch, F f) {
  extern __shared__ __attribute__((aligned(16))) char pfv_lds[];
  char* mine = scratch + (size_t)blockIdx.x * bytes;
  for (int64_t b = blockIdx.x; b < n; b += gridDim.x) {
    WaveCtx w{b, mine, (int)threadIdx.x, T, pfv_lds};
    f(w);
    __syncthreads();
  }
}
#endif
template <int T = 256, class F>
inline void block_for_global(stream_t s, int64_t n, size_t bytes, Buf<char>& scratch, F f) {
  if (n <= 0) return;
  bytes = (bytes + 255) & ~size_t(255);
#ifdef PFV_EMULATE
  (void)s;
  (void)scratch;
  std::vector<double> mem((bytes + 7) / 8 + 2);
  for (int64_t b = 0; b < n; ++b) {
    WaveCtx w{b, reinterpret_cast<char*>(mem.data()), 0, 1, nullptr};
    f(w);
  }
#else
  const int64_t blocks = n < 256 ? n : 256;
  char* buf = scratch.ensure((size_t)blocks * bytes);
  PFV_LAUNCH(HIP_KERNEL_NAME(k_block_for_global<T, F>), dim3((unsigned)blocks), dim3(T), 256, s, n, bytes, buf, f);
  PFV_HIP_CHECK(hipGetLastError());
#endif
}

template <int T = 256, class F>
inline void block_for(stream_t s, int64_t n, size_t lds_bytes, F f) {
  if (n <= 0) return;
  lds_bytes = (lds_bytes + 15) & ~size_t(15);
#ifdef PFV_EMULATE
  (void)s;
  std::vector<double> lds((lds_bytes + 7) / 8 + 2);
  for (int64_t b = 0; b < n; ++b) {
    WaveCtx w{b, reinterpret_cast<char*>(lds.data()), 0, 1, nullptr};
    f(w);
  }
#else
  const size_t total = lds_bytes + 256;
  if (total > 160 * 1024) throw Error(5, "work item needs more than 160 KiB of LDS");
  int64_t blocks = n < 256 * 8 ? n : 256 * 8;
  if (total > 48 * 1024) {
    PFV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_block_for<T, F>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)total));
  }
  PFV_LAUNCH(HIP_KERNEL_NAME(k_block_for<T, F>), dim3((unsigned)blocks), dim3(T), total, s, n, lds_bytes, f);
  PFV_HIP_CHECK(hipGetLastError());
#endif
}

// ---------------------------------------------------------------- sort / scan
struct Scratch {
  Buf<char> tmp;
};

// stable sort of (key, value) pairs by the low `bits` bits of the key
inline void sort_pairs(stream_t s, Scratch& sc, const uint32_t* kin, uint32_t* kout,
                       const int32_t* vin, int32_t* vout, size_t n, int bits) {
  if (n == 0) return;
#ifdef PFV_EMULATE
  (void)s; (void)sc; (void)bits;
  std::vector<size_t> idx(n);
  std::iota(idx.begin(), idx.end(), size_t(0));
  std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return kin[a] < kin[b]; });
  for (size_t i = 0; i < n; ++i) {
    kout[i] = kin[idx[i]];
    vout[i] = vin[idx[i]];
  }
#else
  size_t bytes = 0;
  PFV_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, 0, bits, s));
  sc.tmp.ensure(bytes);
  PFV_HIP_CHECK(rocprim::radix_sort_pairs(sc.tmp.p, bytes, kin, kout, vin, vout, n, 0, bits, s));
#endif
}

// out[i] = sum_{j<i} in[j] for i in [0, n]   (n+1 outputs; in[n] is not read)
template <class TI, class TO>
inline void exclusive_scan(stream_t s, Scratch& sc, const TI* in, TO* out, size_t n) {
#ifdef PFV_EMULATE
  (void)s; (void)sc;
  TO acc = 0;
  for (size_t i = 0; i < n; ++i) {
    out[i] = acc;
    acc += (TO)in[i];
  }
  out[n] = acc;
#else
  // scan n+1 entries of a transform iterator that yields 0 past the end
  auto it = rocprim::make_transform_iterator(
      rocprim::make_counting_iterator<size_t>(0),
      [in, n] __device__(size_t i) -> TO { return i < n ? (TO)in[i] : TO(0); });
  size_t bytes = 0;
  PFV_HIP_CHECK(rocprim::exclusive_scan(nullptr, bytes, it, out, TO(0), n + 1, rocprim::plus<TO>(), s));
  sc.tmp.ensure(bytes);
  PFV_HIP_CHECK(rocprim::exclusive_scan(sc.tmp.p, bytes, it, out, TO(0), n + 1, rocprim::plus<TO>(), s));
#endif
}

// A device double read back WITHOUT draining the stream: issue() enqueues the copy into page-locked memory and records
// an event behind it; take() waits for that event only -- whatever was enqueued after the copy keeps the GPU busy
// meanwhile.  (A plain read_scalar makes the host wait for everything and the GPU then idles until the next launch
// arrives: 120 us per BiCGStab iteration on the 2 M-cell system.)
struct LaggedScalar {
  bool pending = false;
#ifdef PFV_EMULATE
  const double* src = nullptr;
  void issue(const double* dptr, stream_t) { src = dptr; pending = true; }
  double take() { pending = false; return *src; }
#else
  double* host = nullptr;
  hipEvent_t ev = nullptr;
  void issue(const double* dptr, stream_t s) {
    if (!host) {
      PFV_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&host), 64, hipHostMallocDefault));
      PFV_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    PFV_HIP_CHECK(hipMemcpyAsync(host, dptr, sizeof(double), hipMemcpyDeviceToHost, s));
    PFV_HIP_CHECK(hipEventRecord(ev, s));
    pending = true;
  }
  double take() {
    if (spin_wait_enabled()) {
      for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) break;
        if (e != hipErrorNotReady) PFV_HIP_CHECK(e);
      }
    } else {
      PFV_HIP_CHECK(hipEventSynchronize(ev));
    }
    pending = false;
    return *host;
  }
  ~LaggedScalar() {
    if (host) (void)hipHostFree(host);
    if (ev) (void)hipEventDestroy(ev);
  }
#endif
  LaggedScalar() = default;
  LaggedScalar(const LaggedScalar&) = delete;
  LaggedScalar& operator=(const LaggedScalar&) = delete;
};

template <class T>
inline T read_scalar(stream_t s, const T* dptr) {
  T v;
  be_d2h(&v, dptr, sizeof(T), s);
  return v;
}

// ---------------------------------------------------------------- small device helpers
template <class T>
PFV_FN int lower_bound_idx(const T* a, int n, T key) {
  int lo = 0, hi = n;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (a[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

template <class T>
PFV_FN int upper_bound_idx(const T* a, int n, T key) {
  int lo = 0, hi = n;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (a[mid] <= key) lo = mid + 1; else hi = mid;
  }
  return lo;
}
PFV_HD inline int popcount64(unsigned long long x) { return __builtin_popcountll(x); }
PFV_HD inline void atomic_or_u64(unsigned long long* addr, unsigned long long v) {
#ifdef PFV_EMULATE
  *addr |= v;
#else
#if defined(__HIP_DEVICE_COMPILE__)
  atomicOr(addr, v);
#else
  (void)addr; (void)v;
#endif
#endif
}

// running maximum kept by one lane per work item: same-address atomics serialise at ~11 ns
// each (4 M items = 45 ms), so only touch the word when the value can raise it
PFV_HD inline void atomic_max_i32(int* addr, int v);
PFV_HD inline void track_max_i32(int* addr, int v) {
  if (v > *reinterpret_cast<volatile int*>(addr)) atomic_max_i32(addr, v);
}
PFV_HD inline void atomic_max_i32(int* addr, int v) {
#ifdef PFV_EMULATE
  if (v > *addr) *addr = v;
#else
#if defined(__HIP_DEVICE_COMPILE__)
  atomicMax(addr, v);
#else
  (void)addr; (void)v;
#endif
#endif
}
PFV_HD inline void atomic_min_i32(int* addr, int v) {
#ifdef PFV_EMULATE
  if (v < *addr) *addr = v;
#else
#if defined(__HIP_DEVICE_COMPILE__)
  atomicMin(addr, v);
#else
  (void)addr; (void)v;
#endif
#endif
}
PFV_HD inline int atomic_fetch_add_i32(int* addr, int v) {
#ifdef PFV_EMULATE
  const int old = *addr;
  *addr += v;
  return old;
#else
#if defined(__HIP_DEVICE_COMPILE__)
  return atomicAdd(addr, v);
#else
  (void)addr; (void)v;
  return 0;
#endif
#endif
}
PFV_HD inline void atomic_add_i32(int* addr, int v) {
#ifdef PFV_EMULATE
  *addr += v;
#else
#if defined(__HIP_DEVICE_COMPILE__)
  atomicAdd(addr, v);
#else
  (void)addr; (void)v;
#endif
#endif
}

}  // namespace pfv
