// Micro-benchmark (lab): cost of a grid-wide barrier on the MI355X against the cost of a kernel boundary.
//   hipcc --offload-arch=gfx950 -O3 tools/lab/gridsync.hip -o /tmp/gridsync && /tmp/gridsync
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;

__global__ void k_sync_loop(int iters, double* x, int n) {
  cg::grid_group g = cg::this_grid();
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
  for (int it = 0; it < iters; ++it) {
    for (int i = tid; i < n; i += nt) x[i] = x[(i + 1) % n] * 0.5 + 1.0;  // a little dependent work per phase
    g.sync();
  }
}
// hand-rolled barrier: one atomic counter, agent scope
__global__ void k_atomic_loop(int iters, double* x, int n, unsigned* counter) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
  unsigned target = 0;
  for (int it = 0; it < iters; ++it) {
    for (int i = tid; i < n; i += nt) x[i] = x[(i + 1) % n] * 0.5 + 1.0;
    __syncthreads();
    target += gridDim.x;
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {}
    }
    __syncthreads();
  }
}
__global__ void k_phase(double* x, int n) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
  for (int i = tid; i < n; i += nt) x[i] = x[(i + 1) % n] * 0.5 + 1.0;
}

int main() {
  const int iters = 2000;
  double* x; unsigned* cnt;
  hipMalloc(&x, sizeof(double) * 1 << 20); hipMemset(x, 0, sizeof(double) * 1 << 20);
  hipMalloc(&cnt, 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int blocks : {32, 64, 128, 256, 512}) {
    for (int n : {1024, 20000}) {
      float ms = 0;
      int it = iters; double* xp = x; int nn = n;
      void* args[] = {&it, &xp, &nn};
      hipLaunchCooperativeKernel((void*)k_sync_loop, dim3(blocks), dim3(256), args, 0, 0);  // warm-up
      hipDeviceSynchronize();
      hipEventRecord(a);
      hipError_t e = hipLaunchCooperativeKernel((void*)k_sync_loop, dim3(blocks), dim3(256), args, 0, 0);
      hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
      printf("cg grid.sync   blocks %3d n %5d: %.2f us per phase (%s)\n", blocks, n, 1e3 * ms / iters, hipGetErrorString(e));
      hipMemset(cnt, 0, 4);
      hipLaunchKernelGGL(k_atomic_loop, dim3(blocks), dim3(256), 0, 0, 10, x, n, cnt);
      hipDeviceSynchronize();
      hipMemset(cnt, 0, 4);
      hipEventRecord(a);
      hipLaunchKernelGGL(k_atomic_loop, dim3(blocks), dim3(256), 0, 0, iters, x, n, cnt);
      hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
      printf("atomic barrier blocks %3d n %5d: %.2f us per phase\n", blocks, n, 1e3 * ms / iters);
      hipEventRecord(a);
      for (int k = 0; k < iters; ++k) hipLaunchKernelGGL(k_phase, dim3(blocks), dim3(256), 0, 0, x, n);
      hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
      printf("kernel per phase blocks %3d n %5d: %.2f us per phase\n", blocks, n, 1e3 * ms / iters);
    }
  }
  return 0;
}
