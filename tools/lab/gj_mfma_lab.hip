// Lab: gj_mfma48 (porepy_amd/csrc/gj_mfma.inc) against a host Gauss-Jordan on random n x n systems.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/gj_mfma_lab.hip -o /tmp/gj_mfma_lab && /tmp/gj_mfma_lab [n [count]]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>
#include "../../porepy_amd/csrc/gj_mfma.inc"

__global__ void __launch_bounds__(64) k_lab(const double* A, double* Ainv, int* status, double* rs, int n, int count) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  double* S = reinterpret_cast<double*>(lds);
  const int ld = n;
  int32_t* ipiv = reinterpret_cast<int32_t*>(S + (n * ld > pfv::kGjmScratchDoubles ? n * ld : pfv::kGjmScratchDoubles));
  for (int m = blockIdx.x; m < count; m += gridDim.x) {
    for (int i = threadIdx.x; i < n * n; i += 64) S[i] = A[(size_t)m * n * n + i];
    __syncthreads();
    bool bad;
    const double kappa = pfv::gj_mfma48(S, n, ld, ipiv, 1e-13, bad);
    __syncthreads();
    for (int i = threadIdx.x; i < n * n; i += 64) Ainv[(size_t)m * n * n + i] = S[i];
    if (threadIdx.x == 0) status[m] = bad ? 1 : 0;
    if (threadIdx.x == 0) rs[m] = kappa;
    __syncthreads();
  }
}

static void host_inverse(const double* A, double* X, int n) {
  std::vector<double> M(A, A + n * n), I(n * n, 0.0);
  for (int i = 0; i < n; ++i) I[i * n + i] = 1.0;
  for (int k = 0; k < n; ++k) {
    int p = k;
    for (int r = k + 1; r < n; ++r) if (std::fabs(M[r * n + k]) > std::fabs(M[p * n + k])) p = r;
    for (int c = 0; c < n; ++c) { std::swap(M[k * n + c], M[p * n + c]); std::swap(I[k * n + c], I[p * n + c]); }
    const double pi = 1.0 / M[k * n + k];
    for (int c = 0; c < n; ++c) { M[k * n + c] *= pi; I[k * n + c] *= pi; }
    for (int r = 0; r < n; ++r) if (r != k) {
      const double f = M[r * n + k];
      for (int c = 0; c < n; ++c) { M[r * n + c] -= f * M[k * n + c]; I[r * n + c] -= f * I[k * n + c]; }
    }
  }
  for (int i = 0; i < n * n; ++i) X[i] = I[i];
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 36, count = argc > 2 ? atoi(argv[2]) : 2000;
  std::vector<double> A((size_t)count * n * n), X((size_t)count * n * n);
  srand(7);
  for (int m = 0; m < count; ++m)
    for (int r = 0; r < n; ++r) {
      double sum = 0.0;
      for (int c = 0; c < n; ++c) {
        // sparse-ish rows with a random dominant entry somewhere (pivoting is exercised), then row-scaled to 1-norm 1
        double v = (rand() % 100 < 30) ? (rand() / (double)RAND_MAX - 0.5) : 0.0;
        if (c == (r * 7 + m) % n) v += 1.5;
        A[((size_t)m * n + r) * n + c] = v;
        sum += std::fabs(v);
      }
      for (int c = 0; c < n; ++c) A[((size_t)m * n + r) * n + c] /= sum;
    }
  double *dA, *dX, *dR;
  int* dS;
  hipMalloc(&dA, A.size() * 8); hipMalloc(&dX, A.size() * 8); hipMalloc(&dS, count * 4); hipMalloc(&dR, count * 8);
  hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
  const size_t lds = (size_t)std::max(n * n, pfv::kGjmScratchDoubles) * 8 + 64 * 4;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k_lab<<<1024, 64, lds>>>(dA, dX, dS, dR, n, count);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k_lab<<<1024, 64, lds>>>(dA, dX, dS, dR, n, count);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
  hipMemcpy(X.data(), dX, X.size() * 8, hipMemcpyDeviceToHost);
  std::vector<int> st(count); hipMemcpy(st.data(), dS, count * 4, hipMemcpyDeviceToHost);
  std::vector<double> rs(count); hipMemcpy(rs.data(), dR, count * 8, hipMemcpyDeviceToHost);
  double worst = 0.0, worst_rs = 0.0; int nbad = 0, worst_m = -1;
  std::vector<double> H(n * n);
  for (int m = 0; m < count; ++m) {
    nbad += st[m];
    host_inverse(&A[(size_t)m * n * n], H.data(), n);
    double mx = 0.0, err = 0.0, hrs = 0.0;
    for (int r = 0; r < n; ++r) { double s = 0.0; for (int c = 0; c < n; ++c) s += std::fabs(H[r * n + c]); hrs = std::max(hrs, s); }
    for (int i = 0; i < n * n; ++i) { mx = std::max(mx, std::fabs(H[i])); err = std::max(err, std::fabs(H[i] - X[(size_t)m * n * n + i])); }
    if (err / mx > worst) { worst = err / mx; worst_m = m; }
    worst_rs = std::max(worst_rs, std::fabs(hrs - rs[m]) / hrs);
  }
  printf("n %d, %d systems: worst max-abs error / max|Ainv| %.3e (system %d), flagged singular %d, worst rel error of the row-sum norm %.3e, kernel %.3f ms (%.2f us per system per wavefront slot)\n",
         n, count, worst, worst_m, nbad, worst_rs, ms, ms * 1e3 / count * 1024);
  return worst < 1e-10 ? 0 : 2;
}
