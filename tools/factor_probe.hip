// Phase timing of the 64 x 64 diagonal-block factorisations (developer tool): one workgroup per matrix,
// wall_clock64 (100 MHz) stamps from workgroup 0, plus whole-kernel time at 2 workgroups per CU.
#define LSSPA_FACTOR_STAMPS 1
#include "../ls-spa_amd/csrc/k_factor.hip"
#include <cstdio>
#include <vector>
using namespace lsspa;

template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(double* A, double* Dinv, const double* diag0, int32_t* info, int p_pad,
                                                long long* stamps) {
  __shared__ __attribute__((aligned(16))) double s_a[2 * 128 * RK_LD];
  __shared__ __attribute__((aligned(16))) double s_x[128 * RK_LD];
  double* M = A + (int64_t)blockIdx.x * p_pad * p_pad;
  const long long t0 = wall_clock64();
  if (MODE == 1)
    factor_block64<double, 256>(M, p_pad, 0, Dinv + (int64_t)blockIdx.x * 4096, diag0 + (int64_t)blockIdx.x * p_pad,
                                1e-13, info, s_a, s_x, threadIdx.x);
  if (MODE == 2)
    factor_diag128<double, 256>(M, p_pad, 0, Dinv + (int64_t)blockIdx.x * 8192, diag0 + (int64_t)blockIdx.x * p_pad, 1e-13,
                                info, s_a, s_x, threadIdx.x);
  const long long t1 = wall_clock64();
  if (blockIdx.x == 0 && threadIdx.x == 0) stamps[MODE] = t1 - t0;
}

int main() {
  const int p_pad = 128, n = 512;
  std::vector<double> h((size_t)n * p_pad * p_pad, 0.0), d0((size_t)n * p_pad, 1.0);
  for (int m = 0; m < n; ++m)
    for (int r = 0; r < 64; ++r)
      for (int c = 0; c <= r; ++c)
        h[(size_t)m * p_pad * p_pad + cm_off(p_pad, r, c)] = (r == c) ? 64.0 + r : 1.0 / (1 + r - c);
  double *A, *Dinv, *diag0; int32_t* info; long long* st;
  (void)hipMalloc(&A, h.size() * 8); (void)hipMalloc(&Dinv, (size_t)n * 8192 * 8); (void)hipMalloc(&diag0, d0.size() * 8);
  (void)hipMalloc(&info, 64); (void)hipMalloc(&st, 64); (void)hipMemset(info, 0, 64);
  (void)hipMemcpy(diag0, d0.data(), d0.size() * 8, hipMemcpyHostToDevice);
  for (int m = 0; m < n; ++m)     // a full SPD 128 x 128 block for mode 2
    for (int r = 0; r < 128; ++r)
      for (int c = 0; c <= r; ++c)
        h[(size_t)m * p_pad * p_pad + cm_off(p_pad, r, c)] = (r == c) ? 128.0 + r : 1.0 / (1 + r - c);
  for (int mode = 1; mode < 3; ++mode)     // (mode 0 was the column-at-a-time elimination of round 1, removed in round 4)
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice);
      hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      (void)hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(n), dim3(256), 0, 0, A, Dinv, diag0, info, p_pad, st);
      else if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(n), dim3(256), 0, 0, A, Dinv, diag0, info, p_pad, st);
      else hipLaunchKernelGGL(probe<2>, dim3(n), dim3(256), 0, 0, A, Dinv, diag0, info, p_pad, st);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      long long s[3]; (void)hipMemcpy(s, st, 24, hipMemcpyDeviceToHost);
      std::vector<double> L(64), X(4096);
      (void)hipMemcpy(X.data(), Dinv, 4096 * 8, hipMemcpyDeviceToHost);
      printf("%s: kernel %.1f us (512 matrices, 2 per CU), workgroup 0: %.1f us, Linv[63][0] = %.6e\n",
             mode == 0 ? "columnwise" : mode == 1 ? "blocked  " : "diag128  ", ms * 1e3, s[mode] * 0.01, X[63 * 64]);
      if (mode == 2 && rep == 1) {
        long long g[32];
        (void)hipMemcpyFromSymbol(g, HIP_SYMBOL(lsspa::g_stamps), sizeof g);
        const char* nm[8] = {"c-load", "factor1", "L21 mma", "L21->lds", "store+syrk", "A22 load", "subtract", "factor2"};
        for (int i = 17; i < 24; ++i) printf("  %-10s %+6.2f us\n", nm[i - 16], (g[i] - g[i - 1]) * 0.01);
      }
      if (mode == 1 && rep == 1) {
        long long g[32];
        (void)hipMemcpyFromSymbol(g, HIP_SYMBOL(lsspa::g_stamps), sizeof g);
        // stamps of factor_block64_core: 2 + 3 kb after [pivot chain kb beside the trailing products of step kb - 1],
        // 3 + 3 kb after the panel / inverse tiles of step kb
        const int idx[11] = {0, 1, 2, 3, 5, 6, 8, 9, 11, 12, 14};
        const char* nm[11] = {"start", "load", "f0", "p0", "t0+f1", "p1", "t1+f2", "p2", "t2+f3", "p3", "store"};
        for (int i = 1; i < 11; ++i) printf("  %-6s %+6.2f us\n", nm[i], (g[idx[i]] - g[idx[i - 1]]) * 0.01);
      }
    }
  return 0;
}
