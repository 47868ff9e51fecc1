// The 16 x 16 elimination in accumulator layout, one pivot at a time (factor16_acc) against four at a time
// (factor16_acc_b4): same contract -- compared element by element on random SPD blocks -- and the cycles of either.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/factor16_probe.hip -o tools/bin/factor16_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "../ls-spa_amd/csrc/tiles.h"
using namespace lsspa;

template <bool B4>
__global__ void run(const double* Tin, double* Tout, double* Yout, long long* cyc, int reps) {
  const int lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
  d4 t0, y0;
  for (int r = 0; r < 4; ++r) {
    const int row = acc_row(l4, r);
    t0[r] = Tin[row * 16 + l15];
    y0[r] = (row == l15) ? 1.0 : 0.0;
  }
  d4 t = t0, y = y0;
  int bad = 0;
  double sink = 0.0;
  const long long c0 = __builtin_amdgcn_s_memtime();
  for (int rep = 0; rep < reps; ++rep) {
    t = t0;
    y = y0;
    t[0] += sink * 1e-300;
    if (B4) factor16_acc_b4(t, y, 1e-12, lane, bad);
    else factor16_acc<double, true>(t, y, 1e-12, lane, bad);
    sink += t[3] + y[1];
  }
  const long long c1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = c1 - c0;
  if (blockIdx.x == 0 && threadIdx.x < 64)
    for (int r = 0; r < 4; ++r) {
      Tout[acc_row(l4, r) * 16 + l15] = t[r];
      Yout[acc_row(l4, r) * 16 + l15] = y[r] + bad * 0.0;
    }
}

int main() {
  std::vector<double> A(256), T(256);
  srand(7);
  for (int i = 0; i < 256; ++i) A[i] = rand() / (double)RAND_MAX - 0.5;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double v = 0;
      for (int k = 0; k < 16; ++k) v += A[i * 16 + k] * A[j * 16 + k];
      T[i * 16 + j] = v + (i == j ? 0.5 : 0.0);
    }
  // host Cholesky for the contract: t[k][j >= k] = L[j][k] L[k][k], y[i][j <= i] = Linv[i][j] L[i][i]
  std::vector<double> L(256, 0.0), Li(256, 0.0);
  for (int j = 0; j < 16; ++j) {
    double d = T[j * 16 + j];
    for (int k = 0; k < j; ++k) d -= L[j * 16 + k] * L[j * 16 + k];
    L[j * 16 + j] = sqrt(d);
    for (int i = j + 1; i < 16; ++i) {
      double v = T[i * 16 + j];
      for (int k = 0; k < j; ++k) v -= L[i * 16 + k] * L[j * 16 + k];
      L[i * 16 + j] = v / L[j * 16 + j];
    }
  }
  for (int c = 0; c < 16; ++c)
    for (int i = 0; i < 16; ++i) {
      double v = (i == c) ? 1.0 : 0.0;
      for (int k = 0; k < i; ++k) v -= L[i * 16 + k] * Li[k * 16 + c];
      Li[i * 16 + c] = v / L[i * 16 + i];
    }
  double *dT, *dTo, *dYo;
  long long* dc;
  (void)hipMalloc(&dT, 2048); (void)hipMalloc(&dTo, 2048); (void)hipMalloc(&dYo, 2048); (void)hipMalloc(&dc, 64);
  (void)hipMemcpy(dT, T.data(), 2048, hipMemcpyHostToDevice);
  for (int b4 = 0; b4 < 2; ++b4) {
    std::vector<double> To(256), Yo(256);
    for (int warm = 0; warm < 2; ++warm) {
      if (b4) hipLaunchKernelGGL(run<true>, dim3(1), dim3(64), 0, 0, dT, dTo, dYo, dc, 64);
      else hipLaunchKernelGGL(run<false>, dim3(1), dim3(64), 0, 0, dT, dTo, dYo, dc, 64);
    }
    (void)hipDeviceSynchronize();
    long long h;
    (void)hipMemcpy(&h, dc, 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(To.data(), dTo, 2048, hipMemcpyDeviceToHost);
    (void)hipMemcpy(Yo.data(), dYo, 2048, hipMemcpyDeviceToHost);
    double et = 0, ey = 0;
    for (int k = 0; k < 16; ++k)
      for (int j = k; j < 16; ++j) et = fmax(et, fabs(To[k * 16 + j] - L[j * 16 + k] * L[k * 16 + k]));
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j <= i; ++j) ey = fmax(ey, fabs(Yo[i * 16 + j] - Li[i * 16 + j] * L[i * 16 + i]));
    printf("%s: %.0f ticks per 16 x 16 block (one wave alone); max |t - contract| %.2e, max |y - contract| %.2e\n",
           b4 ? "four pivots a step (factor16_acc_b4)" : "one pivot a step   (factor16_acc)   ", h / 64.0, et, ey);
    // 2048 workgroups of 512 threads, waves 0 and 5 working (as lat_probe does): the loaded figure
    if (b4) hipLaunchKernelGGL(run<true>, dim3(2048), dim3(64), 0, 0, dT, dTo, dYo, dc, 64);
    else hipLaunchKernelGGL(run<false>, dim3(2048), dim3(64), 0, 0, dT, dTo, dYo, dc, 64);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h, dc, 8, hipMemcpyDeviceToHost);
    printf("    2048 one-wave workgroups: %.0f ticks per block\n", h / 64.0);
  }
  return 0;
}
