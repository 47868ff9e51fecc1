// Microbenchmark 4: 4x4x4 fp64 MFMA with distinct operand registers (8 a x 4 b -> 32 accumulators),
// operands in registers vs re-read from LDS every step.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters, const double* in) {
  __shared__ double lds[64 * 18 * 2 + 64];
  const int l = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 64 * 18 * 2; i += 256) lds[i] = in[i % 64] + i;
  __syncthreads();
  double acc[8][4];
  for (int x = 0; x < 8; ++x) for (int s = 0; s < 4; ++s) acc[x][s] = 0;
  double av[8], bv[4];
  for (int x = 0; x < 8; ++x) av[x] = in[l] + x;
  for (int s = 0; s < 4; ++s) bv[s] = in[l] * (s + 1);
  for (int it = 0; it < iters; ++it) {
    if (MODE == 1) {
#pragma unroll
      for (int x = 0; x < 8; ++x) av[x] = lds[(x * 8 + (l & 15)) * 18 + (l >> 4) + (it & 3) * 4];
#pragma unroll
      for (int s = 0; s < 4; ++s) bv[s] = lds[64 * 18 + ((l + 4 * s) & 15) * 18 + (l >> 4) + (it & 3) * 4];
    }
#pragma unroll
    for (int x = 0; x < 8; ++x)
#pragma unroll
      for (int s = 0; s < 4; ++s) acc[x][s] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[x], bv[s], acc[x][s], 0, 0, 0);
  }
  double sum = 0;
  for (int x = 0; x < 8; ++x) for (int s = 0; s < 4; ++s) sum += acc[x][s];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}
template <typename F> float timeit(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  double *out, *in; (void)hipMalloc(&out, 256 * 1024 * 8 * 8); (void)hipMalloc(&in, 4096 * 8);
  (void)hipMemset(in, 0, 4096 * 8);
  const int iters = 10000;
  for (int w : {1, 2, 3}) {
    int grid = 256 * w;
    float ms = timeit([&] { hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, iters, in); });
    printf("regs  %d WG/CU: %.3f ms  %.1f TFLOP/s\n", w, ms, (double)grid * 4 * iters * 32 * 512.0 / ms * 1e-9);
    ms = timeit([&] { hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, iters, in); });
    printf("lds   %d WG/CU: %.3f ms  %.1f TFLOP/s\n", w, ms, (double)grid * 4 * iters * 32 * 512.0 / ms * 1e-9);
  }
  return 0;
}
