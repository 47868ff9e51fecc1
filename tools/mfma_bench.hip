// Microbenchmark: sustained rate of v_mfma_f64_16x16x4_f64 and of v_fma_f64 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(double* out, int iters, double a0, double b0) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void fma_loop(double* out, int iters, double a0, double b0) {
  double acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = i;
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = fma(a, acc[i], b);
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F>
float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  double* out; hipMalloc(&out, 256 * 2048 * 8 * sizeof(double));
  const int iters = 20000;
  for (int wg_per_cu : {1, 2, 4}) {
    int grid = 256 * wg_per_cu;
    float ms = timeit([&] { hipLaunchKernelGGL(mfma_loop<8>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1.0); });
    double fl = (double)grid * 4 * iters * 8 * 2048.0;
    printf("mfma_f64 16x16x4, 8 acc, %d WG/CU (256 thr): %.3f ms  %.1f TFLOP/s\n", wg_per_cu, ms, fl / ms * 1e-9);
  }
  {
    float ms = timeit([&] { hipLaunchKernelGGL(mfma_loop<1>, dim3(256), dim3(256), 0, 0, out, iters, 1.0, 1.0); });
    double fl = 256.0 * 4 * iters * 1 * 2048.0;
    printf("mfma_f64 dependent chain (1 acc), 1 WG/CU: %.3f ms  %.1f TFLOP/s -> %.1f cycles/MFMA @2.4GHz\n", ms, fl / ms * 1e-9, ms * 1e-3 * 2.4e9 / iters);
  }
  for (int wg_per_cu : {1, 2, 4}) {
    int grid = 256 * wg_per_cu;
    float ms = timeit([&] { hipLaunchKernelGGL(fma_loop, dim3(grid), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9); });
    double fl = (double)grid * 256 * iters * 16 * 2.0;
    printf("v_fma_f64, 16 chains, %d WG/CU: %.3f ms  %.1f TFLOP/s\n", wg_per_cu, ms, fl / ms * 1e-9);
  }
  return 0;
}
