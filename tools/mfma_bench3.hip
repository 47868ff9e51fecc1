// Microbenchmark 3: v_mfma_f64_4x4x4_4b_f64 vs v_mfma_f64_16x16x4_f64 issue rates on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k4(double* out, long long* cyc, int iters, double a0, double b0) {
  double acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = 0;
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ __launch_bounds__(256) void k16(double* out, long long* cyc, int iters, double a0, double b0) {
  d4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <typename F> float timeit(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  double* out; long long* cyc; (void)hipMalloc(&out, 256 * 1024 * 8 * sizeof(double)); (void)hipMalloc(&cyc, 4096 * 8);
  const int iters = 20000;
  for (int w : {1, 2, 4}) {
    int grid = 256 * w; long long h[1];
    float ms = timeit([&] { hipLaunchKernelGGL(k4, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1.0000001, 1e-9); });
    (void)hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
    printf("4x4x4_4b  %d WG/CU: %.3f ms  %.1f TFLOP/s  %.1f cyc per MFMA per wave, clock %.0f MHz\n", w, ms,
           (double)grid * 4 * iters * 16 * 512.0 / ms * 1e-9, (double)h[0] / (iters * 16.0), h[0] / (ms * 1e3));
    ms = timeit([&] { hipLaunchKernelGGL(k16, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1.0000001, 1e-9); });
    (void)hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
    printf("16x16x4   %d WG/CU: %.3f ms  %.1f TFLOP/s  %.1f cyc per MFMA per wave, clock %.0f MHz\n", w, ms,
           (double)grid * 4 * iters * 8 * 2048.0 / ms * 1e-9, (double)h[0] / (iters * 8.0), h[0] / (ms * 1e3));
  }
  return 0;
}
