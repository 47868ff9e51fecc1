// Issue rate of the int8 matrix instructions on gfx950 (register-resident operands, no memory traffic): the ceiling an
// fp64-by-int8 (Ozaki-type) contraction would start from.  tools/ozaki_probe.py has the error model (how many int8
// products an fp64 product costs at a given accuracy).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_i8_bench tools/mfma_i8_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i4 __attribute__((ext_vector_type(4)));
typedef int i16 __attribute__((ext_vector_type(16)));
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k_i8_16(int* out, long long* cyc, int iters, int seed) {
  i4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = i4{0, 0, 0, 0};
  i4 a = {seed + (int)threadIdx.x, seed * 3, seed * 5, seed * 7}, b = {seed * 11, seed * 13, (int)threadIdx.x, seed};
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = clock64();
  int s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ __launch_bounds__(512) void k_i8_32(int* out, long long* cyc, int iters, int seed) {
  i16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0;
  i4 a = {seed + (int)threadIdx.x, seed * 3, seed * 5, seed * 7}, b = {seed * 11, seed * 13, (int)threadIdx.x, seed};
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = clock64();
  int s = 0;
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ __launch_bounds__(512) void k_f64(double* out, long long* cyc, int iters, double a0, double b0) {
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
  f(); f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
  int* out; double* outd; long long* cyc;
  (void)hipMalloc(&out, 256 * 2048 * sizeof(int)); (void)hipMalloc(&outd, 256 * 2048 * sizeof(double)); (void)hipMalloc(&cyc, 4096 * 8);
  const int iters = 40000;
  for (int w : {1, 2}) {
    const int grid = 256 * w; long long h[1];
    float ms = timeit([&] { hipLaunchKernelGGL(k_i8_16, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 3); });
    (void)hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
    printf("i32_16x16x64_i8 %d WG/CU: %.3f ms  %.0f TOP/s  %.1f cyc per instruction per wave\n", w, ms,
           (double)grid * 4 * iters * 8 * 32768.0 / ms * 1e-9, (double)h[0] / (iters * 8.0));
    ms = timeit([&] { hipLaunchKernelGGL(k_i8_32, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 3); });
    (void)hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
    printf("i32_32x32x32_i8 %d WG/CU: %.3f ms  %.0f TOP/s  %.1f cyc per instruction per wave\n", w, ms,
           (double)grid * 4 * iters * 4 * 65536.0 / ms * 1e-9, (double)h[0] / (iters * 4.0));
    ms = timeit([&] { hipLaunchKernelGGL(k_f64, dim3(grid), dim3(256), 0, 0, outd, cyc, iters / 4, 1.0000001, 1e-9); });
    (void)hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
    printf("f64_16x16x4     %d WG/CU: %.3f ms  %.1f TFLOP/s  %.1f cyc per instruction per wave\n", w, ms,
           (double)grid * 4 * (iters / 4) * 8 * 2048.0 / ms * 1e-9, (double)h[0] / (iters / 4 * 8.0));
  }
  return 0;
}
