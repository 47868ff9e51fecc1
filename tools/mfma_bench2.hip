// Microbenchmark 2: (a) shader clock held under fp64 MFMA / FMA load, cycles per MFMA;
// (b) do fp64 MFMA and fp64 VALU FMA co-execute (same wave, interleaved)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NF>
__global__ __launch_bounds__(256) void mixed(double* out, long long* cyc, int iters, double a0, double b0) {
  d4 acc[8];
  double f[16];
  for (int i = 0; i < 8; ++i) acc[i] = d4{0, 0, 0, 0};
  for (int i = 0; i < 16; ++i) f[i] = i;
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NF; ++j) f[(i * NF + j) & 15] = fma(a, f[(i * NF + j) & 15], b);
    }
  }
  long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <typename F>
float timeit(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
template <int NF> void run(double* out, long long* cyc, int wg_per_cu) {
  const int iters = 20000; int grid = 256 * wg_per_cu;
  float ms = timeit([&] { hipLaunchKernelGGL(mixed<NF>, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1.0000001, 1e-9); });
  long long h[4]; (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
  double mf = (double)grid * 4 * iters * 8 * 2048.0, vf = (double)grid * 256 * iters * 8.0 * NF * 2.0;
  printf("NF=%d FMA per MFMA, %d WG/CU: %.3f ms | MFMA %.1f TF + VALU %.1f TF = %.1f TF | clock64 %lld cyc -> %.0f MHz, %.1f cyc per MFMA slot\n",
         NF, wg_per_cu, ms, mf / ms * 1e-9, vf / ms * 1e-9, (mf + vf) / ms * 1e-9, h[0], h[0] / (ms * 1e3), (double)h[0] / (iters * 8.0));
}
int main() {
  double* out; long long* cyc; (void)hipMalloc(&out, 256 * 1024 * 8 * sizeof(double)); (void)hipMalloc(&cyc, 4096 * 8);
  for (int w : {1, 2}) { run<0>(out, cyc, w); run<1>(out, cyc, w); run<2>(out, cyc, w); run<4>(out, cyc, w); run<8>(out, cyc, w); run<16>(out, cyc, w); }
  return 0;
}
