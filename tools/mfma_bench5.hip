// Microbenchmark 5: the k-loop of the two-level kernels in isolation -- global prefetch -> LDS -> barrier
// -> fragment reads -> MFMA -- with the 16x16x4 and the 4x4x4 fp64 instruction, real HBM streaming
// (every workgroup walks its own operand rows), 2 workgroups per CU.  Answers whether the 4x4x4 form's
// issue-rate advantage survives next to the memory pipeline.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
constexpr int RK_LD = 18;
template <int MODE>   // 0: 16x16x4, 1: 4x4x4, 2: 16x16x4 without global loads, 3: 4x4x4 without global loads
__global__ __launch_bounds__(256, 2) void kloop(const double* __restrict__ src, double* out, int nch, long long* cyc) {
  __shared__ __attribute__((aligned(16))) double s_a[128 * RK_LD];
  __shared__ __attribute__((aligned(16))) double s_b[128 * RK_LD];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  const double* pa = src + (size_t)blockIdx.x * 2 * nch * 2048;   // [chunk][128 rows][16]
  const double* pb = pa + (size_t)nch * 2048;
  const int c8 = tid & 7, row = tid >> 3;
  v2d ra[4], rb[4];
  auto load = [&](int c) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      ra[q] = *reinterpret_cast<const v2d*>(pa + (size_t)c * 2048 + (row + 32 * q) * 16 + 2 * c8);
      rb[q] = *reinterpret_cast<const v2d*>(pb + (size_t)c * 2048 + (row + 32 * q) * 16 + 2 * c8);
    }
  };
  constexpr bool GL = MODE < 2;
  constexpr bool K4 = (MODE & 1) != 0;
  load(0);
  d4 acc[8][2];
  double c4[8][2][4];
  for (int x = 0; x < 8; ++x) for (int y = 0; y < 2; ++y) { acc[x][y] = d4{0, 0, 0, 0}; for (int s = 0; s < 4; ++s) c4[x][y][s] = 0; }
  int bcol[4];
  for (int s = 0; s < 4; ++s) bcol[s] = 4 * ((((lane >> 2) & 3) + s) & 3) + (lane & 3);
  const long long t0 = clock64(), w0 = wall_clock64();
  for (int c = 0; c < nch; ++c) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      *reinterpret_cast<v2d*>(s_a + (row + 32 * q) * RK_LD + 2 * c8) = ra[q];
      *reinterpret_cast<v2d*>(s_b + (row + 32 * q) * RK_LD + 2 * c8) = rb[q];
    }
    __syncthreads();
    if (GL && c + 1 < nch) load(c + 1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      double av[8];
#pragma unroll
      for (int x = 0; x < 8; ++x) av[x] = s_a[(16 * x + l15) * RK_LD + 4 * kk + l4];
      if constexpr (K4) {
        double bv[2][4];
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
          for (int s = 0; s < 4; ++s) bv[y][s] = s_b[(32 * w + 16 * y + bcol[s]) * RK_LD + 4 * kk + l4];
#pragma unroll
        for (int x = 0; x < 8; ++x)
#pragma unroll
          for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int s = 0; s < 4; ++s) c4[x][y][s] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[x], bv[y][s], c4[x][y][s], 0, 0, 0);
      } else {
        double bv[2];
#pragma unroll
        for (int y = 0; y < 2; ++y) bv[y] = s_b[(32 * w + 16 * y + l15) * RK_LD + 4 * kk + l4];
#pragma unroll
        for (int x = 0; x < 8; ++x)
#pragma unroll
          for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[x], bv[y], acc[x][y], 0, 0, 0);
      }
    }
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  double sum = 0;
  for (int x = 0; x < 8; ++x) for (int y = 0; y < 2; ++y) for (int s = 0; s < 4; ++s) sum += acc[x][y][s] + c4[x][y][s];
  out[(size_t)blockIdx.x * 256 + tid] = sum;
  if (tid == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}
// the same loop with double-buffered LDS: one barrier per chunk (loads of chunk c + 1 issued before the MFMAs of chunk c,
// parked in the other buffer after them)
__global__ __launch_bounds__(256, 2) void kloop_db(const double* __restrict__ src, double* out, int nch, long long* cyc) {
  __shared__ __attribute__((aligned(16))) double s_a[2][128 * RK_LD];
  __shared__ __attribute__((aligned(16))) double s_b[2][128 * RK_LD];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  const double* pa = src + (size_t)blockIdx.x * 2 * nch * 2048;
  const double* pb = pa + (size_t)nch * 2048;
  const int c8 = tid & 7, row = tid >> 3;
  v2d ra[4], rb[4];
  auto load = [&](int c) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      ra[q] = *reinterpret_cast<const v2d*>(pa + (size_t)c * 2048 + (row + 32 * q) * 16 + 2 * c8);
      rb[q] = *reinterpret_cast<const v2d*>(pb + (size_t)c * 2048 + (row + 32 * q) * 16 + 2 * c8);
    }
  };
  auto park = [&](int b) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      *reinterpret_cast<v2d*>(s_a[b] + (row + 32 * q) * RK_LD + 2 * c8) = ra[q];
      *reinterpret_cast<v2d*>(s_b[b] + (row + 32 * q) * RK_LD + 2 * c8) = rb[q];
    }
  };
  d4 acc[8][2];
  for (int x = 0; x < 8; ++x) for (int y = 0; y < 2; ++y) acc[x][y] = d4{0, 0, 0, 0};
  load(0); park(0);
  __syncthreads();
  const long long t0 = clock64(), w0 = wall_clock64();
  for (int c = 0; c < nch; ++c) {
    const int cur = c & 1;
    if (c + 1 < nch) load(c + 1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      double av[8], bv[2];
#pragma unroll
      for (int x = 0; x < 8; ++x) av[x] = s_a[cur][(16 * x + l15) * RK_LD + 4 * kk + l4];
#pragma unroll
      for (int y = 0; y < 2; ++y) bv[y] = s_b[cur][(32 * w + 16 * y + l15) * RK_LD + 4 * kk + l4];
#pragma unroll
      for (int x = 0; x < 8; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[x], bv[y], acc[x][y], 0, 0, 0);
    }
    if (c + 1 < nch) park(cur ^ 1);
    __syncthreads();
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  double sum = 0;
  for (int x = 0; x < 8; ++x) for (int y = 0; y < 2; ++y) for (int s = 0; s < 4; ++s) sum += acc[x][y][s];
  out[(size_t)blockIdx.x * 256 + tid] = sum;
  if (tid == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}
template <typename F> float timeit(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
__global__ void fill_random(double* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long z = i * 0x9E3779B97F4A7C15ull + 0x1234567ull;
    z ^= z >> 31; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 29;
    p[i] = ((double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5) * 1e-3;
  }
}
int main(int argc, char**) {
  const int nch = 64, grid = 512 * 8;   // 8 rounds of 2 workgroups per CU
  double *src, *out; long long* cyc;
  const size_t elems = (size_t)grid * 2 * nch * 2048;   // 8.6 GB
  if (hipMalloc(&src, elems * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMemset(src, 0, elems * 8);
  if (argc > 1) {   // random operands: MFMA power, hence the sustained clock, depends on the data
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, src, elems);
    (void)hipDeviceSynchronize();
    printf("random operands\n");
  }
  (void)hipMalloc(&out, (size_t)grid * 256 * 8); (void)hipMalloc(&cyc, 64);
  const double flops = (double)grid * nch * 128.0 * 128 * 16 * 2;
  const double bytes = (double)grid * nch * 2 * 2048 * 8;
  const char* names[5] = {"16x16x4 + HBM", "4x4x4   + HBM", "16x16x4 no loads", "4x4x4   no loads", "16x16x4 + HBM, 2 LDS buffers"};
  for (int rep = 0; rep < 2; ++rep)
    for (int m = 0; m < 5; ++m) {
      float ms = 0;
      if (m == 0) ms = timeit([&] { hipLaunchKernelGGL(kloop<0>, dim3(grid), dim3(256), 0, 0, src, out, nch, cyc); });
      if (m == 1) ms = timeit([&] { hipLaunchKernelGGL(kloop<1>, dim3(grid), dim3(256), 0, 0, src, out, nch, cyc); });
      if (m == 2) ms = timeit([&] { hipLaunchKernelGGL(kloop<2>, dim3(grid), dim3(256), 0, 0, src, out, nch, cyc); });
      if (m == 3) ms = timeit([&] { hipLaunchKernelGGL(kloop<3>, dim3(grid), dim3(256), 0, 0, src, out, nch, cyc); });
      if (m == 4) ms = timeit([&] { hipLaunchKernelGGL(kloop_db, dim3(grid), dim3(256), 0, 0, src, out, nch, cyc); });
      long long h[2] = {0, 0}; (void)hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
      // shader clock held inside the loop = shader-cycle counter / 100 MHz wall counter (MI355X_MICROARCH.md, DVFS item 6)
      const double ghz = h[1] > 0 ? (double)h[0] / (double)h[1] * 0.1 : 0.0;
      printf("%-18s %.3f ms  %.1f TFLOP/s  %.2f TB/s operand stream  (wg0 loop: %lld cycles, clock held %.2f GHz -> "
             "peak at that clock %.1f TFLOP/s)\n", names[m], ms, flops / ms * 1e-9, ((m < 2 || m == 4) ? bytes : 0.0) / ms * 1e-9,
             h[0], ghz, 78.6 * ghz / 2.4);
    }
  return 0;
}
