// Feasibility probe (developer tool, NOT product code): an fp64 contraction C = A B^T carried out on the int8 matrix
// instruction (Ozaki scheme: rows scaled by powers of two and cut into S round-to-nearest 7-bit digits, the slice products
// with t + u < S summed exactly in int32 and put together in fp64), as a k-loop fed from LDS, beside the same loop on
// v_mfma_f64_16x16x4 in the same harness.  What it answers: the RATE such a loop reaches with its operands coming through
// LDS like a real k-loop's (tools/mfma_i8_bench.hip had them in registers), the ACCURACY on the chip (tools/ozaki_probe.py
// has the error model on the CPU), and what the slicing costs.  4096 x 4096 x 1024: one 64 x 64 tile of C per workgroup of
// four waves, a 32 x 32 sub-tile per wave (eight int32 accumulator tiles -- one per weight t + u -- are 128 registers).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bin/ozaki_kloop_probe tools/ozaki_kloop_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef int i4 __attribute__((ext_vector_type(4)));
typedef int i16 __attribute__((ext_vector_type(16)));
typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int R = 4096, K = 1024, S = 8;      // rows of A and of B, contraction length, slices
constexpr int LROW = 48;                      // bytes of LDS per row of a slice's 32-byte k-step (16 bytes of padding)

// ---- slicing: x = 2^e sum_t d_t 2^(-7 (t + 1)), |d_t| <= 64, e = exponent of the row's largest entry + 1 -------------------
__global__ __launch_bounds__(256) void slice_rows(const double* __restrict__ X, int8_t* __restrict__ D, int* __restrict__ E) {
  const int row = blockIdx.x;
  __shared__ double s_max[256];
  double m = 0.0;
  for (int k = threadIdx.x; k < K; k += 256) m = fmax(m, fabs(X[(size_t)row * K + k]));
  s_max[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) s_max[threadIdx.x] = fmax(s_max[threadIdx.x], s_max[threadIdx.x + s]);
    __syncthreads();
  }
  int e;
  (void)frexp(s_max[0], &e);
  e += 1;                                     // |x| 2^-e < 1/2
  if (threadIdx.x == 0) E[row] = e;
  for (int k = threadIdx.x; k < K; k += 256) {
    double a = ldexp(X[(size_t)row * K + k], -e);
#pragma unroll
    for (int t = 0; t < S; ++t) {
      a *= 128.0;
      const double d = rint(a);
      a -= d;
      D[((size_t)t * R + row) * K + k] = (int8_t)d;
    }
  }
}

// ---- the int8 k-loop ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void gemm_i8(const int8_t* __restrict__ DA, const int* __restrict__ EA,
                                                  const int8_t* __restrict__ DB, const int* __restrict__ EB,
                                                  double* __restrict__ C) {
  __shared__ __attribute__((aligned(16))) int8_t s_a[S * 64 * LROW];
  __shared__ __attribute__((aligned(16))) int8_t s_b[S * 64 * LROW];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1;
  const int r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  i16 acc[S];
#pragma unroll
  for (int t = 0; t < S; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0;
  // staging: 2 x S x 64 rows x 2 halves of 16 bytes = 2048 pieces, eight a thread (pieces 0..1023: A, the rest: B)
  i4 st[8];
  auto stage_load = [&](const int k0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = tid + 256 * (i & 3), t = c >> 7, row = (c >> 1) & 63, hh = c & 1;
      const int8_t* src = (i < 4 ? DA + ((size_t)t * R + m0 + row) * K : DB + ((size_t)t * R + n0 + row) * K) + k0 + 16 * hh;
      st[i] = *reinterpret_cast<const i4*>(src);
    }
  };
  auto stage_store = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = tid + 256 * (i & 3), t = c >> 7, row = (c >> 1) & 63, hh = c & 1;
      *reinterpret_cast<i4*>((i < 4 ? s_a : s_b) + (t * 64 + row) * LROW + 16 * hh) = st[i];
    }
  };
  stage_load(0);
  for (int k0 = 0; k0 < K; k0 += 32) {
    __syncthreads();
    stage_store();
    __syncthreads();
    if (k0 + 32 < K) stage_load(k0 + 32);
    i4 a[S], b[S];
#pragma unroll
    for (int t = 0; t < S; ++t) {
      a[t] = *reinterpret_cast<const i4*>(s_a + (t * 64 + 32 * wm + r) * LROW + 16 * h);
      b[t] = *reinterpret_cast<const i4*>(s_b + (t * 64 + 32 * wn + r) * LROW + 16 * h);
    }
#pragma unroll
    for (int t = 0; t < S; ++t)
#pragma unroll
      for (int u = 0; u + t < S; ++u) acc[t + u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[t], b[u], acc[t + u], 0, 0, 0);
  }
  // C/D map of the 32 x 32 forms: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  const int eb = EB[n0 + 32 * wn + r];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
    double v = (double)acc[S - 1][i];
#pragma unroll
    for (int t = S - 2; t >= 0; --t) v = v * (1.0 / 128.0) + (double)acc[t][i];
    const int ea = EA[m0 + 32 * wm + row];
    C[(size_t)(m0 + 32 * wm + row) * R + n0 + 32 * wn + r] = ldexp(v, ea + eb - 14);
  }
}

// ---- the same harness on the fp64 matrix instruction ----------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void gemm_f64(const double* __restrict__ A, const double* __restrict__ B,
                                                   double* __restrict__ C) {
  constexpr int LD = 18;
  __shared__ __attribute__((aligned(16))) double s_a[64 * LD];
  __shared__ __attribute__((aligned(16))) double s_b[64 * LD];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  d4 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = d4{0, 0, 0, 0};
  // staging: 2 x 64 rows x 16 k = 2048 doubles, eight a thread as four 16-byte pieces
  double2 st[4];
  auto stage_load = [&](const int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 256 * (i & 1), row = c >> 3, kk = 2 * (c & 7);
      st[i] = *reinterpret_cast<const double2*>((i < 2 ? A + (size_t)(m0 + row) * K : B + (size_t)(n0 + row) * K) + k0 + kk);
    }
  };
  auto stage_store = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 256 * (i & 1), row = c >> 3, kk = 2 * (c & 7);
      *reinterpret_cast<double2*>((i < 2 ? s_a : s_b) + row * LD + kk) = st[i];
    }
  };
  stage_load(0);
  for (int k0 = 0; k0 < K; k0 += 16) {
    __syncthreads();
    stage_store();
    __syncthreads();
    if (k0 + 16 < K) stage_load(k0 + 16);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      double a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = s_a[(32 * wm + 16 * i + l15) * LD + 4 * kk + l4];
        b[i] = s_b[(32 * wn + 16 * i + l15) * LD + 4 * kk + l4];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  // f64 C/D map: column = lane & 15, row = (lane >> 4) + 4 reg
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
        C[(size_t)(m0 + 32 * wm + 16 * i + l4 + 4 * rr) * R + n0 + 32 * wn + 16 * j + l15] = acc[i][j][rr];
}

template <typename F> static float best_ms(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  return best;
}

int main(int argc, char** argv) {
  const int spread = argc > 1 ? atoi(argv[1]) : 0;      // decades the entries of a row are spread over (0: Gaussian)
  std::vector<double> hA((size_t)R * K), hB((size_t)R * K);
  unsigned long long z = 0x243F6A8885A308D3ull;
  auto uni = [&]() { z ^= z << 13; z ^= z >> 7; z ^= z << 17; return ((z >> 11) + 0.5) * (1.0 / 9007199254740992.0); };
  auto gauss = [&]() { return sqrt(-2.0 * log(uni())) * cos(6.283185307179586 * uni()); };
  for (auto& v : hA) v = gauss() * (spread ? pow(10.0, -spread * uni()) : 1.0);
  for (auto& v : hB) v = gauss() * (spread ? pow(10.0, -spread * uni()) : 1.0);
  double *A, *B, *C8, *C64; int8_t *DA, *DB; int *EA, *EB;
  (void)hipMalloc(&A, hA.size() * 8); (void)hipMalloc(&B, hB.size() * 8);
  (void)hipMalloc(&C8, (size_t)R * R * 8); (void)hipMalloc(&C64, (size_t)R * R * 8);
  (void)hipMalloc(&DA, (size_t)S * R * K); (void)hipMalloc(&DB, (size_t)S * R * K);
  (void)hipMalloc(&EA, R * 4); (void)hipMalloc(&EB, R * 4);
  (void)hipMemcpy(A, hA.data(), hA.size() * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(B, hB.data(), hB.size() * 8, hipMemcpyHostToDevice);
  const dim3 grid(R / 64, R / 64);
  const float ms_slice = best_ms([&] {
    hipLaunchKernelGGL(slice_rows, dim3(R), dim3(256), 0, 0, A, DA, EA);
    hipLaunchKernelGGL(slice_rows, dim3(R), dim3(256), 0, 0, B, DB, EB);
  });
  const float ms_i8 = best_ms([&] { hipLaunchKernelGGL(gemm_i8, grid, dim3(256), 0, 0, DA, EA, DB, EB, C8); });
  const float ms_f64 = best_ms([&] { hipLaunchKernelGGL(gemm_f64, grid, dim3(256), 0, 0, A, B, C64); });
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
  const double flop = 2.0 * R * (double)R * K;
  printf("%s operands, %d x %d x %d, S = %d slices of 7 bits (%d int8 products per fp64 product)\n",
         spread ? "ill-scaled" : "Gaussian", R, R, K, S, S * (S + 1) / 2);
  printf("int8 k-loop (v_mfma_i32_32x32x32_i8, operands through LDS): %.3f ms = %.1f TFLOP/s fp64-equivalent\n", ms_i8, flop / ms_i8 * 1e-9);
  printf("fp64 k-loop (v_mfma_f64_16x16x4, same harness)            : %.3f ms = %.1f TFLOP/s\n", ms_f64, flop / ms_f64 * 1e-9);
  printf("slicing both operands (read 8 B, write %d B an element)      : %.3f ms = %.0f GB/s; as a share of this product %.0f %%\n", S,
         ms_slice, 2.0 * R * K * (8.0 + S) / ms_slice * 1e-6, 100.0 * ms_slice / ms_i8);
  // accuracy on a sample of entries against long double
  std::vector<double> c8((size_t)R * R), c64((size_t)R * R);
  (void)hipMemcpy(c8.data(), C8, c8.size() * 8, hipMemcpyDeviceToHost);
  (void)hipMemcpy(c64.data(), C64, c64.size() * 8, hipMemcpyDeviceToHost);
  double w8 = 0, w64 = 0;
  for (int s = 0; s < 4096; ++s) {
    const int i = (int)(uni() * R), j = (int)(uni() * R);
    long double ex = 0, sab = 0;
    for (int k = 0; k < K; ++k) {
      const long double t = (long double)hA[(size_t)i * K + k] * hB[(size_t)j * K + k];
      ex += t; sab += fabsl(t);
    }
    w8 = fmax(w8, (double)(fabsl(c8[(size_t)i * R + j] - ex) / sab));
    w64 = fmax(w64, (double)(fabsl(c64[(size_t)i * R + j] - ex) / sab));
  }
  printf("max |error| / sum|a||b| over 4096 entries: int8 slices %.2e, fp64 matrix instruction %.2e\n", w8, w64);
  return 0;
}
