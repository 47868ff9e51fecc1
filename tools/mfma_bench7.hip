// Microbenchmark 7: the fp32 k-loop of the two-level kernels (v_mfma_f32_16x16x4_f32: 32 cycles an instruction, so
// barriers, LDS instructions and load latency weigh twice what they do in fp64).  Same tile shape as the kernels
// (128 x 128 per workgroup, wave tile 128 x 32, 2 workgroups per CU), random operands, steady-state clock.  Variants:
//   k16      : 16-wide k-chunks, [128][18] LDS tiles, one 4-byte LDS read per fragment (the shipped loop)
//   k32      : 32 k per iteration (two chunks per barrier pair), [128][34] tiles, 4-byte reads
//   k16 b128 : 16-wide, [128][20] tiles, ONE 16-byte LDS read per lane and operand tile row feeds the four MFMAs of the
//              chunk (lane (i, q) reads k = 4 q .. 4 q + 3; MFMA s takes element s of both operands)
//   k32 b128 : 32 k per iteration, [128][36] tiles, two 16-byte reads
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_bench7 mfma_bench7.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f4 mfma(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// source layout: [chunk][128 rows][16] floats (the kernels' chunk-major work matrices)
template <int KC, bool B128>
__global__ __launch_bounds__(256, 2) void kloop(const float* __restrict__ src, float* out, int nch, int nsets,
                                                long long* cyc) {
  constexpr int LD = B128 ? KC + 4 : KC + 2;
  constexpr int NC = KC / 16;                 // chunks per iteration
  __shared__ __attribute__((aligned(16))) float s_a[128 * LD];
  __shared__ __attribute__((aligned(16))) float s_b[128 * LD];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int set = blockIdx.x % nsets;
  const float* pa = src + (size_t)set * 2 * nch * 2048;
  const float* pb = pa + (size_t)nch * 2048;
  const int c4 = tid & 3, row = tid >> 2;     // 4 x 16 B per 16-float row, 64 rows per pass
  f4 ra[NC][2], rb[NC][2];
  auto load = [&](int it) {
#pragma unroll
    for (int h = 0; h < NC; ++h)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        ra[h][q] = *reinterpret_cast<const f4*>(pa + (size_t)(it * NC + h) * 2048 + (row + 64 * q) * 16 + 4 * c4);
        rb[h][q] = *reinterpret_cast<const f4*>(pb + (size_t)(it * NC + h) * 2048 + (row + 64 * q) * 16 + 4 * c4);
      }
  };
  auto park = [&]() {
#pragma unroll
    for (int h = 0; h < NC; ++h)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float* da = s_a + (row + 64 * q) * LD + 16 * h + 4 * c4;
        float* db = s_b + (row + 64 * q) * LD + 16 * h + 4 * c4;
        if constexpr (B128) {
          *reinterpret_cast<f4*>(da) = ra[h][q];
          *reinterpret_cast<f4*>(db) = rb[h][q];
        } else {
          *reinterpret_cast<f2*>(da) = f2{ra[h][q][0], ra[h][q][1]};
          *reinterpret_cast<f2*>(da + 2) = f2{ra[h][q][2], ra[h][q][3]};
          *reinterpret_cast<f2*>(db) = f2{rb[h][q][0], rb[h][q][1]};
          *reinterpret_cast<f2*>(db + 2) = f2{rb[h][q][2], rb[h][q][3]};
        }
      }
  };
  f4 acc[8][2];
  for (int x = 0; x < 8; ++x) for (int y = 0; y < 2; ++y) acc[x][y] = f4{0, 0, 0, 0};
  const int nit = nch / NC;
  load(0);
  const long long t0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < nit; ++it) {
    __syncthreads();
    park();
    __syncthreads();
    if (it + 1 < nit) load(it + 1);
    if constexpr (B128) {
#pragma unroll
      for (int g = 0; g < KC / 16; ++g) {
        f4 av[8], bv[2];
#pragma unroll
        for (int x = 0; x < 8; ++x) av[x] = *reinterpret_cast<const f4*>(s_a + (16 * x + l15) * LD + 16 * g + 4 * l4);
#pragma unroll
        for (int y = 0; y < 2; ++y) bv[y] = *reinterpret_cast<const f4*>(s_b + (32 * w + 16 * y + l15) * LD + 16 * g + 4 * l4);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int x = 0; x < 8; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y) acc[x][y] = mfma(av[x][s], bv[y][s], acc[x][y]);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < KC / 4; ++kk) {
        float av[8], bv[2];
#pragma unroll
        for (int x = 0; x < 8; ++x) av[x] = s_a[(16 * x + l15) * LD + 4 * kk + l4];
#pragma unroll
        for (int y = 0; y < 2; ++y) bv[y] = s_b[(32 * w + 16 * y + l15) * LD + 4 * kk + l4];
#pragma unroll
        for (int x = 0; x < 8; ++x)
#pragma unroll
          for (int y = 0; y < 2; ++y) acc[x][y] = mfma(av[x], bv[y], acc[x][y]);
      }
    }
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  float sum = 0;
  for (int x = 0; x < 8; ++x) for (int y = 0; y < 2; ++y) for (int s = 0; s < 4; ++s) sum += acc[x][y][s];
  out[(size_t)blockIdx.x * 256 + tid] = sum;
  if (tid == 0 && blockIdx.x == gridDim.x / 2) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}

__global__ void fill_random(float* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long z = i * 0x9E3779B97F4A7C15ull + 0x1234567ull;
    z ^= z >> 31; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 29;
    p[i] = (float)(((double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5) * 1e-3);
  }
}

typedef void (*kern_t)(const float*, float*, int, int, long long*);
int main(int argc, char** argv) {
  const int nch = 128, grid = 512 * 8;
  const bool zeros = argc > 1 && !strcmp(argv[1], "zeros");
  float *src, *out; long long* cyc;
  const size_t elems = (size_t)grid * 2 * nch * 2048;   // 8.6 GB
  if (hipMalloc(&src, elems * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMemset(src, 0, elems * 4);
  if (!zeros) { hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, src, elems); (void)hipDeviceSynchronize(); }
  printf("fp32, %s operands; 128 chunks of 16 per workgroup, 4096 workgroups\n", zeros ? "all-zero" : "random");
  (void)hipMalloc(&out, (size_t)grid * 256 * 4); (void)hipMalloc(&cyc, 64);
  const double flops = (double)grid * nch * 128.0 * 128 * 16 * 2;
  struct V { const char* name; kern_t k; };
  const V vs[] = {{"k16", kloop<16, false>}, {"k32", kloop<32, false>}, {"k16 b128", kloop<16, true>},
                  {"k32 b128", kloop<32, true>}};
  const int footprints[2] = {grid, 64};
  const char* fnames[2] = {"HBM stream (own rows)", "Infinity-Cache resident (64 sets)"};
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int round = 0; round < 2; ++round)
    for (int fp = 0; fp < 2; ++fp)
      for (const V& v : vs) {
        const int reps_warm = 150, reps = 60;
        for (int r = 0; r < reps_warm; ++r) hipLaunchKernelGGL(v.k, dim3(grid), dim3(256), 0, 0, src, out, nch, footprints[fp], cyc);
        (void)hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(v.k, dim3(grid), dim3(256), 0, 0, src, out, nch, footprints[fp], cyc);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        long long h[2] = {0, 0}; (void)hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
        const double ghz = h[1] > 0 ? (double)h[0] / (double)h[1] * 0.1 : 0.0;
        printf("%-36s %-10s %.3f ms  %.1f TFLOP/s  (a mid-grid workgroup: %lld cycles, clock held %.2f GHz)\n", fnames[fp], v.name,
               ms, flops / ms * 1e-9, h[0], ghz);
        fflush(stdout);
      }
  return 0;
}
