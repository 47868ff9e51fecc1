// Microbenchmark 6: what the k-loop of the two-level kernels could still gain, measured in isolation on random data
// after >= 1 s of back-to-back launches (steady-state clock).  128 x 128 tile per workgroup, 256 threads, wave tile
// 128 (shared operand, 8 fragments) x 32 (the wave's own rows, 2 fragments), 16-wide k-chunks, 2 workgroups per CU --
// the shape of chol_panel2 / strip2.  Variants:
//   base   : single LDS buffer, two barriers per chunk, fragments read where they are used (the shipped loop)
//   pf     : + the fragments of step kk + 1 are read from LDS before the MFMAs of step kk are issued
//   db     : double-buffered LDS, one barrier per chunk, + fragment prefetch
//   dtv    : db + the wave's own operand rows go from global memory straight to registers in MFMA layout (no LDS)
//   d2     : db + global prefetch two chunks ahead (latency test)
// and where the operands come from: every workgroup its own rows (HBM stream), 64 sets shared by all workgroups
// (Infinity-Cache resident, 256 MB), one set (L2 resident).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_bench6 mfma_bench6.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
constexpr int LD = 18;   // LDS row stride of a [128][16] operand tile (doubles)

__device__ __forceinline__ d4 mfma(double a, double b, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

struct Frag { double a[8], b[2]; };

template <bool PF, bool DB, bool DTV, int DEPTH>
__global__ __launch_bounds__(256, 2) void kloop(const double* __restrict__ src, double* out, int nch, int nsets,
                                                long long* cyc) {
  constexpr int NBUF = DB ? 2 : 1;
  constexpr int SB = DTV ? 16 : 128 * LD;   // the wave's own operand needs no LDS when it is loaded straight to registers
  __shared__ __attribute__((aligned(16))) double s_a[NBUF][128 * LD];
  __shared__ __attribute__((aligned(16))) double s_b[NBUF][SB];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int set = blockIdx.x % nsets;
  const double* pa = src + (size_t)set * 2 * nch * 2048;   // [chunk][128 rows][16]
  const double* pb = pa + (size_t)nch * 2048;
  const int c8 = tid & 7, row = tid >> 3;
  v2d ra[DEPTH][4], rb[DEPTH][4];
  // DTV: lane (l15, l4) holds rows 32 w + 16 y + l15, k = 4 l4 .. 4 l4 + 3 of the chunk (2 x 16 B per y): the MFMA of
  // step kk takes element kk -- any assignment of the 16 k of a chunk to (step, k-slot) is fine as long as both
  // operands use the same one, so the shared operand is read from LDS at k = 4 l4 + kk as well
  v2d rd[DEPTH][2][2];
  auto load = [&](int c, int slot) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      ra[slot][q] = *reinterpret_cast<const v2d*>(pa + (size_t)c * 2048 + (row + 32 * q) * 16 + 2 * c8);
      if (!DTV) rb[slot][q] = *reinterpret_cast<const v2d*>(pb + (size_t)c * 2048 + (row + 32 * q) * 16 + 2 * c8);
    }
    if (DTV) {
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          rd[slot][y][h] = *reinterpret_cast<const v2d*>(pb + (size_t)c * 2048 + (32 * w + 16 * y + l15) * 16 + 4 * l4 + 2 * h);
    }
  };
  auto park = [&](int b, int slot) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      *reinterpret_cast<v2d*>(s_a[b] + (row + 32 * q) * LD + 2 * c8) = ra[slot][q];
      if (!DTV) *reinterpret_cast<v2d*>(s_b[b] + (row + 32 * q) * LD + 2 * c8) = rb[slot][q];
    }
  };
  auto frag = [&](Frag& f, int b, int kk, const v2d (&own)[2][2]) {
    const int kidx = DTV ? 4 * l4 + kk : 4 * kk + l4;
#pragma unroll
    for (int x = 0; x < 8; ++x) f.a[x] = s_a[b][(16 * x + l15) * LD + kidx];
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      if (DTV) f.b[y] = own[y][kk >> 1][kk & 1];
      else f.b[y] = s_b[b][(32 * w + 16 * y + l15) * LD + kidx];
    }
  };
  d4 acc[8][2];
  for (int x = 0; x < 8; ++x) for (int y = 0; y < 2; ++y) acc[x][y] = d4{0, 0, 0, 0};
  auto mma = [&](const Frag& f) {
#pragma unroll
    for (int x = 0; x < 8; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y) acc[x][y] = mfma(f.a[x], f.b[y], acc[x][y]);
  };
  long long t0 = 0, w0 = 0;
  if constexpr (!DB) {
    load(0, 0);
    t0 = clock64(); w0 = wall_clock64();
    for (int c = 0; c < nch; ++c) {
      __syncthreads();
      park(0, 0);
      __syncthreads();
      if (c + 1 < nch) load(c + 1, 0);
      if constexpr (PF) {
        Frag f[2];
        frag(f[0], 0, 0, rd[0]);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          if (kk < 3) frag(f[(kk + 1) & 1], 0, kk + 1, rd[0]);
          mma(f[kk & 1]);
        }
      } else {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          Frag f;
          frag(f, 0, kk, rd[0]);
          mma(f);
        }
      }
    }
  } else {
    // LDS buffer c & 1 holds chunk c; registers slot (c + 1 .. c + DEPTH) % DEPTH hold the chunks in flight
    load(0, 0);
    park(0, 0);
    v2d own[2][2];
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int h = 0; h < 2; ++h) own[y][h] = rd[0][y][h];
#pragma unroll
    for (int d = 1; d <= DEPTH; ++d) if (d < nch) load(d, d % DEPTH);
    __syncthreads();
    t0 = clock64(); w0 = wall_clock64();
    Frag f[2];
    frag(f[0], 0, 0, own);
    for (int c0 = 0; c0 < nch; c0 += DEPTH) {
#pragma unroll
      for (int dd = 0; dd < DEPTH; ++dd) {
        const int c = c0 + dd;
        const int cur = c & 1;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          if (kk < 3) frag(f[(kk + 1) & 1], cur, kk + 1, own);
          mma(f[kk & 1]);
        }
        // chunk c + 1 (in registers since DEPTH iterations) -> the other LDS buffer; its slot takes chunk c + 1 + DEPTH
        const int slot = (c + 1) % DEPTH;
        if (c + 1 < nch) {
          park(cur ^ 1, slot);
          if (DTV) {
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
              for (int h = 0; h < 2; ++h) own[y][h] = rd[slot][y][h];
          }
          if (c + 1 + DEPTH < nch) load(c + 1 + DEPTH, slot);
        }
        __syncthreads();
        frag(f[0], cur ^ 1, 0, own);
      }
    }
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  double sum = 0;
  for (int x = 0; x < 8; ++x) for (int y = 0; y < 2; ++y) for (int s = 0; s < 4; ++s) sum += acc[x][y][s];
  out[(size_t)blockIdx.x * 256 + tid] = sum;
  if (tid == 0 && blockIdx.x == gridDim.x / 2) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}

// Row-tile PAIR per workgroup: 512 threads, 256 x 128 output (eight waves, each 128 x 32 as before), the shared operand
// (128 rows x 16 k) staged once for both halves -- 21 instead of 16 flop per operand byte, one 8-wave barrier domain,
// one workgroup per CU.  Operand layout: pa = shared rows, pb = the pair's 256 own rows ([chunk][256][16]).
__global__ __launch_bounds__(512, 2) void kloop_pair(const double* __restrict__ src, double* out, int nch, int nsets,
                                                     long long* cyc) {
  __shared__ __attribute__((aligned(16))) double s_a[128 * LD];
  __shared__ __attribute__((aligned(16))) double s_b[256 * LD];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int set = blockIdx.x % nsets;
  const double* pa = src + (size_t)set * 3 * nch * 2048;   // [chunk][128][16] then [chunk][256][16]
  const double* pb = pa + (size_t)nch * 2048;
  const int c8 = tid & 7, row = tid >> 3;                  // 64 rows per pass
  v2d ra[2], rb[4];
  auto load = [&](int c) {
#pragma unroll
    for (int q = 0; q < 2; ++q) ra[q] = *reinterpret_cast<const v2d*>(pa + (size_t)c * 2048 + (row + 64 * q) * 16 + 2 * c8);
#pragma unroll
    for (int q = 0; q < 4; ++q) rb[q] = *reinterpret_cast<const v2d*>(pb + (size_t)c * 4096 + (row + 64 * q) * 16 + 2 * c8);
  };
  d4 acc[8][2];
  for (int x = 0; x < 8; ++x) for (int y = 0; y < 2; ++y) acc[x][y] = d4{0, 0, 0, 0};
  load(0);
  const long long t0 = clock64(), w0 = wall_clock64();
  for (int c = 0; c < nch; ++c) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q) *reinterpret_cast<v2d*>(s_a + (row + 64 * q) * LD + 2 * c8) = ra[q];
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<v2d*>(s_b + (row + 64 * q) * LD + 2 * c8) = rb[q];
    __syncthreads();
    if (c + 1 < nch) load(c + 1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      double av[8], bv[2];
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
  const long long t1 = clock64(), w1 = wall_clock64();
  double sum = 0;
  for (int x = 0; x < 8; ++x) for (int y = 0; y < 2; ++y) for (int s2 = 0; s2 < 4; ++s2) sum += acc[x][y][s2];
  out[(size_t)blockIdx.x * 512 + tid] = sum;
  if (tid == 0 && blockIdx.x == gridDim.x / 2) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}

// Half-height tile: 256 threads, 64 own rows x 128 shared rows (wave tile 128 x 16, 64 accumulator registers), four
// workgroups per CU -- twice the waves to hide the epilogues behind, 10.7 instead of 16 flop per operand byte.
__global__ __launch_bounds__(256, 4) void kloop_half(const double* __restrict__ src, double* out, int nch, int nsets,
                                                     long long* cyc) {
  __shared__ __attribute__((aligned(16))) double s_a[128 * LD];
  __shared__ __attribute__((aligned(16))) double s_b[64 * LD];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int set = blockIdx.x % nsets;
  const double* pa = src + (size_t)set * 2 * nch * 2048;
  const double* pb = pa + (size_t)nch * 2048;
  const int c8 = tid & 7, row = tid >> 3;
  v2d ra[4], rb[2];
  auto load = [&](int c) {
#pragma unroll
    for (int q = 0; q < 4; ++q) ra[q] = *reinterpret_cast<const v2d*>(pa + (size_t)c * 2048 + (row + 32 * q) * 16 + 2 * c8);
#pragma unroll
    for (int q = 0; q < 2; ++q) rb[q] = *reinterpret_cast<const v2d*>(pb + (size_t)c * 2048 + (row + 32 * q) * 16 + 2 * c8);
  };
  d4 acc[8];
  for (int x = 0; x < 8; ++x) acc[x] = d4{0, 0, 0, 0};
  load(0);
  const long long t0 = clock64(), w0 = wall_clock64();
  for (int c = 0; c < nch; ++c) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<v2d*>(s_a + (row + 32 * q) * LD + 2 * c8) = ra[q];
#pragma unroll
    for (int q = 0; q < 2; ++q) *reinterpret_cast<v2d*>(s_b + (row + 32 * q) * LD + 2 * c8) = rb[q];
    __syncthreads();
    if (c + 1 < nch) load(c + 1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      double av[8];
#pragma unroll
      for (int x = 0; x < 8; ++x) av[x] = s_a[(16 * x + l15) * LD + 4 * kk + l4];
      const double bv = s_b[(16 * w + l15) * LD + 4 * kk + l4];
#pragma unroll
      for (int x = 0; x < 8; ++x) acc[x] = mfma(av[x], bv, acc[x]);
    }
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  double sum = 0;
  for (int x = 0; x < 8; ++x) for (int s2 = 0; s2 < 4; ++s2) sum += acc[x][s2];
  out[(size_t)blockIdx.x * 256 + tid] = sum;
  if (tid == 0 && blockIdx.x == gridDim.x / 2) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}

__global__ void fill_random(double* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long z = i * 0x9E3779B97F4A7C15ull + 0x1234567ull;
    z ^= z >> 31; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 29;
    p[i] = ((double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5) * 1e-3;
  }
}

typedef void (*kern_t)(const double*, double*, int, int, long long*);
int main(int argc, char** argv) {
  const int nch = 64, grid = 512 * 8;
  const bool zeros = argc > 1 && !strcmp(argv[1], "zeros");
  double *src, *out; long long* cyc;
  const size_t elems = (size_t)grid * 2 * nch * 2048;   // 8.6 GB
  if (hipMalloc(&src, elems * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMemset(src, 0, elems * 8);
  if (!zeros) { hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, src, elems); (void)hipDeviceSynchronize(); }
  printf("%s operands; 64 chunks of 16 per workgroup, 4096 workgroups\n", zeros ? "all-zero" : "random");
  (void)hipMalloc(&out, (size_t)grid * 256 * 8); (void)hipMalloc(&cyc, 64);
  const double flops = (double)grid * nch * 128.0 * 128 * 16 * 2;
  struct V { const char* name; kern_t k; };
  const V vs[] = {{"base", kloop<false, false, false, 1>}, {"pf", kloop<true, false, false, 1>},
                  {"db+pf", kloop<true, true, false, 1>}, {"db+pf+dtv", kloop<true, true, true, 1>},
                  {"db+pf depth2", kloop<true, true, false, 2>}, {"db+pf+dtv depth2", kloop<true, true, true, 2>}};
  const int footprints[3] = {grid, 64, 1};
  const char* fnames[3] = {"HBM stream (own rows)", "Infinity-Cache resident (64 sets)", "L2 resident (1 set)"};
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int round = 0; round < 2; ++round)
    for (int fp = 0; fp < 3; ++fp)
      for (const V& v : vs) {
        const int reps_warm = 150, reps = 60;     // ~0.4 s + ~0.15 s of back-to-back launches per line
        for (int r = 0; r < reps_warm; ++r) hipLaunchKernelGGL(v.k, dim3(grid), dim3(256), 0, 0, src, out, nch, footprints[fp], cyc);
        (void)hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(v.k, dim3(grid), dim3(256), 0, 0, src, out, nch, footprints[fp], cyc);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        long long h[2] = {0, 0}; (void)hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
        const double ghz = h[1] > 0 ? (double)h[0] / (double)h[1] * 0.1 : 0.0;
        printf("%-36s %-18s %.3f ms  %.1f TFLOP/s  (a mid-grid workgroup: %lld cycles, clock held %.2f GHz)\n", fnames[fp], v.name,
               ms, flops / ms * 1e-9, h[0], ghz);
        fflush(stdout);
      }
  // the row-tile pair: half as many workgroups for the same flops; the source holds 3 x nch chunks per workgroup
  for (int round = 0; round < 2; ++round)
    for (int fp = 0; fp < 3; ++fp) {
      const int g2 = grid / 2, sets = footprints[fp] == grid ? g2 : footprints[fp];
      for (int r = 0; r < 150; ++r) hipLaunchKernelGGL(kloop_pair, dim3(g2), dim3(512), 0, 0, src, out, nch, sets, cyc);
      (void)hipEventRecord(e0);
      for (int r = 0; r < 60; ++r) hipLaunchKernelGGL(kloop_pair, dim3(g2), dim3(512), 0, 0, src, out, nch, sets, cyc);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 60;
      long long h[2] = {0, 0}; (void)hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
      const double ghz = h[1] > 0 ? (double)h[0] / (double)h[1] * 0.1 : 0.0;
      printf("%-36s %-18s %.3f ms  %.1f TFLOP/s  (a mid-grid workgroup: %lld cycles, clock held %.2f GHz)\n", fnames[fp],
             "pair 256x128 / 512", ms, flops / ms * 1e-9, h[0], ghz);
      fflush(stdout);
    }
  // the half-height tile: the same number of workgroups, half the flops each
  for (int round = 0; round < 2; ++round)
    for (int fp = 0; fp < 3; ++fp) {
      for (int r = 0; r < 150; ++r) hipLaunchKernelGGL(kloop_half, dim3(grid), dim3(256), 0, 0, src, out, nch, footprints[fp], cyc);
      (void)hipEventRecord(e0);
      for (int r = 0; r < 60; ++r) hipLaunchKernelGGL(kloop_half, dim3(grid), dim3(256), 0, 0, src, out, nch, footprints[fp], cyc);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 60;
      long long h[2] = {0, 0}; (void)hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
      const double ghz = h[1] > 0 ? (double)h[0] / (double)h[1] * 0.1 : 0.0;
      printf("%-36s %-18s %.3f ms  %.1f TFLOP/s  (a mid-grid workgroup: %lld cycles, clock held %.2f GHz)\n", fnames[fp],
             "half 64x128 / 256 x4", ms, 0.5 * flops / ms * 1e-9, h[0], ghz);
      fflush(stdout);
    }
  return 0;
}
