// What the fp64 matrix pipes of the WHOLE chip sustain (developer tool).  A register-resident loop of nothing but
// v_mfma_f64_16x16x4 (eight independent accumulators, no memory, no LDS) is launched on 8 ... 512 workgroups of four
// or eight waves -- one workgroup per CU, one or two waves per SIMD -- for bursts of different length.  Each workgroup reports
// its own shader cycles (s_memtime) and wall-clock ticks (100 MHz), so the line shows both the clock the burst ran at
// and the cycles one instruction took on its SIMD: a chip-wide limit shows as cycles per instruction growing with the
// number of busy CUs while the clock stays where it was.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_cap_probe tools/mfma_cap_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

// GAP: s_nop cycles (x 16) after each group of eight instructions: the same instruction stream at a lower duty
template <int GAP>
__global__ __launch_bounds__(512) void k_f64(double* out, long long* cyc, long long* wall, int iters, double a0, double b0) {
  d4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = d4{0, 0, 0, 0};
  const double a = a0 + threadIdx.x * 1e-9, b = b0 + threadIdx.x * 3e-9;
  const long long w0 = wall_clock64(), t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < GAP; ++g) asm volatile("s_nop 15");
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; wall[blockIdx.x] = w1 - w0; }
}

// the same loop fed with operands whose mantissa bits differ from instruction to instruction (what real data does to
// the multiplier arrays: switching power depends on it)
__global__ __launch_bounds__(512) void k_f64_rand(double* out, long long* cyc, long long* wall, int iters, unsigned seed) {
  d4 acc[8];
  double a[8], b[8];
  unsigned long long x = seed * 0x9E3779B97F4A7C15ull + threadIdx.x * 0xD1B54A32D192ED03ull + blockIdx.x;
  for (int i = 0; i < 8; ++i) {
    acc[i] = d4{0, 0, 0, 0};
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    a[i] = __longlong_as_double((x & 0x800FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);   // +-[1, 2), random mantissa
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    b[i] = __longlong_as_double((x & 0x800FFFFFFFFFFFFFull) | 0x3F50000000000000ull) ;  // +-2^-10 [1, 2)
  }
  const long long w0 = wall_clock64(), t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[(i + 3) & 7], acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[(i + 5) & 7], b[i], acc[i], 0, 0, 0);
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; wall[blockIdx.x] = w1 - w0; }
}

static void run_rand(int grid, int block, int iters, double* out, long long* cyc, long long* wall) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k_f64_rand, dim3(grid), dim3(block), 0, 0, out, cyc, wall, iters, 7u);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k_f64_rand, dim3(grid), dim3(block), 0, 0, out, cyc, wall, iters, 7u);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> c(grid), w(grid);
  (void)hipMemcpy(c.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
  (void)hipMemcpy(w.data(), wall, grid * 8, hipMemcpyDeviceToHost);
  double cs = 0, ws = 0; long long cmax = 0;
  for (int i = 0; i < grid; ++i) { cs += c[i]; ws += w[i]; cmax = std::max(cmax, c[i]); }
  const double n_inst = (double)iters * 16;
  printf("random operands  %3d workgroups of %d waves  %6d x 16 instr  %8.3f ms  %6.1f TFLOP/s  clock %.3f GHz  %6.1f cycles "
         "per instruction of a wave (slowest workgroup %6.1f)\n",
         grid, block / 64, iters, ms, (double)grid * (block / 64) * n_inst * 2048.0 / ms * 1e-9, cs / ws * 0.1,
         cs / grid / n_inst, cmax / n_inst);
}

// What vector fp64 arithmetic costs beside matrix fp64 arithmetic on the SAME SIMD: workgroups of eight waves, waves 0-3
// (one per SIMD) run the matrix loop (or idle, mode 0), waves 4-7 a loop of v_fma_f64 -- eight independent chains
// (ILP 8) or one dependent chain (ILP 1) -- and report their own cycles per instruction.
template <int ILP>
__global__ __launch_bounds__(512) void k_beside(double* out, long long* cyc, int iters, int matrix_on, double a0) {
  const int w = threadIdx.x >> 6;
  if (w < 4) {
    if (!matrix_on) return;
    d4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = d4{0, 0, 0, 0};
    const double a = a0 + threadIdx.x * 1e-9, b = 1e-9 + threadIdx.x * 3e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    return;
  }
  double x[8];
  for (int i = 0; i < 8; ++i) x[i] = a0 + i + threadIdx.x * 1e-9;
  const double m = 0.999999999, c = 1e-12;
  // the vector waves run a quarter as many groups: they should end well inside the matrix waves' run
  const long long t0 = clock64();
  for (int it = 0; it < iters / 4; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (ILP == 8) x[i] = fma(x[i], m, c);
      else x[0] = fma(x[0], m, c);
    }
  }
  const long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && w == 4) cyc[blockIdx.x] = t1 - t0;
}

template <int ILP>
static void run_beside(int grid, int iters, int matrix_on, double* out, long long* cyc) {
  hipLaunchKernelGGL(k_beside<ILP>, dim3(grid), dim3(512), 0, 0, out, cyc, iters, matrix_on, 1.0000001);
  (void)hipDeviceSynchronize();
  hipLaunchKernelGGL(k_beside<ILP>, dim3(grid), dim3(512), 0, 0, out, cyc, iters, matrix_on, 1.0000001);
  (void)hipDeviceSynchronize();
  std::vector<long long> c(grid);
  (void)hipMemcpy(c.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
  double cs = 0;
  for (int i = 0; i < grid; ++i) cs += c[i];
  printf("v_fma_f64, %s, beside %s: %6.1f cycles per instruction of a wave\n", ILP == 8 ? "eight independent chains" : "one dependent chain",
         matrix_on ? "a wave of fp64 matrix instructions on its SIMD" : "nothing", cs / grid / ((double)(iters / 4) * 8));
}

template <int GAP>
static void run(int grid, int block, int iters, double* out, long long* cyc, long long* wall) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k_f64<GAP>, dim3(grid), dim3(block), 0, 0, out, cyc, wall, iters, 1.0000001, 1e-9);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k_f64<GAP>, dim3(grid), dim3(block), 0, 0, out, cyc, wall, iters, 1.0000001, 1e-9);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> c(grid), w(grid);
  (void)hipMemcpy(c.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
  (void)hipMemcpy(w.data(), wall, grid * 8, hipMemcpyDeviceToHost);
  double cs = 0, ws = 0; long long cmax = 0;
  for (int i = 0; i < grid; ++i) { cs += c[i]; ws += w[i]; cmax = std::max(cmax, c[i]); }
  const double n_inst = (double)iters * 8;
  printf("gap %2d  %3d workgroups of %d waves  %6d x 8 instr  %8.3f ms  %6.1f TFLOP/s  clock %.3f GHz  %6.1f cycles per instruction "
         "of a wave (slowest workgroup %6.1f)  %5.1f GFLOP/s per workgroup-us\n",
         GAP * 16, grid, block / 64, iters, ms, (double)grid * (block / 64) * n_inst * 2048.0 / ms * 1e-9, cs / ws * 0.1, cs / grid / n_inst,
         cmax / n_inst, (block / 64) * n_inst * 2048.0 / (ws / grid * 0.01) * 1e-3);
}

int main() {
  double* out; long long *cyc, *wall;
  (void)hipMalloc(&out, 256 * 2048 * sizeof(double)); (void)hipMalloc(&cyc, 4096 * 8); (void)hipMalloc(&wall, 4096 * 8);
  // warm the chip: two long full-chip bursts
  run<0>(512, 256, 40000, out, cyc, wall);
  printf("-- burst of ~5 ms at one wave per SIMD, growing number of busy CUs\n");
  for (int grid : {8, 32, 64, 128, 192, 256}) run<0>(grid, 256, 20000, out, cyc, wall);
  printf("-- two waves per SIMD (workgroups of eight waves, one per CU)\n");
  for (int grid : {8, 32, 64, 128, 192, 256}) run<0>(grid, 512, 10000, out, cyc, wall);
  printf("-- burst length, full chip, two waves per SIMD\n");
  for (int iters : {250, 1000, 4000, 16000, 64000}) run<0>(256, 512, iters, out, cyc, wall);
  printf("-- the same stream at a lower duty (s_nop gaps), full chip, two waves per SIMD\n");
  run<4>(256, 512, 10000, out, cyc, wall);
  run<16>(256, 512, 10000, out, cyc, wall);
  run<32>(256, 512, 5000, out, cyc, wall);
  printf("-- vector fp64 beside matrix fp64 on one SIMD (256 workgroups of eight waves)\n");
  run_beside<8>(256, 20000, 0, out, cyc);
  run_beside<8>(256, 20000, 1, out, cyc);
  run_beside<1>(256, 20000, 0, out, cyc);
  run_beside<1>(256, 20000, 1, out, cyc);
  printf("-- operands with random mantissas, two waves per SIMD, growing number of busy CUs, then burst length\n");
  for (int grid : {8, 64, 128, 192, 256}) run_rand(grid, 512, 5000, out, cyc, wall);
  for (int iters : {500, 2000, 20000, 100000}) run_rand(256, 512, iters, out, cyc, wall);
  printf("-- constant operands again, straight after\n");
  run<0>(256, 512, 10000, out, cyc, wall);
  return 0;
}
