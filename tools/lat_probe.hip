// Dependent-chain latencies of the instructions on the 16 x 16 elimination's critical path (one wave, alone on its CU;
// cycles by s_memtime).   hipcc --offload-arch=gfx950 -O3 tools/lat_probe.hip -o tools/bin/lat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe(double* out, long long* cyc, double seed) {
  const int lane = threadIdx.x;
  double x = seed + lane * 1e-3;
  long long t0, t1;
  constexpr int N = 256;
  // (a) dependent v_rcp_f64
  t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 16
  for (int i = 0; i < N; ++i) x = __builtin_amdgcn_rcp(x);
  asm volatile("" : "+v"(x));
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[0] = (t1 - t0);
  // (b) dependent v_fma_f64
  t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 16
  for (int i = 0; i < N; ++i) x = fma(x, 0.999999, 1e-9);
  asm volatile("" : "+v"(x));
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[1] = (t1 - t0);
  // (c) dependent MFMA f64 16x16x4 (accumulator chain)
  d4 acc = {x, x, x, x};
  t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 16
  for (int i = 0; i < N; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(1e-3, 1e-3, acc, 0, 0, 0);
  asm volatile("" : "+v"(acc));
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[2] = (t1 - t0);
  // (d) MFMA whose A operand depends on the previous MFMA's result through a VALU op
  t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 16
  for (int i = 0; i < N; ++i) {
    const double a = acc[0] * 1e-3;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, 1e-3, acc, 0, 0, 0);
  }
  asm volatile("" : "+v"(acc));
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[3] = (t1 - t0);
  // (e) the same through readlane -> scalar -> VALU
  t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 16
  for (int i = 0; i < N; ++i) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(acc[0]), 5), hi = __builtin_amdgcn_readlane(__double2hiint(acc[0]), 5);
    const double a = __hiloint2double(hi, lo) * 1e-3;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, 1e-3, acc, 0, 0, 0);
  }
  asm volatile("" : "+v"(acc));
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[4] = (t1 - t0);
  // (f) readlane -> rcp + two Newton steps -> mul -> MFMA -> second independent MFMA (the pivot step)
  d4 acc2 = acc;
  t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 8
  for (int i = 0; i < N; ++i) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(acc[0]), 5), hi = __builtin_amdgcn_readlane(__double2hiint(acc[0]), 5);
    const double d = __hiloint2double(hi, lo);
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    const double a = -acc[1] * r * 1e-9;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc[1], acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc2[1], acc2, 0, 0, 0);
  }
  asm volatile("" : "+v"(acc), "+v"(acc2));
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[5] = (t1 - t0);
  // (g) the 4x4x4 f64 MFMA (4 blocks): latency of the small shape
  t0 = __builtin_amdgcn_s_memtime();
  double s = acc[0];
#pragma unroll 16
  for (int i = 0; i < N; ++i) s = __builtin_amdgcn_mfma_f64_4x4x4f64(1e-3, 1e-3, s, 0, 0, 0);
  asm volatile("" : "+v"(s));
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[6] = (t1 - t0);
  // (h) dependent ds_bpermute chain
  t0 = __builtin_amdgcn_s_memtime();
  int v = lane;
#pragma unroll 16
  for (int i = 0; i < N; ++i) v = __builtin_amdgcn_ds_bpermute(((v + 1) & 63) * 4, v);
  asm volatile("" : "+v"(v));
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[7] = (t1 - t0);
  // (i) one f64 MFMA (accumulator chain) + eight INDEPENDENT-of-it dependent v_fma_f64 per iteration: 64 if the vector
  //     pipe's f64 operations run beside the matrix instruction, 64 + 8 x 7.4 if they queue behind it
  double z = x;
  t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 4
  for (int i = 0; i < N; ++i) {
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(1e-3, 1e-3, acc, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 8; ++j) z = fma(z, 0.999999, 1e-9);
  }
  asm volatile("" : "+v"(acc), "+v"(z));
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[8] = (t1 - t0);
  // (j) the same with eight dependent v_fma_f32
  float zf = (float)x;
  t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 4
  for (int i = 0; i < N; ++i) {
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(1e-3, 1e-3, acc, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 8; ++j) zf = fmaf(zf, 0.999999f, 1e-9f);
  }
  asm volatile("" : "+v"(acc), "+v"(zf));
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[9] = (t1 - t0);
  // (k) two independent MFMA chains + the eight f64 fma: does the second instruction hold the vector operations back?
  t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 4
  for (int i = 0; i < N; ++i) {
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(1e-3, 1e-3, acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(1e-3, 1e-3, acc2, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 8; ++j) z = fma(z, 0.999999, 1e-9);
  }
  asm volatile("" : "+v"(acc), "+v"(acc2), "+v"(z));
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[10] = (t1 - t0);
  // (l) v_rcp_f64 + 4 fma beside one MFMA
  t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 4
  for (int i = 0; i < N; ++i) {
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(1e-3, 1e-3, acc, 0, 0, 0);
    double r = __builtin_amdgcn_rcp(z);
    r = fma(fma(-z, r, 1.0), r, r);
    r = fma(fma(-z, r, 1.0), r, r);
    z = r + 1.5;
  }
  asm volatile("" : "+v"(acc), "+v"(z));
  t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[11] = (t1 - t0);
  out[lane] = x + acc[0] + acc[1] + acc2[2] + s + v + z + zf;
}
int main() {
  double* out;
  long long* cyc;
  hipMalloc(&out, 64 * 8);
  hipMalloc(&cyc, 64 * 8);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, cyc, 1.25);
  hipDeviceSynchronize();
  long long h[12];
  hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
  const char* names[12] = {"v_rcp_f64 dependent", "v_fma_f64 dependent", "mfma_f64_16x16x4 accumulator chain",
                          "mfma <- VALU <- mfma result", "mfma <- readlane <- mfma result",
                          "pivot step (readlane, rcp + 2 Newton, mul, 2 mfma)", "mfma_f64_4x4x4 chain", "ds_bpermute dependent",
                          "1 mfma + 8 dependent v_fma_f64 (independent of it)", "1 mfma + 8 dependent v_fma_f32",
                          "2 mfma + 8 dependent v_fma_f64", "1 mfma + rcp, 4 fma, add (f64)"};
  for (int i = 0; i < 12; ++i) printf("%-55s %7.1f s_memtime ticks per iteration\n", names[i], h[i] / 256.0);
  int main2();
  return main2();
}

// ---- the whole 16 x 16 elimination (tiles.h: factor16_acc) on one wave, repeated: cycles per call --------------------
#include "../ls-spa_amd/csrc/tiles.h"
__global__ void probe_factor(double* out, long long* cyc, double seed) {
  using namespace lsspa;
  const int lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
  const int wave = threadIdx.x >> 6;
  if (wave != 0 && wave != 5) {     // the other waves of a 512-thread workgroup wait at the barrier, as in small_p
    __syncthreads();
    return;
  }
  d4 t0v, y0v;
  for (int r = 0; r < 4; ++r) {
    const int row = acc_row(l4, r);
    t0v[r] = (row == l15) ? 4.0 + 0.01 * row : 0.01 * seed / (1 + row + l15);
    y0v[r] = (row == l15) ? 1.0 : 0.0;
  }
  double sink = 0.0;
  int bad = 0;
  const long long c0 = __builtin_amdgcn_s_memtime();
  for (int rep = 0; rep < 64; ++rep) {
    lsspa::d4 t = t0v, y = y0v;
    t[0] += sink * 1e-30;
    factor16_acc<double>(t, y, 1e-12, lane, bad);
    sink += t[3] + y[1];
  }
  const long long c1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = c1 - c0;
  out[lane] = sink + bad;
  __syncthreads();
}
int main2() {
  double* out;
  long long* cyc;
  (void)hipMalloc(&out, 64 * 8);
  (void)hipMalloc(&cyc, 64 * 8);
  const int cfg[4][2] = {{1, 64}, {1, 512}, {256, 512}, {2048, 512}};
  for (int c = 0; c < 4; ++c) {
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe_factor, dim3(cfg[c][0]), dim3(cfg[c][1]), 0, 0, out, cyc, 1.25);
    (void)hipDeviceSynchronize();
    long long h[1];
    (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    printf("factor16_acc<double>, grid %4d x %3d threads: %.0f ticks per 16 x 16 block (%.1f per pivot)\n", cfg[c][0], cfg[c][1],
           h[0] / 64.0, h[0] / 64.0 / 16.0);
  }
  int main3();
  return main3();
}

// ---- does an fp64 matrix instruction of ONE wave hold up the vector instructions of ANOTHER wave of the same SIMD? -----
// 512 threads: waves w and w + 4 share a SIMD.  Wave `busy` runs mode_busy (0: nothing, 1: back-to-back fp64 MFMAs,
// 2: dependent v_fma_f64) for much longer than wave `timed` needs for 2048 dependent v_fma_f64 (or 256 MFMAs), which is timed.
__global__ void probe_pair(double* out, long long* cyc, int busy, int timed, int mode_busy, int mode_timed, double seed) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double x = seed + lane * 1e-3;
  d4 acc = {x, x, x, x};
  if (wave == busy) {
    if (mode_busy == 1) {
#pragma unroll 16
      for (int i = 0; i < 4096; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc, 0, 0, 0);   // register operands
    } else if (mode_busy == 2) {
#pragma unroll 16
      for (int i = 0; i < 40000; ++i) x = fma(x, 0.999999, 1e-9);
    }
  } else if (wave == timed) {
    // let the busy wave get going
    for (int i = 0; i < 64; ++i) x = fma(x, 0.999999, 1e-9);
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (mode_timed == 0) {
#pragma unroll 16
      for (int i = 0; i < 2048; ++i) x = fma(x, 0.999999, 1e-9);
    } else {
#pragma unroll 16
      for (int i = 0; i < 256; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc, 0, 0, 0);
    }
    asm volatile("" : "+v"(x), "+v"(acc));
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  }
  out[threadIdx.x] = x + acc[0];
}
int main3() {
  double* out;
  long long* cyc;
  (void)hipMalloc(&out, 512 * 8);
  (void)hipMalloc(&cyc, 64 * 8);
  struct { int busy, timed, mb, mt; const char* what; } cfg[] = {
      {0, 4, 0, 0, "2048 dependent v_fma_f64, the SIMD's other wave idle                    "},
      {0, 4, 1, 0, "2048 dependent v_fma_f64, the SIMD's other wave issuing fp64 MFMAs      "},
      {0, 4, 2, 0, "2048 dependent v_fma_f64, the SIMD's other wave issuing v_fma_f64       "},
      {0, 5, 1, 0, "2048 dependent v_fma_f64, a wave of ANOTHER SIMD issuing fp64 MFMAs     "},
      {0, 4, 0, 1, "256 fp64 MFMAs, the SIMD's other wave idle                              "},
      {0, 4, 1, 1, "256 fp64 MFMAs, the SIMD's other wave issuing fp64 MFMAs                "},
      {0, 4, 2, 1, "256 fp64 MFMAs, the SIMD's other wave issuing dependent v_fma_f64       "}};
  for (auto& c : cfg) {
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe_pair, dim3(1), dim3(512), 0, 0, out, cyc, c.busy, c.timed, c.mb, c.mt, 1.25);
    (void)hipDeviceSynchronize();
    long long h[1];
    (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    printf("%s %8lld cycles (%.1f per instruction)\n", c.what, h[0], h[0] / (c.mt ? 256.0 : 2048.0));
  }
  // the same pair on every CU of the chip (2048 workgroups): is the rate a property of the SIMD or of the chip?
  for (int mb = 0; mb < 3; ++mb) {
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe_pair, dim3(2048), dim3(512), 0, 0, out, cyc, 0, 4, mb, 1, 1.25);
    (void)hipDeviceSynchronize();
    long long h[1];
    (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    printf("2048 workgroups: 256 fp64 MFMAs of wave 4 while wave 0 of the same SIMD %s: %.1f cycles each\n",
           mb == 0 ? "is idle" : (mb == 1 ? "issues fp64 MFMAs" : "issues v_fma_f64"), h[0] / 256.0);
  }
  // which waves of a 512-thread workgroup share a SIMD?  Two MFMA streams on one SIMD must halve each other.
  for (int timed = 1; timed < 8; ++timed) {
    long long h[2][1];
    for (int mb = 0; mb < 2; ++mb) {
      for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe_pair, dim3(1), dim3(512), 0, 0, out, cyc, 0, timed, mb, 1, 1.25);
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(h[mb], cyc, sizeof h[mb], hipMemcpyDeviceToHost);
    }
    long long v[2][1];
    for (int mb = 0; mb < 2; ++mb) {
      for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe_pair, dim3(1), dim3(512), 0, 0, out, cyc, 0, timed, mb ? 2 : 0, 0, 1.25);
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(v[mb], cyc, sizeof v[mb], hipMemcpyDeviceToHost);
    }
    printf("wave %d timed, wave 0 busy: 256 MFMAs %.1f -> %.1f cycles each when wave 0 issues MFMAs too; 2048 v_fma_f64 %.1f -> %.1f each when wave 0 issues v_fma_f64\n",
           timed, h[0][0] / 256.0, h[1][0] / 256.0, v[0][0] / 2048.0, v[1][0] / 2048.0);
  }
  return 0;
}
