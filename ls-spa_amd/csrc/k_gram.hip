// One-pass data reduction: the Gram contraction  C = [X | y]^T [X | y]  on the fp64
// matrix pipe.  Replaces the two Householder QRs of reduce_data
// (cvxgrp/ls-spa, ls_spa/ls_spa.py:309-317): R_tr^T R_tr = X^T X / N + reg I and
// R_tr^T y~ = X^T y / N are all the sampling loop ever needs of the N x p data.
//
// Decomposition: the symmetric output is cut into 128 x 128 tiles (lower tile pairs only);
// the N rows are cut into n_split slices; one workgroup owns (tile pair, slice) and writes
// its partial tile to a slab; a second kernel sums the slabs in a fixed order, so the result
// is bitwise reproducible (no float atomics).
#include "kernels.h"
#include "tiles.h"

namespace lsspa {

template <typename T>
__device__ __forceinline__ double zload(const T* __restrict__ X, const T* __restrict__ y, int64_t row,
                                        int col, int64_t ld, int p, int64_t n) {
  if (row >= n) return 0.0;
  if (col < p) return (double)X[row * ld + col];
  if (col == p) return (double)y[row];
  return 0.0;
}

template <typename T>
__global__ __launch_bounds__(256, 1) void gram_kernel(const T* __restrict__ X, const T* __restrict__ y,
                                                      int64_t n, int64_t ld, int p, int rows_per_split,
                                                      int n_pairs, double* __restrict__ slabs) {
  __shared__ __attribute__((aligned(16))) double s_i[16 * KC_LD];
  __shared__ __attribute__((aligned(16))) double s_j[16 * KC_LD];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  // tile pair index -> (ti >= tj)
  int pair = blockIdx.x, ti = 0;
  while (pair >= ti + 1) {
    pair -= ti + 1;
    ++ti;
  }
  const int tj = pair;
  const bool diag = (ti == tj);
  const int ci0 = ti * 128, cj0 = tj * 128;
  const int64_t r_lo = (int64_t)blockIdx.y * rows_per_split;
  const int64_t r_hi = (r_lo + rows_per_split < n) ? r_lo + rows_per_split : n;
  const int wi = w >> 1, wj = w & 1;  // wave quadrant: rows 64 wi .., cols 64 wj ..

  d4 acc[4][4];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int yv = 0; yv < 4; ++yv) acc[x][yv] = d4_zero();

  const int col = tid & 127, kr = tid >> 7;  // 2 k-rows per pass, 8 passes
  double ri[8], rj[8];
  auto fetch = [&](int64_t r0) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int64_t row = r0 + kr + 2 * q;
      ri[q] = (row < r_hi) ? zload<T>(X, y, row, ci0 + col, ld, p, n) : 0.0;
      if (!diag) rj[q] = (row < r_hi) ? zload<T>(X, y, row, cj0 + col, ld, p, n) : 0.0;
    }
  };
  if (r_lo < r_hi) fetch(r_lo);
  for (int64_t r0 = r_lo; r0 < r_hi; r0 += 16) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      s_i[(kr + 2 * q) * KC_LD + col] = ri[q];
      if (!diag) s_j[(kr + 2 * q) * KC_LD + col] = rj[q];
    }
    __syncthreads();
    if (r0 + 16 < r_hi) fetch(r0 + 16);
    const double* sj = diag ? s_i : s_j;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      double av[4], bv[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) av[x] = s_i[(4 * kk + l4) * KC_LD + 64 * wi + 16 * x + l15];
#pragma unroll
      for (int yv = 0; yv < 4; ++yv) bv[yv] = sj[(4 * kk + l4) * KC_LD + 64 * wj + 16 * yv + l15];
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int yv = 0; yv < 4; ++yv) acc[x][yv] = mfma(av[x], bv[yv], acc[x][yv]);
    }
  }
  double* slab = slabs + ((int64_t)blockIdx.y * n_pairs + blockIdx.x) * (128 * 128);
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int yv = 0; yv < 4; ++yv)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        slab[(64 * wi + 16 * x + acc_row(l4, r)) * 128 + 64 * wj + 16 * yv + l15] = acc[x][yv][r];
}

// C[i][j] = C[j][i] = sum over slices of the pair's slab, fixed order
__global__ __launch_bounds__(256) void gram_reduce_kernel(const double* __restrict__ slabs, int n_split,
                                                          int n_pairs, int P1pad, double* __restrict__ C,
                                                          int accumulate) {
  int pair = blockIdx.x, ti = 0;
  while (pair >= ti + 1) {
    pair -= ti + 1;
    ++ti;
  }
  const int tj = pair;
  for (int e = threadIdx.x + 256 * blockIdx.y; e < 128 * 128; e += 256 * gridDim.y) {
    double s = 0.0;
    for (int k = 0; k < n_split; ++k) s += slabs[((int64_t)k * n_pairs + blockIdx.x) * (128 * 128) + e];
    const int i = ti * 128 + (e >> 7), j = tj * 128 + (e & 127);
    if (accumulate) s += C[(int64_t)i * P1pad + j];   // fixed chunk order: still reproducible
    C[(int64_t)i * P1pad + j] = s;
    if (ti != tj) C[(int64_t)j * P1pad + i] = s;
  }
}

__global__ __launch_bounds__(256) void gram_finalize_kernel(const double* __restrict__ C, int P1pad, int p,
                                                            double scale, double reg, double* __restrict__ G,
                                                            int64_t ldg, double* __restrict__ g,
                                                            double* __restrict__ scalar_out) {
  const int64_t total = (int64_t)p * p;
  for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
    const int a = (int)(o / p), b = (int)(o - (int64_t)a * p);
    // the diagonal tiles were accumulated as full tiles: symmetrise from the lower part
    const double v = (a >= b) ? C[(int64_t)a * P1pad + b] : C[(int64_t)b * P1pad + a];
    G[(int64_t)a * ldg + b] = v * scale + ((a == b) ? reg : 0.0);
  }
  if (blockIdx.x == 0) {
    for (int a = threadIdx.x; a < p; a += 256) g[a] = C[(int64_t)p * P1pad + a] * scale;
    if (threadIdx.x == 0) scalar_out[0] = C[(int64_t)p * P1pad + p] * scale;
  }
}

static inline int n_pairs_of(int p) {
  const int nt = (p + 1 + 127) / 128;
  return nt * (nt + 1) / 2;
}

size_t gram_workspace_bytes(int p, int n_split) {
  return (size_t)n_pairs_of(p) * n_split * 128 * 128 * sizeof(double);
}

int gram_default_split(int64_t n, int p) {
  // aim at ~4 workgroups per CU, slices of at least 64 rows, multiples of 16 rows
  const int np = n_pairs_of(p);
  int64_t s = (1024 + np - 1) / np;
  const int64_t max_s = (n + 63) / 64;
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  return (int)s;
}

hipError_t launch_gram(const GramArgs& a, hipStream_t st) {
  if (a.n < 1 || a.p < 1 || a.ld < a.p || a.n_split < 1) return hipErrorInvalidValue;
  const int np = n_pairs_of(a.p);
  const int P1pad = ((a.p + 1 + 127) / 128) * 128;
  int64_t rps = (a.n + a.n_split - 1) / a.n_split;
  rps = ((rps + 15) / 16) * 16;
  if (rps * a.n_split < a.n || rps > 0x7fffffff) return hipErrorInvalidValue;
  dim3 grid(np, a.n_split);
  if (a.is_f32)
    hipLaunchKernelGGL(gram_kernel<float>, grid, dim3(256), 0, st, (const float*)a.X, (const float*)a.y,
                       a.n, a.ld, a.p, (int)rps, np, a.slabs);
  else
    hipLaunchKernelGGL(gram_kernel<double>, grid, dim3(256), 0, st, (const double*)a.X,
                       (const double*)a.y, a.n, a.ld, a.p, (int)rps, np, a.slabs);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(gram_reduce_kernel, dim3(np, 8), dim3(256), 0, st, a.slabs, a.n_split, np, P1pad, a.C,
                     a.accumulate);
  return hipGetLastError();
}

hipError_t launch_gram_finalize(const double* C, int p, double scale, double reg, double* G, int64_t ldg,
                                double* g, double* scalar_out, hipStream_t st) {
  if (p < 1 || ldg < p) return hipErrorInvalidValue;
  const int P1pad = ((p + 1 + 127) / 128) * 128;
  const int64_t total = (int64_t)p * p;
  const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(gram_finalize_kernel, dim3(grid), dim3(256), 0, st, C, P1pad, p, scale, reg, G, ldg, g,
                     scalar_out);
  return hipGetLastError();
}

}  // namespace lsspa
