// One-pass data reduction: the Gram contraction  C = [X | y]^T [X | y]  on the fp64
// matrix pipe.  Replaces the two Householder QRs of reduce_data
// (cvxgrp/ls-spa, ls_spa/ls_spa.py:309-317): R_tr^T R_tr = X^T X / N + reg I and
// R_tr^T y~ = X^T y / N are all the sampling loop ever needs of the N x p data.
//
// Decomposition: the symmetric output is cut into 128 x 128 tiles (lower tile pairs only);
// the N rows are cut into n_split slices; one workgroup owns (tile pair, slice) and writes
// its partial tile to a slab; a second kernel sums the slabs in a fixed order, so the result
// is bitwise reproducible (no float atomics).
#include <algorithm>

#include "kernels.h"
#include "tiles.h"

namespace lsspa {

// One 16-row chunk of a 128-column tile of Z = [X | y], held in registers between its global load and its LDS
// store: 16-byte vectors, one wave instruction = one contiguous row of the tile (1 KB fp64, 512 B fp32).
template <typename T>
struct ZRegs {
  static constexpr int VE = 16 / sizeof(T);     // elements per vector: 2 / 4
  static constexpr int VPR = 128 / VE;          // vectors per tile row: 64 / 32
  static constexpr int RPP = 256 / VPR;         // rows per pass: 4 / 8
  static constexpr int NP = 16 / RPP;           // passes: 4 / 2
  typedef T vec_t __attribute__((ext_vector_type(16 / sizeof(T))));
  // the same vector as it lies in the caller's matrix: rows start at any multiple of sizeof(T) (ld need not be even)
  typedef T gvec_t __attribute__((ext_vector_type(16 / sizeof(T)), aligned(sizeof(T))));
  vec_t v[NP];
};

// FULL: the tile lies inside the features and the 16 rows exist -- no guards at all, 16-byte loads.
// Otherwise (ragged last tile, y column, last rows of a slice): element loads, every one of them unconditional --
// the row is clamped into the slice, the column's source is chosen by an address select (a feature column of X, or
// y for column p and, harmlessly, beyond) -- and the values that must read as zero are cleared by zmask() AFTER the
// products the loads run under.  A guarded load or a select right behind the load would put a wait for the loads
// in front of those products.
template <typename T, bool FULL>
__device__ __forceinline__ void zload(ZRegs<T>& r, const T* __restrict__ X, const T* __restrict__ y, int64_t ld,
                                      int p, int col_tile0, int64_t row0, int64_t r_hi, int tid) {
  typedef ZRegs<T> R;
  const int vc = tid % R::VPR, k0 = tid / R::VPR;
  const int col0 = col_tile0 + R::VE * vc;
#pragma unroll
  for (int q = 0; q < R::NP; ++q) {
    const int64_t row = row0 + k0 + R::RPP * q;
    if constexpr (FULL) {
      r.v[q] = *reinterpret_cast<const typename R::gvec_t*>(X + row * ld + col0);
    } else {
      const int64_t rc = row < r_hi ? row : r_hi - 1;
      typename R::vec_t v;
#pragma unroll
      for (int e = 0; e < R::VE; ++e) {
        const int c = col0 + e;
        const T* src = (c < p) ? X + rc * ld + c : y + rc;
        v[e] = *src;
      }
      r.v[q] = v;
    }
  }
}

// the zeros of a guarded chunk: columns beyond y, rows at or beyond r_hi
template <typename T>
__device__ __forceinline__ void zmask(ZRegs<T>& r, int p, int col_tile0, int64_t row0, int64_t r_hi, int tid) {
  typedef ZRegs<T> R;
  const int vc = tid % R::VPR, k0 = tid / R::VPR;
  const int col0 = col_tile0 + R::VE * vc;
#pragma unroll
  for (int q = 0; q < R::NP; ++q) {
    const bool row_ok = row0 + k0 + R::RPP * q < r_hi;
#pragma unroll
    for (int e = 0; e < R::VE; ++e)
      if (!(row_ok && col0 + e <= p)) r.v[q][e] = (T)0;
  }
}

// registers -> LDS tile [16 k][KC_LD] of doubles (fp32 data is widened here: the contraction runs in fp64)
template <typename T>
__device__ __forceinline__ void zstore(const ZRegs<T>& r, double* lds, int tid) {
  typedef ZRegs<T> R;
  const int vc = tid % R::VPR, k0 = tid / R::VPR;
#pragma unroll
  for (int q = 0; q < R::NP; ++q) {
    double* dst = lds + (k0 + R::RPP * q) * KC_LD + R::VE * vc;
#pragma unroll
    for (int e = 0; e < R::VE; e += 2) {
      v2d w = {(double)r.v[q][e], (double)r.v[q][e + 1]};
      *reinterpret_cast<v2d*>(dst + e) = w;
    }
  }
}

// Workgroup = (tile pair, row slice): C_slab[128][128] = Z[rows, tile i]^T Z[rows, tile j] over the slice's rows,
// 16 rows per k-chunk.  Double-buffered LDS (one barrier per chunk): the global loads of chunk c + 1 are issued
// before the 64 MFMAs per wave of chunk c and parked in LDS after them.  Two workgroups per CU.
template <typename T>
__global__ __launch_bounds__(256, 2) void gram_kernel(const T* __restrict__ X, const T* __restrict__ y,
                                                      int64_t n, int64_t ld, int p, int rows_per_split,
                                                      int n_pairs, double* __restrict__ slabs) {
  __shared__ __attribute__((aligned(16))) double s_i[2][16 * KC_LD];
  __shared__ __attribute__((aligned(16))) double s_j[2][16 * KC_LD];
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
  const bool full_i = ci0 + 128 <= p;     // tile j <= tile i: full_i implies that tile j is full too

  d4 acc[4][4];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int yv = 0; yv < 4; ++yv) acc[x][yv] = d4_zero();

  const int64_t n_rows = r_hi > r_lo ? r_hi - r_lo : 0;
  const int n_chunks = (int)((n_rows + 15) / 16);
  const int n_full = (int)(n_rows / 16);                           // chunks whose 16 rows all exist
  ZRegs<T> ri, rj;
  auto fetch_full = [&](int c) {      // a chunk whose 16 rows exist, of tiles inside the features: no guards
    const int64_t row0 = r_lo + (int64_t)c * 16;
    zload<T, true>(ri, X, y, ld, p, ci0, row0, r_hi, tid);
    if (!diag) zload<T, true>(rj, X, y, ld, p, cj0, row0, r_hi, tid);
  };
  auto fetch_guarded = [&](int c) {   // raw values; mask_guarded(c) clears what must read as zero
    const int64_t row0 = r_lo + (int64_t)c * 16;
    zload<T, false>(ri, X, y, ld, p, ci0, row0, r_hi, tid);
    if (!diag) zload<T, false>(rj, X, y, ld, p, cj0, row0, r_hi, tid);
  };
  auto mask_guarded = [&](int c) {
    const int64_t row0 = r_lo + (int64_t)c * 16;
    zmask<T>(ri, p, ci0, row0, r_hi, tid);
    if (!diag) zmask<T>(rj, p, cj0, row0, r_hi, tid);
  };
  auto products = [&](int cur) {
    const double* si = s_i[cur];
    const double* sj = diag ? s_i[cur] : s_j[cur];
    // diagonal pair: only the lower triangle of the tile is ever read (gram_finalize symmetrises from it), so the
    // wave that owns the upper 64 x 64 quadrant sits the products out -- its SIMD's matrix pipe goes to the
    // co-resident workgroup
    if (!(diag && wi < wj)) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        double av[4], bv[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) av[x] = si[(4 * kk + l4) * KC_LD + 64 * wi + 16 * x + l15];
#pragma unroll
        for (int yv = 0; yv < 4; ++yv) bv[yv] = sj[(4 * kk + l4) * KC_LD + 64 * wj + 16 * yv + l15];
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
          for (int yv = 0; yv < 4; ++yv) acc[x][yv] = mfma(av[x], bv[yv], acc[x][yv]);
      }
    }
  };
  auto park = [&](int buf) {
    zstore<T>(ri, s_i[buf], tid);
    if (!diag) zstore<T>(rj, s_j[buf], tid);
  };
  // Chunks [0, c_split) prefetch a chunk of the guard-free kind, the rest one of the guarded kind: TWO loops, each
  // with one kind of load.  (One loop choosing the kind per chunk made the compiler merge the loaded registers of
  // the two kinds after the choice -- a full wait for the loads BEFORE the products they were meant to run under.)
  const int c_split = (full_i && n_full > 1) ? n_full - 1 : 0;
  if (n_chunks > 0) {
    if (full_i && n_full > 0) {
      fetch_full(0);
    } else {
      fetch_guarded(0);
      mask_guarded(0);
    }
    park(0);
  }
  __syncthreads();
  for (int c = 0; c < c_split; ++c) {
    fetch_full(c + 1);
    products(c & 1);
    park((c & 1) ^ 1);
    __syncthreads();
  }
  for (int c = c_split; c < n_chunks; ++c) {
    const bool more = c + 1 < n_chunks;
    if (more) fetch_guarded(c + 1);
    products(c & 1);
    if (more) {
      mask_guarded(c + 1);
      park((c & 1) ^ 1);
    }
    __syncthreads();
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

// C[i][j] = C[j][i] = sum over slices of the pair's slab, fixed order.  One workgroup per 32 x 32 sub-tile of a pair
// (16 per pair: enough workgroups to cover the load latency of the n_split slabs), the mirror image written from an
// LDS copy so that both stores are coalesced (written straight, the mirror was a 64-way scattered store per wave and
// the whole reduction took a tenth of the Gram time for a fiftieth of its bytes).
__global__ __launch_bounds__(256) void gram_reduce_kernel(const double* __restrict__ slabs, int n_split,
                                                          int n_pairs, int P1pad, double* __restrict__ C,
                                                          int accumulate) {
  __shared__ double s_t[32][33];
  int pair = blockIdx.x, ti = 0;
  while (pair >= ti + 1) {
    pair -= ti + 1;
    ++ti;
  }
  const int tj = pair;
  const int sy = blockIdx.y >> 2, sx = blockIdx.y & 3;          // sub-tile (rows, columns) of the 128 x 128 tile
  const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;        // column, first row; rows r0 + 8 q
  const int i0 = ti * 128 + sy * 32, j0 = tj * 128 + sx * 32;
  double v[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = (sy * 32 + r0 + 8 * q) * 128 + sx * 32 + c;
    double s = 0.0;
    for (int k = 0; k < n_split; ++k) s += slabs[((int64_t)k * n_pairs + blockIdx.x) * (128 * 128) + e];
    const int64_t o = (int64_t)(i0 + r0 + 8 * q) * P1pad + j0 + c;
    if (accumulate) s += C[o];   // fixed chunk order: still reproducible
    C[o] = s;
    v[q] = s;
  }
  if (ti == tj) return;            // workgroup-uniform
#pragma unroll
  for (int q = 0; q < 4; ++q) s_t[r0 + 8 * q][c] = v[q];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q)      // row j0 + (r0 + 8 q) of the mirror image, columns i0 + c
    C[(int64_t)(j0 + r0 + 8 * q) * P1pad + i0 + c] = s_t[c][r0 + 8 * q];
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
  // Workgroups = tile pairs x row slices, two resident per CU (512 at a time) and all about equally long: pick
  // the slice count whose total fills whole rounds of 512 best (36 pairs x 28 slices = 1008 at p = 1000), with
  // slices of at least 256 rows and at most ~8 rounds
  const int np = n_pairs_of(p);
  const int64_t max_s = std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, (8 * 512 + np - 1) / np));
  int best = 1;
  double best_eff = 0.0;
  for (int64_t s = 1; s <= max_s; ++s) {
    const int64_t total = (int64_t)np * s, rounds = (total + 511) / 512;
    const double eff = (double)total / (double)(rounds * 512);
    if (eff > best_eff + 1e-9) {
      best_eff = eff;
      best = (int)s;
    }
  }
  return best;
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
  hipLaunchKernelGGL(gram_reduce_kernel, dim3(np, 16), dim3(256), 0, st, a.slabs, a.n_split, np, P1pad, a.C,
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
