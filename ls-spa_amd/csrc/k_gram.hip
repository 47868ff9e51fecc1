// One-pass data reduction: the Gram contraction  C = [X | y]^T [X | y]  on the fp64
// matrix pipe.  Replaces the two Householder QRs of reduce_data
// (cvxgrp/ls-spa, ls_spa/ls_spa.py:309-317): R_tr^T R_tr = X^T X / N + reg I and
// R_tr^T y~ = X^T y / N are all the sampling loop ever needs of the N x p data.
//
// Decomposition: see gram_kernel below (128 x 128 tiles of the symmetric output x row slices; slabs of partial
// tiles summed in a fixed order by a second kernel: bitwise reproducible, no float atomics).
#include <algorithm>
#include <type_traits>

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

// A ragged tile (the last one: feature columns, then y at column p, then padding) in a chunk whose 16 rows all exist,
// WITHOUT a branch or a select between the loads and the products they run under (either puts a wait for the loads
// in front of the products; the element-wise guarded form below moves 8 bytes a lane and instruction through two to
// four times as many instructions -- on every unit that touches the last tile, and a launch waits for its slowest
// workgroups).  Every thread issues the same two loads per row: ONE 16-byte vector of X at a column that is always
// inside the row -- its own when its vector lies inside X, the last full vector of the row (columns p - VE .. p - 1)
// when its vector holds column p, column 0 when it lies beyond -- and y of that row (one address per wave
// instruction).  zfix_ragged() assembles what the thread's vector really is, after the products.  Needs p >= VE.
template <typename T>
struct RaggedRegs {
  T yv[ZRegs<T>::NP];
};

template <typename T>
__device__ __forceinline__ void zload_ragged(ZRegs<T>& r, RaggedRegs<T>& ry, const T* __restrict__ X,
                                             const T* __restrict__ y, int64_t ld, int p, int col_tile0, int64_t row0,
                                             int tid) {
  typedef ZRegs<T> R;
  const int vc = tid % R::VPR, k0 = tid / R::VPR;
  const int col0 = col_tile0 + R::VE * vc;
  const int colv = (col0 + R::VE <= p) ? col0 : (col0 <= p ? p - R::VE : 0);
#pragma unroll
  for (int q = 0; q < R::NP; ++q) {
    const int64_t row = row0 + k0 + R::RPP * q;
    r.v[q] = *reinterpret_cast<const typename R::gvec_t*>(X + row * ld + colv);
    ry.yv[q] = y[row];
  }
}

template <typename T>
__device__ __forceinline__ void zfix_ragged(ZRegs<T>& r, const RaggedRegs<T>& ry, int p, int col_tile0, int tid) {
  typedef ZRegs<T> R;
  const int vc = tid % R::VPR;
  const int col0 = col_tile0 + R::VE * vc;
  if (col0 + R::VE <= p) return;                 // inside X: the vector is what was loaded (per-thread, VALU only)
  const int m = p - col0;                        // feature columns in this vector: 0 .. VE - 1, or negative: beyond y
#pragma unroll
  for (int q = 0; q < R::NP; ++q) {
    typename R::vec_t v;
#pragma unroll
    for (int e = 0; e < R::VE; ++e) {
      // element e is column col0 + e: a feature (e < m: loaded as element e + VE - m of the row's last full vector),
      // y (e == m) or padding
      T x = (T)0;
#pragma unroll
      for (int f = 0; f < R::VE; ++f)
        if (f == e + R::VE - m) x = r.v[q][f];
      v[e] = (e < m) ? x : (e == m ? ry.yv[q] : (T)0);
    }
    r.v[q] = v;
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

// Work decomposition (round 3).  The symmetric output is cut into 128 x 128 tiles; the rows into slices.
// The UNITS are
//   * the off-diagonal tile pairs (ti > tj): a full 128 x 128 product, and
//   * "duos": TWO diagonal tiles (2 d, 2 d + 1) in one workgroup, each reduced to the 36 lower 16 x 16 blocks of its
//     8 x 8 block grid and dealt 18 / 18 over two waves -- 72 blocks for the workgroup against the 64 of a pair,
//     where a diagonal tile computed as a full pair would cost 64 for 36 useful (the four waves meet at a barrier
//     every chunk, so leaving out blocks pays only when every wave leaves out as many).
// Workgroup -> unit map: hardware deals consecutive workgroup ids round-robin over the 8 XCDs (each with its own L2),
// so unit u = (id % 8) * per_xcd + id / 8: every XCD works through a CONTIGUOUS range of units, i.e. the units of a
// row slice run side by side on one XCD and at the same pace, and a 128-column tile of rows that up to 8 units need
// is fetched into that L2 once.
// Inside a pair each wave owns all 8 row blocks x 2 column blocks of the tile (16 blocks, 8 + 2 operand fragments
// per k-step): the blocks of a ragged last tile that hold padding columns only are left out by EVERY wave alike.
// The partial tile goes to a slab; a second kernel sums the slabs in a fixed order, so the result is bitwise
// reproducible (no float atomics).
// lower blocks (bi >= bj) of a 4 x 4 block grid, row by row: t = 0..9
#define LSSPA_TRI_BI(t) ((t) < 1 ? 0 : (t) < 3 ? 1 : (t) < 6 ? 2 : 3)
#define LSSPA_TRI_BJ(t) ((t) - LSSPA_TRI_BI(t) * (LSSPA_TRI_BI(t) + 1) / 2)

// What a launch is cut into (host and device agree on it through this struct).  Units come in three classes -- A:
// off-diagonal pairs of full tiles (16 blocks a wave and k-step), B: pairs whose tile i is the ragged last tile
// (2 xlive blocks), C: duos (18) -- each with its own slice count (gram_plan: 16 : 18 : 20 from six tiles on).
struct GramPlan {
  int nt;                 // 128-column tiles of Z = [X | y]
  int xlive;              // live 16-column blocks of the last tile
  int cnt[3];             // units per slice, classes A, B, C
  int slices[3];          // row slices per class
  int rps[3];             // rows per slice (multiple of 16)
  int per_xcd;            // workgroups per XCD (grid = 8 x per_xcd)
  int natural;            // developer A/B: unit = workgroup id (units of a slice spread over the XCDs)
  __host__ __device__ int total() const { return cnt[0] * slices[0] + cnt[1] * slices[1] + cnt[2] * slices[2]; }
};

template <typename T>
__global__ __launch_bounds__(256, 2) void gram_kernel(const T* __restrict__ X, const T* __restrict__ y,
                                                      int64_t n, int64_t ld, int p, GramPlan plan,
                                                      double* __restrict__ slabs) {
  __shared__ __attribute__((aligned(16))) double s_i[2][16 * KC_LD];
  __shared__ __attribute__((aligned(16))) double s_j[2][16 * KC_LD];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // the wave index as the scalar it is
  const int l15 = lane & 15, l4 = lane >> 4;
  const int nt = plan.nt, n_pairs = nt * (nt + 1) / 2;
  int u = plan.natural ? (int)blockIdx.x : (int)(blockIdx.x & 7) * plan.per_xcd + (int)(blockIdx.x >> 3);
  if (u >= plan.total()) return;                           // padding of the grid to 8 x per_xcd (workgroup-uniform)
  int cls = 0;
  while (u >= plan.cnt[cls] * plan.slices[cls]) {
    u -= plan.cnt[cls] * plan.slices[cls];
    ++cls;
  }
  const int slice = u / plan.cnt[cls], unit = u - slice * plan.cnt[cls];
  const int rows_per_split = plan.rps[cls];
  // unit -> tiles.  Pair: tile i > tile j (class A: rows ti = 1 .. of the tile triangle in order, the ragged last row
  // excluded when class B holds it; class B: ti = nt - 1, tj = unit).  Duo: tile i = the LATER diagonal tile
  // (2 d + 1, or 2 d when it is the last of an odd count), tile j = 2 d (then "single": one tile only)
  const bool duo = (cls == 2);
  int ti, tj;
  if (cls == 0) {
    int q = unit;
    ti = 1;
    while (q >= ti) {
      q -= ti;
      ++ti;
    }
    tj = q;
  } else if (cls == 1) {
    ti = nt - 1;
    tj = unit;
  } else {
    tj = 2 * unit;
    ti = (tj + 1 < nt) ? tj + 1 : tj;
  }
  const bool single = (ti == tj);
  const int ci0 = ti * 128, cj0 = tj * 128;
  const int64_t r_lo = (int64_t)slice * rows_per_split;   // may lie beyond n for the last slices of a short matrix
  const int64_t r_hi = (r_lo + rows_per_split < n) ? r_lo + rows_per_split : n;
  const bool full_i = ci0 + 128 <= p;     // tile j <= tile i: full_i implies that tile j is full too
  // live 16-column blocks of tile i (columns 0 .. p, the last one being y); tile j < tile i is always all live
  const int xlive = (ti == nt - 1) ? plan.xlive : 8;

  const int64_t n_rows = r_hi > r_lo ? r_hi - r_lo : 0;
  const int n_chunks = (int)((n_rows + 15) / 16);
  // chunks whose 16 rows all exist (none, i.e. the row-guarded element loads throughout, for a p below one vector)
  const int n_full = (p >= ZRegs<T>::VE) ? (int)(n_rows / 16) : 0;
  ZRegs<T> ri, rj;
  RaggedRegs<T> ryi;
  // tile j < tile i is a full tile whenever it is staged at all; tile i is full or the ragged last one
  auto fetch_rows16 = [&](int c, auto full_i_tag) {   // a chunk whose 16 rows exist: no row guards
    const int64_t row0 = r_lo + (int64_t)c * 16;
    if constexpr (decltype(full_i_tag)::value) zload<T, true>(ri, X, y, ld, p, ci0, row0, r_hi, tid);
    else zload_ragged<T>(ri, ryi, X, y, ld, p, ci0, row0, tid);
    if (!single) zload<T, true>(rj, X, y, ld, p, cj0, row0, r_hi, tid);
  };
  auto fix_rows16 = [&](auto full_i_tag) {            // after the products: what a ragged tile's vectors really are
    if constexpr (!decltype(full_i_tag)::value) zfix_ragged<T>(ri, ryi, p, ci0, tid);
  };
  auto fetch_guarded = [&](int c) {   // raw values; mask_guarded(c) clears what must read as zero
    const int64_t row0 = r_lo + (int64_t)c * 16;
    zload<T, false>(ri, X, y, ld, p, ci0, row0, r_hi, tid);
    if (!single) zload<T, false>(rj, X, y, ld, p, cj0, row0, r_hi, tid);
  };
  auto mask_guarded = [&](int c) {
    const int64_t row0 = r_lo + (int64_t)c * 16;
    zmask<T>(ri, p, ci0, row0, r_hi, tid);
    if (!single) zmask<T>(rj, p, cj0, row0, r_hi, tid);
  };
  auto park = [&](int buf) {
    zstore<T>(ri, s_i[buf], tid);
    if (!single) zstore<T>(rj, s_j[buf], tid);
  };
  // The chunk pipeline, shared by both kinds of unit (products = the products of one staged chunk): double-buffered
  // LDS, one barrier per chunk, the global loads of chunk c + 1 issued before the products of chunk c and parked in
  // LDS after them.  Chunks [0, c_split) prefetch a chunk whose 16 rows exist, the last one possibly a chunk of the
  // row-guarded kind: separate loops, each with ONE kind of load, and the first of them in two copies (tile i full /
  // ragged).  (One loop choosing the kind per chunk made the compiler merge the loaded registers of the kinds after
  // the choice -- a full wait for the loads BEFORE the products they were meant to run under.)
  auto pipeline = [&](auto&& products) {
    const int c_split = (n_full > 1) ? n_full - 1 : 0;
    auto head_and_main = [&](auto full_i_tag) {
      if (n_chunks > 0) {
        if (n_full > 0) {
          fetch_rows16(0, full_i_tag);
          fix_rows16(full_i_tag);
        } else {
          fetch_guarded(0);
          mask_guarded(0);
        }
        park(0);
      }
      __syncthreads();
      for (int c = 0; c < c_split; ++c) {
        fetch_rows16(c + 1, full_i_tag);
        products(c & 1);
        fix_rows16(full_i_tag);
        park((c & 1) ^ 1);
        __syncthreads();
      }
    };
    if (full_i) head_and_main(std::true_type());
    else head_and_main(std::false_type());
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
  };
  double* const slab_base = slabs + (int64_t)slice * n_pairs * (128 * 128);

  if (!duo) {
    // ---- off-diagonal pair: wave w owns row blocks 0..7 of tile i x column blocks 2 w, 2 w + 1 of tile j
    d4 acc[8][2];
#pragma unroll
    for (int x = 0; x < 8; ++x)
#pragma unroll
      for (int yv = 0; yv < 2; ++yv) acc[x][yv] = d4_zero();
    auto products = [&](int cur, auto all_live_tag) {
      const double* si = s_i[cur];
      const double* sj = s_j[cur];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        double av[8], bv[2];
#pragma unroll
        for (int x = 0; x < 8; ++x) av[x] = si[(4 * kk + l4) * KC_LD + 16 * x + l15];
#pragma unroll
        for (int yv = 0; yv < 2; ++yv) bv[yv] = sj[(4 * kk + l4) * KC_LD + 32 * w + 16 * yv + l15];
#pragma unroll
        for (int x = 0; x < 8; ++x)
          if (decltype(all_live_tag)::value || x < xlive) {   // scalar condition: the same for every wave
#pragma unroll
            for (int yv = 0; yv < 2; ++yv) acc[x][yv] = mfma(av[x], bv[yv], acc[x][yv]);
          }
      }
    };
    // two copies of the loops: the common one without a condition in it
    if (xlive == 8) pipeline([&](int cur) { products(cur, std::true_type()); });
    else pipeline([&](int cur) { products(cur, std::false_type()); });
    double* slab = slab_base + (int64_t)(ti * (ti + 1) / 2 + tj) * (128 * 128);
#pragma unroll
    for (int x = 0; x < 8; ++x)
#pragma unroll
      for (int yv = 0; yv < 2; ++yv)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          slab[(16 * x + acc_row(l4, r)) * 128 + 32 * w + 16 * yv + l15] = acc[x][yv][r];
    return;
  }

  // ---- duo: waves 0, 1 take diagonal tile j (staged in s_j; in s_i when it is the only tile), waves 2, 3 tile i.
  // A diagonal block (bi, bj) is Z_bi^T Z_bj with both operands from the SAME staged tile.  The 36 lower blocks of
  // the 8 x 8 block grid are the lower triangles of two 4 x 4 grids (rows 0..3; rows 4..7 x columns 4..7: 10 blocks
  // each) and the 4 x 4 square between them (rows 4..7 x columns 0..3); the wave with half = 0 takes the first
  // triangle and the square's rows 4, 5, the other the second triangle and rows 6, 7: 18 blocks each, and the SAME
  // instruction stream for both -- only the LDS offsets of the operand fragments depend on `half`.
  const int half = w & 1;
  const bool mine_is_i = single ? (w < 2) : (w >= 2);      // which staged tile this wave works on
  const bool idle = single && w >= 2;
  d4 tri[10], sq[2][4];
#pragma unroll
  for (int t = 0; t < 10; ++t) tri[t] = d4_zero();
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) sq[r][c] = d4_zero();
  const int o_tri = 64 * half + l15, o_row = 64 + 32 * half + l15, o_col = l15;
  pipeline([&](int cur) {
    if (idle) return;
    const double* sz = mine_is_i ? s_i[cur] : s_j[cur];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const double* row = sz + (4 * kk + l4) * KC_LD;
      double zt[4], zr[2], zc[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) zt[b] = row[o_tri + 16 * b];
#pragma unroll
      for (int b = 0; b < 2; ++b) zr[b] = row[o_row + 16 * b];
#pragma unroll
      for (int b = 0; b < 4; ++b) zc[b] = row[o_col + 16 * b];
#pragma unroll
      for (int t = 0; t < 10; ++t) tri[t] = mfma(zt[LSSPA_TRI_BI(t)], zt[LSSPA_TRI_BJ(t)], tri[t]);
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) sq[r][c] = mfma(zr[r], zc[c], sq[r][c]);
    }
  });
  if (idle) return;
  const int td = mine_is_i ? ti : tj;
  double* slab = slab_base + (int64_t)(td * (td + 1) / 2 + td) * (128 * 128);
#pragma unroll
  for (int t = 0; t < 10; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      slab[(64 * half + 16 * LSSPA_TRI_BI(t) + acc_row(l4, r)) * 128 + 64 * half + 16 * LSSPA_TRI_BJ(t) + l15] = tri[t][r];
#pragma unroll
  for (int rr = 0; rr < 2; ++rr)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        slab[(64 + 32 * half + 16 * rr + acc_row(l4, r)) * 128 + 16 * c + l15] = sq[rr][c][r];
}

// C[i][j] = C[j][i] = sum over slices of the pair's slab, fixed order.  One workgroup per 32 x 32 sub-tile of a pair
// (16 per pair: enough workgroups to cover the load latency of the n_split slabs), the mirror image written from an
// LDS copy so that both stores are coalesced (written straight, the mirror was a 64-way scattered store per wave and
// the whole reduction took a tenth of the Gram time for a fiftieth of its bytes).  Diagonal tiles: the gram kernel
// writes their lower 16 x 16 blocks only; the blocks above the diagonal are stored as zeros (nobody reads them:
// gram_finalize symmetrises from the lower part) and never loaded.
__global__ __launch_bounds__(256) void gram_reduce_kernel(const double* __restrict__ slabs, GramPlan plan,
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
  const bool diag = (ti == tj);
  if (diag && sy < sx) return;     // wholly above the diagonal of a diagonal tile: never written, never read
  // slices of this pair's class (the gram kernel's own classification)
  const int n_split = diag ? plan.slices[2] : (plan.cnt[1] > 0 && ti == plan.nt - 1) ? plan.slices[1] : plan.slices[0];
  double v[4] = {0.0, 0.0, 0.0, 0.0};
  bool live[4];
  const double* src[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = sy * 32 + r0 + 8 * q, col = sx * 32 + c;
    live[q] = !(diag && (row >> 4) < (col >> 4));
    src[q] = slabs + (int64_t)blockIdx.x * (128 * 128) + row * 128 + col;
  }
  // slices outermost: the four loads of a slice are independent and in flight together; each element's sum still
  // runs over the slices in their fixed order
  const int64_t slice_stride = (int64_t)n_pairs * (128 * 128);
  for (int k = 0; k < n_split; ++k) {
    double t[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) t[q] = live[q] ? src[q][(int64_t)k * slice_stride] : 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] += t[q];
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t o = (int64_t)(i0 + r0 + 8 * q) * P1pad + j0 + c;
    if (accumulate && live[q]) v[q] += C[o];   // fixed chunk order: still reproducible
    C[o] = v[q];
  }
  if (diag) return;                // workgroup-uniform
#pragma unroll
  for (int q = 0; q < 4; ++q) s_t[r0 + 8 * q][c] = v[q];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q)      // row j0 + (r0 + 8 q) of the mirror image, columns i0 + c
    C[(int64_t)(j0 + r0 + 8 * q) * P1pad + i0 + c] = s_t[c][r0 + 8 * q];
}

// The same sum for problems with only a few tile pairs (p <= ~250), where 16 workgroups a pair leave the chip empty
// and the slices are many and short: one workgroup per 16 x 16 block of the pair, one element per thread, the slices
// fetched eight at a time (independent loads) and added in their fixed order.
__global__ __launch_bounds__(256) void gram_reduce_small_kernel(const double* __restrict__ slabs, GramPlan plan,
                                                                int n_pairs, int P1pad, double* __restrict__ C,
                                                                int accumulate) {
  int pair = blockIdx.x, ti = 0;
  while (pair >= ti + 1) {
    pair -= ti + 1;
    ++ti;
  }
  const int tj = pair;
  const int by = blockIdx.y >> 3, bx = blockIdx.y & 7;          // 16 x 16 block (rows, columns) of the 128 x 128 tile
  const bool diag = (ti == tj);
  if (diag && by < bx) return;     // above the diagonal of a diagonal tile: never written, never read
  const int n_split = diag ? plan.slices[2] : (plan.cnt[1] > 0 && ti == plan.nt - 1) ? plan.slices[1] : plan.slices[0];
  const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
  const int row = 16 * by + r, col = 16 * bx + c;
  const double* src = slabs + (int64_t)blockIdx.x * (128 * 128) + row * 128 + col;
  const int64_t slice_stride = (int64_t)n_pairs * (128 * 128);
  double v = 0.0;
  int k = 0;
  for (; k + 8 <= n_split; k += 8) {
    double t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = src[(int64_t)(k + u) * slice_stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) v += t[u];
  }
  for (; k < n_split; ++k) v += src[(int64_t)k * slice_stride];
  const int64_t o = (int64_t)(ti * 128 + row) * P1pad + tj * 128 + col;
  if (accumulate) v += C[o];       // fixed chunk order: still reproducible
  C[o] = v;
  if (!diag) C[(int64_t)(tj * 128 + col) * P1pad + ti * 128 + row] = v;
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

static inline int n_tiles_of(int p) { return (p + 1 + 127) / 128; }

static inline int n_pairs_of(int p) {
  const int nt = n_tiles_of(p);
  return nt * (nt + 1) / 2;
}

// n_split = row slices of class A (the caller's knob, gram_default_split); the other classes in proportion to their cost
static GramPlan gram_plan(int64_t n, int p, int n_split, int variant = 0) {
  GramPlan g;
  g.nt = n_tiles_of(p);
  g.xlive = std::min(8, (p + 1 - (g.nt - 1) * 128 + 15) / 16);
  const bool ragged = g.xlive < 8 && g.nt > 1;
  g.cnt[1] = ragged ? g.nt - 1 : 0;
  g.cnt[0] = g.nt * (g.nt - 1) / 2 - g.cnt[1];
  g.cnt[2] = (g.nt + 1) / 2;
  // Relative time per row of the classes.  Round 3 measured "equal" and kept 16 : 16 : 16; round 5 measured again, after
  // the ragged tile's loads had been fixed: a launch waits for its slowest workgroups, and from six tiles on those are
  // the duos (18 matrix instructions a wave and k-step against a pair's 16, two of them on a CU when the class sits
  // together) and, less so, the ragged pairs.  With 16 : 18 : 20 -- the slower classes cut into proportionally more,
  // shorter slices -- p = 1000: 1.91-1.97 -> 1.70-1.80 ms a side, p = 1500 (60 k rows): 2.52-2.58 -> 2.27-2.42,
  // p = 2000: 6.5-7.6 -> 6.3-6.6, p = 3000 / 5000: unchanged; below six tiles (p = 500: 0.54 -> 0.67 ms) the few units
  // fill the chip's slots better undivided.  (Handing the work items out by first row, or the slow classes spread
  // among the fast ones, instead of class by class: 1.80 / 1.86 against 1.72-1.75 -- measured, not kept.)
  const bool split_slow = g.nt >= 6;
  const int cost[3] = {16, split_slow ? 18 : 16, split_slow ? 20 : 16};
  (void)variant;
  // with no class-A unit at all (one or two tiles) the knob applies to the classes that exist, undivided
  const int64_t max_s = std::max<int64_t>(1, (n + 15) / 16);
  for (int c = 0; c < 3; ++c) {
    int64_t sl = ((int64_t)n_split * cost[c] + 8) / 16;
    sl = std::max<int64_t>(1, std::min<int64_t>(sl, max_s));
    g.slices[c] = (int)sl;
    int64_t rps = (n + sl - 1) / sl;
    g.rps[c] = (int)(((rps + 15) / 16) * 16);
  }
  g.per_xcd = (g.total() + 7) / 8;
  g.natural = 0;
  return g;
}

size_t gram_workspace_bytes(int p, int n_split) {
  // the largest per-class slice count is that of the duos: round(20 / 16 n_split)
  const size_t s_max = (size_t)(((int64_t)n_split * 20 + 8) / 16) + 1;
  return (size_t)n_pairs_of(p) * std::max<size_t>(s_max, (size_t)n_split) * 128 * 128 * sizeof(double);
}

int gram_default_split(int64_t n, int p) {
  // Workgroups of equal length, two resident per CU (512 at a time): pick the class-A slice count whose total fills
  // whole rounds of 512 best (p = 1000: 16 -> 21 x 16 + 7 x 14 + 4 x 18 = 506), with slices of at least 256 rows
  // and at most ~8 rounds
  const int nt = n_tiles_of(p);
  // slices of at least 256 rows -- 128 where the units are so few (two tiles) that the launch is a chain of load round
  // trips, one per 16-row chunk, whatever the slice count, and 48 for a single tile (C2, p = 100, 10^4 rows: 34.2 us a
  // side at 128 rows, 29.6 at 96, 28.9 at 64, 28.0 at 48, 35.5 at 32 -- below that the slabs' traffic takes over)
  const int64_t min_rows = (nt == 1) ? 48 : ((nt == 2) ? 128 : 256);
  const int64_t cap = std::max<int64_t>(1, (n + min_rows - 1) / min_rows);
  int best = 1;
  double best_eff = 0.0;
  for (int64_t s = 1; s <= cap; ++s) {
    const GramPlan g = gram_plan(n, p, (int)s);
    const int64_t total = g.total(), rounds = (total + 511) / 512;
    if (rounds > 8 && s > 1) break;
    const double eff = (double)total / (double)(rounds * 512);
    if (eff > best_eff + 1e-9) {
      best_eff = eff;
      best = (int)s;
    }
  }
  (void)nt;
  return best;
}

hipError_t launch_gram(const GramArgs& a, hipStream_t st) {
  if (a.n < 1 || a.p < 1 || a.ld < a.p || a.n_split < 1) return hipErrorInvalidValue;
  const int nt = n_tiles_of(a.p), np = n_pairs_of(a.p);
  const int P1pad = nt * 128;
  GramPlan g = gram_plan(a.n, a.p, a.n_split, a.variant);
  g.natural = a.variant & 1;
  for (int c = 0; c < 3; ++c)
    if ((int64_t)g.rps[c] * g.slices[c] < a.n || (size_t)g.slices[c] * np * 128 * 128 * 8 > gram_workspace_bytes(a.p, a.n_split))
      return hipErrorInvalidValue;
  dim3 grid((unsigned)(g.per_xcd * 8));
  if (a.is_f32)
    hipLaunchKernelGGL(gram_kernel<float>, grid, dim3(256), 0, st, (const float*)a.X, (const float*)a.y,
                       a.n, a.ld, a.p, g, a.slabs);
  else
    hipLaunchKernelGGL(gram_kernel<double>, grid, dim3(256), 0, st, (const double*)a.X,
                       (const double*)a.y, a.n, a.ld, a.p, g, a.slabs);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (np <= 3)
    hipLaunchKernelGGL(gram_reduce_small_kernel, dim3(np, 64), dim3(256), 0, st, a.slabs, g, np, P1pad, a.C,
                       a.accumulate);
  else
    hipLaunchKernelGGL(gram_reduce_kernel, dim3(np, 16), dim3(256), 0, st, a.slabs, g, np, P1pad, a.C, a.accumulate);
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
