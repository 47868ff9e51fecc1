// Fused per-ordering kernel for small problems (p + 1 <= 128, fp64, test Gram available): ONE workgroup takes one
// ordering from the permuted gather to its lift vector with both work matrices resident in LDS -- nothing but the
// source Gram matrices is read and nothing but the p lifts is written.
//
// What it replaces in the reference (cvxgrp/ls-spa, ls_spa/ls_spa.py:256-287, square_shapley), in the Gram form of
// DESIGN.md section 3:   G_pi = L L^T, H_pi = L_t L_t^T (augmented rows carry z = L^-1 g_pi, y~ = L_t^-1 h_pi),
// V = L^-1 L_t,  lift_j = z_j V[j,:] . (2 y~ - N_j - N_{j-1}) / ||y_test||^2  with  N_j = sum_{k<=j} z_k V[k,:].
// The large-p path runs the same steps as five kernel classes with HBM round trips in between; at p = 100 those are
// latency-bound launches of a few dozen microseconds each.
//
// Layout in LDS: each matrix as its lower 16 x 16 blocks only (block (i, j), i >= j, at i (i + 1) / 2 + j), 2 KB a
// block, columns XOR-swizzled by the row pair so that the MFMA operand and result patterns are conflict free without
// padding -- two 128 x 128 lower triangles are 144 KB of the CU's 160 KB.
// 512 threads: waves 0-3 factor the training matrix while waves 4-7 factor the test matrix (same control flow, so
// the workgroup barriers pair up); then each wave solves one 16-column block of V by forward substitution (MFMA for
// the block products, lane shuffles for the 16 x 16 triangular solves -- no stored inverses); then the lift scan.
#include "kernels.h"
#include "tiles.h"

namespace lsspa {

// phase time stamps of workgroup 0 (tools/small_probe.hip builds this file with LSSPA_SMALL_STAMPS)
#ifdef LSSPA_SMALL_STAMPS
__device__ long long g_small_stamps[16];
__device__ long long g_small_cycles[16];
#define SSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { g_small_stamps[i] = wall_clock64(); g_small_cycles[i] = __builtin_amdgcn_s_memtime(); } } while (0)
__device__ long long g_small_helper[16];
__device__ long long g_small_fstamp[4];
#define HSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 64) g_small_helper[i] = __builtin_amdgcn_s_memtime(); } while (0)
__device__ long long g_reg_stamps[2][12];
__device__ long long g_reg_chol[2][8][4];
#define CSTAMP(kb, i) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) g_reg_chol[threadIdx.x >> 6][kb][i] = __builtin_amdgcn_s_memtime(); } while (0)
#define RSTAMP(i) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) g_reg_stamps[threadIdx.x >> 6][i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RSTAMP(i) do { } while (0)
#define CSTAMP(kb, i) do { } while (0)
#define SSTAMP(i) do { } while (0)
#define HSTAMP(i) do { } while (0)
#endif

namespace {

__device__ __forceinline__ int sw(int r, int c) { return r * 16 + (c ^ ((r >> 1) << 1)); }
__device__ __forceinline__ int tri_blk(int i, int j) { return i * (i + 1) / 2 + j; }   // i >= j

constexpr int SINV_LD = 17;

// One wave: factor the 16 x 16 diagonal block blk (swizzled, lower part valid) in place -- L below and on the
// diagonal, the strictly lower part of L^-1 mirrored above it -- and write its inverse to s_inv (16 x SINV_LD) and
// 1 / L[i][i] to s_rd[0..15].  The elimination itself runs on the matrix pipe in accumulator layout (tiles.h:
// factor16_acc, round 3; the lane-permute sweep it replaces was ~4 us of every block step).
__device__ __forceinline__ void wave_factor16_sw(double* blk, double* s_inv, double* s_rd, const double* d0,
                                                 double piv_tol, int lane, int& bad) {
  const int l15 = lane & 15, l4 = lane >> 4;
  d4 t, y;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = acc_row(l4, r);
    t[r] = (l15 <= row) ? blk[sw(row, l15)] : blk[sw(l15, row)];   // upper part: mirrored (what is stored there is ignored)
    y[r] = (row == l15) ? 1.0 : 0.0;
  }
  const double tol_lane = piv_tol * d0[l15];
#ifdef LSSPA_SMALL_STAMPS
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) g_small_fstamp[0] = __builtin_amdgcn_s_memtime();
#endif
  factor16_acc<double>(t, y, tol_lane, lane, bad);
#ifdef LSSPA_SMALL_STAMPS
  asm volatile("" : "+v"(t), "+v"(y));
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) g_small_fstamp[1] = __builtin_amdgcn_s_memtime();
#endif
  double dj;
  const bool holds = acc_diag<double>(t, l15, l4, dj);
  const double rs_mine = fast_rsqrt<double>(dj);                  // 1 / L[j][j]: one reciprocal square root per lane
  if (holds) s_rd[l15] = rs_mine;
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = acc_row(l4, r);
    const double rs_row = s_rd[row];
    const double xv = (l15 <= row) ? y[r] * rs_row : 0.0;         // (L^-1)[row][l15]
    // t holds (row, col = l15): for col >= row the unscaled L[col][row]; L[row][row] = pivot / sqrt(pivot)
    if (l15 >= row) blk[sw(l15, row)] = t[r] * rs_row;
    s_inv[row * SINV_LD + l15] = xv;
    // the strictly lower part of L^-1 lives at the mirrored positions of the block (its diagonal is s_rd);
    // position (col, row), col < row, is nobody's L entry
    if (l15 < row) blk[sw(l15, row)] = xv;
  }
}

// element (m, k) of the inverse of a factored diagonal block (see wave_factor16_sw)
__device__ __forceinline__ double inv_elem(const double* blk, const double* rd, int m, int k) {
  const double off = blk[sw(min(m, k), max(m, k))];
  return (k < m) ? off : (k == m ? rd[m] : 0.0);
}

// 16 x 16 x 16 products of one wave on up to NT tiles at once (independent accumulators: the MFMA chains overlap):
//   T[q] -= A[q] B[q]^T  with A[q], B[q] row blocks of the current panel.  The tiles are given as element OFFSETS into
// the matrix M, not as pointers: an array of pointers loses the LDS address space, and every access through it
// becomes a flat_load / flat_store (found in round 3: the helpers' share of a block step took twice the factoring
// wave's).
template <int NT>
__device__ __forceinline__ void trailing_tiles(double* M, const int (&To)[NT], const int (&Ao)[NT], const int (&Bo)[NT],
                                               int n, int l15, int l4) {
  // every operand fragment and every element of the tiles to be updated is fetched before the first product: one
  // LDS round trip for the batch instead of one in front of each of its 4 NT matrix instructions
  double av[NT][4], bv[NT][4], tv[NT][4];
#pragma unroll
  for (int q = 0; q < NT; ++q) {
    const int qq = (q < n) ? q : 0;        // a short batch re-reads tile 0 (uniform; nothing of it is stored)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      av[q][kk] = M[Ao[qq] + sw(l15, 4 * kk + l4)];
      bv[q][kk] = M[Bo[qq] + sw(l15, 4 * kk + l4)];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) tv[q][r] = M[To[qq] + sw(acc_row(l4, r), l15)];
  }
  d4 o[NT];
#pragma unroll
  for (int q = 0; q < NT; ++q) o[q] = d4_zero();
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
#pragma unroll
    for (int q = 0; q < NT; ++q) o[q] = mfma(av[q][kk], bv[q][kk], o[q]);
#pragma unroll
  for (int q = 0; q < NT; ++q)
    if (q < n) {
#pragma unroll
      for (int r = 0; r < 4; ++r) M[To[q] + sw(acc_row(l4, r), l15)] = tv[q][r] - o[q][r];
    }
}

}  // namespace

__global__ __launch_bounds__(512) void small_p_kernel(SmallArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int nb = a.nb, p = a.p;
  const int ntri = nb * (nb + 1) / 2;
  double* const M0 = smem;                     // training matrix -> L -> lift terms
  double* const M1 = M0 + ntri * 256;          // test matrix -> L_t -> V
  double* const s_inv = M1 + ntri * 256;       // [2][16 * SINV_LD]: inverse of the current diagonal block
  double* const s_rdb = s_inv + 2 * 16 * SINV_LD;   // [2][128]: 1 / L[i][i]
  double* const s_d0 = s_rdb + 256;            // [2][128]: permuted diagonals before any update (pivot scale)
  double* const s_z = s_d0 + 256;              // [128]
  double* const s_y = s_z + 128;               // [128]
  int32_t* const s_perm = reinterpret_cast<int32_t*>(s_y + 128);   // [128]
  __shared__ int s_bad;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // the wave index as the scalar it is: the phase
                                                             // structure below branches on it all the time
  const int half = wv >> 2, w = wv & 3, t2 = tid & 255;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int ord = blockIdx.x;
  const int32_t* perm = a.perms + (int64_t)(a.fwd_only ? ord >> 1 : ord) * p;
  const bool backwards = a.fwd_only && (ord & 1);

  SSTAMP(0);
  if (tid < 128) s_perm[tid] = (tid < p) ? perm[backwards ? p - 1 - tid : tid] : 0;
  if (tid == 0) s_bad = 0;
  __syncthreads();
  SSTAMP(1);

  // ---- permuted gather of both matrices (lower blocks), augmented row p, identity below ----------------------
  {
    double* const M = half ? M1 : M0;
    const double* __restrict__ S = a.S[half];
    const double* __restrict__ sv = a.s[half];
    const double aug = a.aug[half];
    const int r = t2 >> 4, c = t2 & 15;
    const int swrc = sw(r, c);
    // every index this thread needs, then every element: all of a thread's (up to 36) scattered 8-byte loads are in
    // flight together -- one round trip to the L2 instead of one per four elements
    int pr[8], pc[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      pr[b] = s_perm[min(16 * b + r, p - 1)];
      pc[b] = s_perm[min(16 * b + c, p - 1)];
    }
    double gv[36], av[8];
#pragma unroll
    for (int bi = 0; bi < 8; ++bi)
#pragma unroll
      for (int bj = 0; bj <= bi; ++bj)
        if (bi < nb) gv[bi * (bi + 1) / 2 + bj] = S[(int64_t)pr[bi] * a.ld_src + pc[bj]];
#pragma unroll
    for (int bj = 0; bj < 8; ++bj)
      if (bj < nb) av[bj] = sv[pc[bj]];
#pragma unroll
    for (int bi = 0; bi < 8; ++bi)
#pragma unroll
      for (int bj = 0; bj <= bi; ++bj)
        if (bi < nb) {
          const int i = 16 * bi + r, j = 16 * bj + c;
          double v;
          if (i < p) v = (j <= i) ? gv[bi * (bi + 1) / 2 + bj] : 0.0;
          else if (i == p) v = (j < p) ? av[bj] : (j == p ? aug : 0.0);
          else v = (i == j) ? 1.0 : 0.0;
          M[(bi * (bi + 1) / 2 + bj) * 256 + swrc] = v;
        }
    if (t2 < 128) {
      const int i = t2;
      s_d0[half * 128 + i] = (i < p) ? S[(int64_t)s_perm[i] * a.ld_src + s_perm[i]] : (i == p ? aug : 1.0);
    }
  }
  __syncthreads();
  SSTAMP(2);

  // ---- blocked Cholesky of both matrices at once (right-looking, 16 x 16 blocks) ------------------------------
  {
    double* const M = half ? M1 : M0;
    double* const inv = s_inv + half * 16 * SINV_LD;
    double* const rd = s_rdb + half * 128;
    const double* const d0 = s_d0 + half * 128;
    int bad = 0;
    // the wave that factors the diagonal blocks: wave 0 of the training half, wave 1 of the test half -- waves 0 and 4
    // would sit on the same SIMD and take turns on its issue slots
    const int fw = half;
    if (w == fw) wave_factor16_sw(M, inv, rd, d0, a.piv_tol, lane, bad);
    SSTAMP(8);
    // the factoring waves' pivot chains are dependent matrix instructions; the helpers' trailing products are not:
    // priority lets each link of the chain issue when it is ready
    if (w == fw) __builtin_amdgcn_s_setprio(3);
    __syncthreads();
    for (int kb = 0; kb < nb; ++kb) {
      // panel: L[ib][kb] = T[ib][kb] Ld^-T, in place
      for (int ib = kb + 1 + w; ib < nb; ib += 4) {
        double* const Tb = M + tri_blk(ib, kb) * 256;
        d4 o = d4_zero();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int kx = 4 * kk + l4;
          o = mfma(Tb[sw(l15, kx)], inv[l15 * SINV_LD + kx], o);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) Tb[sw(acc_row(l4, r), l15)] = o[r];
      }
      __syncthreads();
      if (kb == 0) SSTAMP(9);
      if (kb + 1 == nb) break;
      // trailing update T[ib][jb] -= L[ib][kb] L[jb][kb]^T (kb < jb <= ib) with look-ahead: the factoring wave takes
      // the next diagonal tile alone and factors it at once, the other three share the rest four tiles at a time
      if (w == fw) {
        const int To[1] = {tri_blk(kb + 1, kb + 1) * 256}, Ao[1] = {tri_blk(kb + 1, kb) * 256};
        trailing_tiles<1>(M, To, Ao, Ao, 1, l15, l4);
        __builtin_amdgcn_wave_barrier();
        if (kb == 0) SSTAMP(10);
        wave_factor16_sw(M + To[0], inv, rd + 16 * (kb + 1), d0 + 16 * (kb + 1), a.piv_tol, lane, bad);
        if (kb == 0) SSTAMP(11);
      } else {
        const int me = (w + 3 - fw) % 4 - 0;      // 0, 1, 2 among the three helpers
        if (kb == 0) HSTAMP(0);
        [[maybe_unused]] int hs = 1;
        int To[4], Ao[4], Bo[4];
        int have = 0, tcount = 0;
        for (int ib = kb + 2; ib < nb; ++ib)       // row kb + 1 holds (kb+1, kb+1) only: the factoring wave's
          for (int jb = kb + 1; jb <= ib; ++jb) {
            if ((tcount++ % 3) != me) continue;
            To[have] = tri_blk(ib, jb) * 256;
            Ao[have] = tri_blk(ib, kb) * 256;
            Bo[have] = tri_blk(jb, kb) * 256;
            if (++have == 4) {
              trailing_tiles<4>(M, To, Ao, Bo, 4, l15, l4);
              if (kb == 0) { HSTAMP(hs); ++hs; }
              have = 0;
            }
          }
        if (have) trailing_tiles<4>(M, To, Ao, Bo, have, l15, l4);
        if (kb == 0) HSTAMP(hs);
      }
      __syncthreads();
      if (kb == 0) SSTAMP(12);
    }
    if (bad && lane == 0) s_bad = 1;
  }

  SSTAMP(3);
  // ---- z = row p of L, y~ = row p of L_t (V overwrites L_t below) ---------------------------------------------
  {
    const int pb = p >> 4, pr = p & 15;
    if (tid < 128) {
      const int j = tid;
      s_z[j] = (j < p) ? M0[tri_blk(pb, j >> 4) * 256 + sw(pr, j & 15)] : 0.0;
      s_y[j] = (j < p) ? M1[tri_blk(pb, j >> 4) * 256 + sw(pr, j & 15)] : 0.0;
    }
  }
  __syncthreads();

  SSTAMP(4);
  // ---- V = L^-1 L_t: wave cb solves the 16-column block cb, top down, in place of L_t.  The solved blocks V[k][cb] stay
  // in the wave's registers: an accumulator tile is, as it stands, the B operand of the products that sum over its row
  // index (tiles.h), so the products of a row need only the fragments of L[i][.] from LDS -- fetched all at once, then
  // the matrix instructions back to back (one LDS round trip in front of every product made this phase 9 us of 53).
  if (wv < nb) {
    const int cb = wv;
    d4 vreg[8];
#pragma unroll
    for (int ii = 0; ii < 8; ++ii) {            // row block i = cb + ii
      const int i = cb + ii;
      if (i >= nb) break;                        // uniform
      double af[7][4];                           // fragments of L[i][cb + kk2], kk2 < ii
#pragma unroll
      for (int kk2 = 0; kk2 < 7; ++kk2)
        if (kk2 < ii) {
          const double* const Ab = M0 + tri_blk(i, cb + kk2) * 256;
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) af[kk2][kk] = Ab[sw(l15, 4 * kk + l4)];
        }
      double* const Tb = M1 + tri_blk(i, cb) * 256;
      double t[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) t[r] = Tb[sw(acc_row(l4, r), l15)];
      const double* const Ld = M0 + tri_blk(i, i) * 256;
      const double* const rd = s_rdb + 16 * i;
      double ie[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) ie[r] = inv_elem(Ld, rd, l15, 4 * r + l4);
      d4 acc = d4_zero();
#pragma unroll
      for (int kk2 = 0; kk2 < 7; ++kk2)
        if (kk2 < ii) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) acc = mfma(af[kk2][kk], vreg[kk2][kk], acc);
        }
#pragma unroll
      for (int r = 0; r < 4; ++r) t[r] -= acc[r];
      if (ii == 0) {   // L_t's diagonal block is lower triangular; its upper positions hold other data by now
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (l15 > acc_row(l4, r)) t[r] = 0.0;
      }
      // X = L_d^-1 t: the registers of t are the B operand of the k-steps {4 r .. 4 r + 3}; L_d^-1 sits in the diagonal
      // block's upper triangle and in s_rdb (wave_factor16_sw)
      d4 x = d4_zero();
#pragma unroll
      for (int r = 0; r < 4; ++r) x = mfma(ie[r], t[r], x);
      vreg[ii] = x;
#pragma unroll
      for (int r = 0; r < 4; ++r) Tb[sw(acc_row(l4, r), l15)] = x[r];
    }
  }
  __syncthreads();
  SSTAMP(5);

  // ---- lift terms: w[j][c] = V[j][c] (2 y~_c - N_j - N_{j-1}),  N_j = sum_{k <= j} z_k V[k][c], down column c.  Four
  // threads per column (adjacent lanes), each a quarter of the rows c .. p - 1: the partial sums of the quarters first,
  // then each quarter's scan from the sum of the quarters above it -- a quarter of the dependent chain.
  {
    const int c = tid >> 2, sq = tid & 3;
    const int cbk = c >> 4, cc = c & 15;
    const int len = (c < p) ? p - c : 0, per = (len + 3) >> 2;
    const int j0 = c + min(sq * per, len), j1 = c + min((sq + 1) * per, len);
    double part = 0.0;
    for (int j = j0; j < j1; ++j) part = fma(s_z[j], M1[tri_blk(j >> 4, cbk) * 256 + sw(j & 15, cc)], part);
    // exclusive prefix over the four quarters of the column
    const double p1 = __shfl_up(part, 1, 4), p2 = __shfl_up(part, 2, 4), p3 = __shfl_up(part, 3, 4);
    double N = (sq >= 1 ? p1 : 0.0) + (sq >= 2 ? p2 : 0.0) + (sq >= 3 ? p3 : 0.0);
    const double y2 = 2.0 * s_y[min(c, 127)];
    for (int j = j0; j < j1; ++j) {
      const int off = tri_blk(j >> 4, cbk) * 256 + sw(j & 15, cc);
      const double v = M1[off];
      const double Nn = fma(s_z[j], v, N);
      M0[off] = v * (y2 - Nn - N);      // L is dead: its storage takes the terms
      N = Nn;
    }
  }
  __syncthreads();
  SSTAMP(6);

  // ---- lift_j = z_j / ||y||^2 * sum_{c <= j} w[j][c]; four threads per row ------------------------------------
  {
    const int j = tid >> 2, q = tid & 3;
    double sacc = 0.0;
    if (j < p)
      for (int c = q; c <= j; c += 4) sacc += M0[tri_blk(j >> 4, c >> 4) * 256 + sw(j & 15, c & 15)];
    sacc += __shfl_xor(sacc, 1);
    sacc += __shfl_xor(sacc, 2);
    if (j < p && q == 0) {
      const double lift = s_z[j] * sacc / a.y_norm_sq;
      double* dst = a.lifts + (int64_t)(ord / a.per_sample) * p + s_perm[j];
      if (a.per_sample == 2) atomicAdd(dst, 0.5 * lift);   // the pair's two terms commute: order-independent sum
      else *dst = lift;
    }
  }
  if (tid == 0 && s_bad) atomicOr(a.info, 1);
  SSTAMP(7);
}

// ---- the register-resident form (round 3) ---------------------------------------------------------------------
// What bounds the kernel above is not work but the number of pivot chains a CU has in flight: LDS holds two matrices,
// so two chains, and every block step of each is a workgroup affair (barriers, LDS round trips between the phases).
// The registers of a CU are 512 KB against the LDS's 160: here ONE WAVE owns a matrix outright -- its lower 16 x 16
// blocks, transposed (U = L^T), in MFMA accumulator layout, 8 registers a block, 224 at p + 1 <= 112 -- and runs the
// whole right-looking factorisation on them without a barrier or an LDS operand:
//   * the diagonal block is factored where it sits (tiles.h: factor16_acc);
//   * a panel block  U[kb][ib] = L_d^-1 T^T[kb][ib]  takes its B operand from the block's own registers (an accumulator
//     tile is the B operand of a product over its row index) and L_d^-1 from a 2 KB scratch (the one transposition);
//   * a trailing block  T^T[jb][ib] -= U[kb][jb]^T U[kb][ib]  takes BOTH operands from registers: the accumulator tile of
//     U[kb][jb], read as an A operand, is U[kb][jb]^T with the same row index as the summation index.
// A workgroup is two waves, one per matrix of an ordering, each alone on its SIMD with the 512-register budget of a
// single-wave SIMD; 79 KB of LDS (L_t, then V; the training matrix's block inverses; vectors) lets two workgroups share a
// CU: four chains in flight instead of two, and no workgroup barrier inside the factorisation.  Afterwards the test
// wave hands L_t over through LDS, the training wave solves V = L^-1 L_t column block by column block with L still in
// its registers, and both waves run the lift scan, a thread per column and then per row.
// p + 1 <= 112 (seven block rows); the eight-block-row case keeps the LDS kernel above (its 288 registers of matrix
// would fit, its 93 KB of LDS would leave one workgroup per CU).
namespace {

__device__ __forceinline__ constexpr int ublk(int kb, int ib) { return ib * (ib + 1) / 2 + kb; }   // kb <= ib

// the region of L_t / V doubles as the gather's staging area: two waves x 16 source rows of up to 16 nb + 2 elements
constexpr int small_reg_vregion(int nb) {
  return (nb * (nb + 1) / 2) * 256 > 32 * (16 * nb + 2) ? (nb * (nb + 1) / 2) * 256 : 32 * (16 * nb + 2);
}
constexpr int small_reg_lds_doubles(int nb) { return small_reg_vregion(nb) + (nb + 2) * 256 + 32 + 128 + 128 + 256; }

}  // namespace

// waves per SIMD the register budget is cut for: one at six or seven block rows (up to 512 registers), two at four or
// five (256), three below (168); LDS lets as many workgroups in
constexpr int small_reg_waves(int nb) { return nb <= 3 ? 3 : (nb <= 4 ? 2 : 1); }

template <int NB>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(small_reg_waves(NB), small_reg_waves(NB))))
void small_reg_kernel(SmallArgs a) {
  constexpr int NTRI = NB * (NB + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int VREG = small_reg_vregion(NB);
  double* const s_V = smem;                         // [NTRI][256]: the gather's staging area, L_t, V, the lift terms
  double* const s_linv = s_V + VREG;                // [NB][256]: -L_d^-1 of the training matrix's diagonal blocks
  double* const s_scr = s_linv + NB * 256;          // [2][256]: per wave, the current block inverse on its way to operand form
  double* const s_rd = s_scr + 512;                 // [2][16]: 1 / L[i][i] of the current diagonal block
  double* const s_z = s_rd + 32;                    // [128]
  double* const s_y = s_z + 128;                    // [128]
  double* const s_tol = s_y + 128;                  // [2][128]: pivot thresholds (fourteen registers for the whole
                                                    // factorisation otherwise, in a kernel that has none to spare)
  int32_t* const s_perm = reinterpret_cast<int32_t*>(s_tol + 256);   // [128]
  __shared__ int s_bad, s_flag;                     // s_flag: column blocks of V the training wave has finished

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0: training matrix, 1: test matrix
  const int l15 = lane & 15, l4 = lane >> 4;
  const int p = a.p, pr = p & 15;                   // p = 16 (NB - 1) + pr: the augmented row sits in the last block row
  const int ord = blockIdx.x;
  const int32_t* perm = a.perms + (int64_t)(a.fwd_only ? ord >> 1 : ord) * p;
  const bool backwards = a.fwd_only && (ord & 1);
  RSTAMP(0);
  {
    const int src = (tid < p) ? perm[backwards ? p - 1 - tid : tid] : 0;
    s_perm[tid] = src;
    s_z[tid] = a.s[0][src];          // the permuted right-hand sides g_pi, h_pi (row p of the two matrices) until the
    s_y[tid] = a.s[1][src];          // factorisations have produced z and y~, which take their place
  }
  if (tid == 0) {
    s_bad = 0;
    s_flag = 0;
  }
  __syncthreads();
  RSTAMP(8);

  // ---- permuted gather straight into the accumulator layout: register r of block (kb, ib) holds element
  //      (row 16 kb + l4 + 4 r, column 16 ib + l15) of U = the transposed lower triangle.
  // Picking the elements out of the source one by one costs a cache line each (7168 lines a matrix: with four matrices
  // per CU in flight that alone was 29 k cycles of the first version's 167 k -- the L1 passes a line per clock).  The
  // sixteen source rows of a block column are fetched whole instead (coalesced, seven lines a row) into the LDS region
  // that will hold L_t later, and the lanes pick from there. ---------------------------------------------------------
  const double* __restrict__ S = a.S[wv];
  const double aug = a.aug[wv];
  const double* const svp = wv ? s_y : s_z;         // s[perm[j]]
  d4 U[NTRI];
  double* const tol = s_tol + wv * 128;
  {
    const int RS = (p + 3) & ~1;                    // row stride of the staging area (even: 16-byte rows)
    double* const stage = s_V + wv * (VREG / 2);    // 16 RS <= VREG / 2 by the definition of VREG
    // lanes past the end of a row repeat its last pair (same value to the same place): no predicated regions
    const int col2 = min(2 * lane, (p - 1) & ~1);
    const int ld = (int)a.ld_src;                   // element offsets fit 32 bits (p_pad^2); a local, or every load re-reads
                                                    // the argument from memory
    int pk[NB][4], pi[NB];
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
      pi[kb] = s_perm[min(16 * kb + l15, p - 1)];
#pragma unroll
      for (int r = 0; r < 4; ++r) pk[kb][r] = s_perm[min(16 * kb + l4 + 4 * r, p - 1)];
    }
    typedef double d2 __attribute__((ext_vector_type(2)));
    constexpr int AHEAD = small_reg_waves(NB) == 1 ? 2 : 1;    // block columns in flight (registers permitting)
    d2 rows[AHEAD][16];
    // source row of position i as a scalar: lane i & 63 of one of two registers (a uniform LDS read per row and a wait
    // for each made a fetch sixteen LDS round trips)
    const int perm_lo = s_perm[min(lane, p - 1)], perm_hi = s_perm[min(64 + lane, p - 1)];
    const double* const Scol = S + col2;
    auto fetch = [&](const int ib) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int i = 16 * ib + q;          // static after unrolling
        const int src = __builtin_amdgcn_readlane(i < 64 ? perm_lo : perm_hi, i & 63);   // positions >= p repeat row p - 1
        rows[ib % AHEAD][q] = *reinterpret_cast<const d2*>(Scol + src * ld);
      }
    };
    fetch(0);
    if (AHEAD > 1 && NB > 1) fetch(1);
    RSTAMP(9);
    double d0[NB];
#pragma unroll
    for (int ib = 0; ib < NB; ++ib) {
#pragma unroll
      for (int q = 0; q < 16; ++q) *reinterpret_cast<d2*>(stage + q * RS + col2) = rows[ib % AHEAD][q];
      if (ib + AHEAD < NB) fetch(ib + AHEAD);       // the next block columns' rows travel while this one is picked
      __builtin_amdgcn_wave_barrier();
      const double* const mine = stage + l15 * RS;
#pragma unroll
      for (int kb = 0; kb <= ib; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) U[ublk(kb, ib)][r] = mine[pk[kb][r]];
      d0[ib] = mine[pi[ib]];
      __builtin_amdgcn_wave_barrier();
    }
    RSTAMP(10);
    // the last block column holds the augmented row (column index p of U) and the identity padding behind it
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
      double svk[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) svk[r] = svp[min(16 * kb + l4 + 4 * r, p - 1)];
      if (kb < NB - 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double g = U[ublk(kb, NB - 1)][r];
          U[ublk(kb, NB - 1)][r] = (l15 < pr) ? g : (l15 == pr ? svk[r] : 0.0);
        }
      } else {
        const double svi = svp[min(16 * (NB - 1) + l15, p - 1)];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kr = l4 + 4 * r, hi = max(l15, kr), lo = min(l15, kr);
          const double g = U[ublk(kb, NB - 1)][r];
          const double edge = (lo < pr) ? (lo == kr ? svk[r] : svi) : aug;
          U[ublk(kb, NB - 1)][r] = (hi < pr) ? g : (hi == pr ? edge : (l15 == kr ? 1.0 : 0.0));
        }
      }
    }
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
      const int i = 16 * kb + l15;
      if (l4 == 0) tol[i] = a.piv_tol * ((i < p) ? d0[kb] : (i == p ? aug : 1.0));
    }
  }
  __syncthreads();           // the staging area is the other wave's too from its hand-over on
#ifdef LSSPA_SMALL_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  RSTAMP(1);
  // ---- blocked Cholesky in registers ------------------------------------------------------------------------------
  int bad = 0;
  double* const scr = s_scr + wv * 256;
  double* const rd = s_rd + wv * 16;
#pragma unroll
  for (int kb = 0; kb < NB; ++kb) {
    d4 t = U[ublk(kb, kb)], y;
#pragma unroll
    for (int r = 0; r < 4; ++r) y[r] = (acc_row(l4, r) == l15) ? 1.0 : 0.0;
    CSTAMP(kb, 0);
    factor16_acc<double>(t, y, tol[16 * kb + l15], lane, bad);
#ifdef LSSPA_SMALL_STAMPS
    asm volatile("" : "+v"(t), "+v"(y));
#endif
    CSTAMP(kb, 1);
    double dj;
    const bool holds = acc_diag<double>(t, l15, l4, dj);
    const double rs_mine = fast_rsqrt<double>(dj);             // 1 / L[j][j]
    if (holds) rd[l15] = rs_mine;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = acc_row(l4, r);
      const double rs_row = rd[row];
      // t: (row, col) for col >= row is the unscaled L[col][row] -- row `row` of U_d; y: (L^-1)[row][col] L[row][row]
      U[ublk(kb, kb)][r] = t[r] * rs_row;
      const double xv = (l15 <= row) ? y[r] * rs_row : 0.0;    // (L_d^-1)[row][l15]
      scr[sw(row, l15)] = xv;
      if (wv == 0) s_linv[kb * 256 + sw(row, l15)] = -xv;      // the V solve multiplies by -L_d^-1
    }
    __builtin_amdgcn_wave_barrier();
    if (kb + 1 < NB) {
      double la[4];                                            // A operand: (L_d^-1)[l15][l4 + 4 r]
#pragma unroll
      for (int r = 0; r < 4; ++r) la[r] = scr[sw(l15, l4 + 4 * r)];
      __builtin_amdgcn_wave_barrier();                         // the scratch is rewritten in the next block step
      CSTAMP(kb, 2);
      // panel: U[kb][ib] = L_d^-1 T^T[kb][ib]; the block's registers are the B operand
#pragma unroll
      for (int ib = kb + 1; ib < NB; ++ib) {
        d4 o = d4_zero();
#pragma unroll
        for (int r = 0; r < 4; ++r) o = mfma(la[r], U[ublk(kb, ib)][r], o);
        U[ublk(kb, ib)] = o;
      }
      // trailing blocks: T^T[jb][ib] -= U[kb][jb]^T U[kb][ib], both operands as they sit in the registers
#pragma unroll
      for (int jb = kb + 1; jb < NB; ++jb) {
        double na[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) na[r] = -U[ublk(kb, jb)][r];
#pragma unroll
        for (int ib = jb; ib < NB; ++ib)
#pragma unroll
          for (int r = 0; r < 4; ++r) U[ublk(jb, ib)] = mfma(na[r], U[ublk(kb, ib)][r], U[ublk(jb, ib)]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (bad && lane == 0) s_bad = 1;
  RSTAMP(2);

  // ---- z = row p of L, y~ = row p of L_t: column pr of the last block column of U; L_t into LDS -------------------
  {
    double* const dst = wv ? s_y : s_z;
    if (l15 == pr) {
#pragma unroll
      for (int kb = 0; kb < NB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[16 * kb + l4 + 4 * r] = U[ublk(kb, NB - 1)][r];
    }
    if (wv == 1) {
#pragma unroll
      for (int kb = 0; kb < NB; ++kb)
#pragma unroll
        for (int ib = kb; ib < NB; ++ib)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int c = l4 + 4 * r;       // L_t[ib][kb] (row l15, column c) = U_t[kb][ib] (row c, column l15)
            const double v = U[ublk(kb, ib)][r];
            s_V[ublk(kb, ib) * 256 + sw(l15, c)] = (ib == kb && c > l15) ? 0.0 : v;
          }
    }
  }
  __syncthreads();
  RSTAMP(3);

  // ---- V = L^-1 L_t by the training wave, column block by column block, top down; L comes from its registers:
  //      the accumulator tile of U[k][i], read as an A operand, is L[i][k] with the summation index on the rows.
  //      The test wave follows one column block behind with the lift scan (a flag in LDS, no barrier): its work hides
  //      behind the solve instead of following it (18.6 k of 133 k cycles when both waves scanned afterwards). ----------
  if (wv == 0) {
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
      d4 vcol[NB];
#pragma unroll
      for (int i = cb; i < NB; ++i) {
        double* const Tb = s_V + ublk(cb, i) * 256;
        d4 w;                                   // sum_k L[i][k] V[k][cb] - L_t[i][cb]
#pragma unroll
        for (int r = 0; r < 4; ++r) w[r] = -Tb[sw(acc_row(l4, r), l15)];
        double nl[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) nl[r] = s_linv[i * 256 + sw(l15, l4 + 4 * r)];
#pragma unroll
        for (int k = cb; k < i; ++k)
#pragma unroll
          for (int r = 0; r < 4; ++r) w = mfma(U[ublk(k, i)][r], vcol[k][r], w);
        d4 x = d4_zero();
#pragma unroll
        for (int r = 0; r < 4; ++r) x = mfma(nl[r], w[r], x);
        vcol[i] = x;
#pragma unroll
        for (int r = 0; r < 4; ++r) Tb[sw(acc_row(l4, r), l15)] = x[r];
        __builtin_amdgcn_sched_barrier(0);      // keep the next blocks' LDS reads from climbing up here: with 224 registers
                                                // of matrix there is no room for them (scratch otherwise; fetching the next
                                                // block's operands ahead by hand was measured: no gain, more scratch)
      }
      if (lane == 0) __hip_atomic_store(&s_flag, cb + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    RSTAMP(4);
  } else {
    // lift terms  w[j][c] = V[j][c] (2 y~_c - N_j - N_{j-1}),  N_j = sum_{k <= j} z_k V[k][c]  down column c, then
    // lift_j = z_j / ||y||^2 * sum_{c <= j} w[j][c].  A lane takes column cc = lane & 15 of the block and rows
    // 4 sq .. 4 sq + 3 (sq = lane / 16) of every block row: a block row's scan is four partial sums, an exclusive prefix
    // over the four lane groups, four terms.  V is exactly zero above the diagonal (so are its terms), and the rows
    // from p on come last in a column: no range tests.  Row sums: lane = row (two passes of 64), added up per column block.
    typedef double d2 __attribute__((ext_vector_type(2)));
    const int cc = l15, sq = l4;
    double racc[2] = {0.0, 0.0};
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
      while (__hip_atomic_load(&s_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= cb) __builtin_amdgcn_s_sleep(8);
      const double y2 = 2.0 * s_y[16 * cb + cc];
      double base = 0.0;
#pragma unroll
      for (int jb = cb; jb < NB; ++jb) {
        double* const B = s_V + ublk(cb, jb) * 256;
        double v[4], z[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v[i] = B[sw(4 * sq + i, cc)];
          z[i] = s_z[16 * jb + 4 * sq + i];
        }
        double q = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) q = fma(z[i], v[i], q);
        const double q1 = __shfl(q, lane - 16), q2 = __shfl(q, lane - 32), q3 = __shfl(q, lane - 48);
        const double pre = (sq >= 1 ? q1 : 0.0) + (sq >= 2 ? q2 : 0.0) + (sq >= 3 ? q3 : 0.0);
        const double tot = __shfl(pre + q, 48 + cc);     // the block row's whole sum, from the last lane group
        double N = base + pre;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const double Nn = fma(z[i], v[i], N);
          B[sw(4 * sq + i, cc)] = v[i] * (y2 - Nn - N);
          N = Nn;
        }
        base += tot;
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const int j = lane + 64 * pass, jb = j >> 4, jr = j & 15, swz = (jr >> 1) << 1;
        if (16 * cb < 64 * (pass + 1) && 64 * pass < 16 * NB) {       // static: this pass has rows of the column block
          const bool live = jb >= cb && jb < NB;
          const int jbx = live ? jb : cb;
          const double* const B = s_V + (jbx * (jbx + 1) / 2 + cb) * 256 + 16 * jr;
          double sacc = 0.0;
#pragma unroll
          for (int qd = 0; qd < 8; ++qd) {
            const d2 t2 = *reinterpret_cast<const d2*>(B + ((2 * qd) ^ swz));
            sacc += t2[0] + t2[1];
          }
          racc[pass] += live ? sacc : 0.0;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    RSTAMP(4);
    double lsum = 0.0;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int j = lane + 64 * pass;
      if (j < p) {
        const double lift = s_z[j] * racc[pass] / a.y_norm_sq;
        double* dst = a.lifts + (int64_t)(ord / a.per_sample) * p + s_perm[j];
        if (a.per_sample == 2) atomicAdd(dst, 0.5 * lift);   // the pair's two terms commute: order-independent sum
        else *dst = lift;
        lsum += lift;
      }
    }
    // the ordering's lifts telescope to the full model's R^2 (sum_check_kernel in k_lift.hip makes this check per
    // sample by a launch of its own; here the wave holds all p lifts, and a dependent launch is 4 % of a p = 100 group)
    if (a.sum_tol >= 0.0) {
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) lsum += __shfl_xor(lsum, o, 64);
      if (lane == 0) {
        double dev = fabs(lsum - a.r2);
        if (!(dev == dev)) dev = __longlong_as_double(0x7ff0000000000000ll);
        if (!(dev <= a.sum_tol)) atomicOr(a.info, 8);                  // LSSPA_INFO_SUM
        if (dev > a.sum_quiet)       // non-negative doubles order like their bit patterns
          atomicMax(reinterpret_cast<unsigned long long*>(a.info + 2), (unsigned long long)__double_as_longlong(dev));
      }
    }
  }
  if (tid == 0 && s_bad) atomicOr(a.info, 1);
  RSTAMP(7);
}

size_t small_p_lds_bytes(int nb) {
  const size_t ntri = (size_t)nb * (nb + 1) / 2;
  return (2 * ntri * 256 + 2 * 16 * SINV_LD + 256 + 256 + 128 + 128) * sizeof(double) + 128 * sizeof(int32_t);
}

bool small_p_eligible(int p) { return p >= 1 && p + 1 <= 128; }

// does the kernel launch_small_p picks check the orderings' sums itself (a.sum_tol >= 0)?
bool small_p_checks_sum(const SmallArgs& a) { return a.variant == 0 && a.nb <= 7; }

template <int NB>
static hipError_t launch_small_reg(const SmallArgs& a, hipStream_t st) {
  const size_t bytes = (size_t)small_reg_lds_doubles(NB) * sizeof(double) + 128 * sizeof(int32_t);
  static DynLdsGrant grant;   // per device, per instance
  hipError_t e = grant.ensure(reinterpret_cast<const void*>(small_reg_kernel<NB>), bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(small_reg_kernel<NB>, dim3(a.n_ord), dim3(128), bytes, st, a);
  return hipGetLastError();
}

hipError_t launch_small_p(const SmallArgs& a, hipStream_t st) {
  if (!small_p_eligible(a.p) || a.nb != (a.p + 1 + 15) / 16 || a.n_ord < 1 || (a.per_sample != 1 && a.per_sample != 2) ||
      (a.n_ord % a.per_sample) != 0 || !a.S[0] || !a.S[1] || !a.perms || !a.lifts || (a.fwd_only && a.per_sample != 2))
    return hipErrorInvalidValue;
  if (a.variant == 0) {       // the register-resident form wherever the matrix fits its wave (see small_reg_kernel)
    switch (a.nb) {
      case 1: return launch_small_reg<1>(a, st);
      case 2: return launch_small_reg<2>(a, st);
      case 3: return launch_small_reg<3>(a, st);
      case 4: return launch_small_reg<4>(a, st);
      case 5: return launch_small_reg<5>(a, st);
      case 6: return launch_small_reg<6>(a, st);
      case 7: return launch_small_reg<7>(a, st);
      default: break;
    }
  }
  const size_t bytes = small_p_lds_bytes(a.nb);
  static DynLdsGrant grant;   // per device (a second engine on another GPU sets the attribute there too)
  hipError_t e = grant.ensure(reinterpret_cast<const void*>(small_p_kernel), bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(small_p_kernel, dim3(a.n_ord), dim3(512), bytes, st, a);
  return hipGetLastError();
}

}  // namespace lsspa
