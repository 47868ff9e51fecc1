// Device-side error estimator.
//
// The reference draws 1024 samples of N(0, C_unbiased / n) and reports 0.95-quantiles of |x_a| per
// feature and of ||x||_2 (cvxgrp/ls-spa ls_spa/ls_spa.py:321-341), which costs a p x p Cholesky or,
// on the always-singular sample covariance, an SVD.  With H the n x p matrix of lift vectors and
// Xi ~ N(0, I) of shape 1024 x n,   x = Xi (H - 1 mu^T) / sqrt(n (n - 1))   has exactly that
// covariance, so the draws are one thin GEMM.  In the thin form with the caller's own normals (lsspa_error_draws) Xi
// comes from the host; everything else stays in HBM:
//   draws_kernel     : 64 x 128 tiles of Xi * H on the fp64 MFMA, centring and scaling fused into the
//                      store.  With several GPUs each rank multiplies its own samples' rows of H by
//                      the matching columns of Xi; the partial draws are summed by ONE all-reduce.
//   row_norms_kernel : ||x_d||_2, one wave per draw, fixed summation order
//   quantile_kernel  : one workgroup per feature (+ one for the norms): bitonic sort of 1024 values
//                      in LDS, numpy's default linear-interpolation quantile.
//
// Running form (round 5): the cost of a check must not grow with the number of samples n, or a run of many checks is
// bound by the estimator instead of the sampling (128 checks at p = 1000 drew 1.08e9 host normals).  Xi is made a pure
// function of (seed, sample id k, draw d) -- Philox4x32-10 keyed by the seed, counter = (k, Philox call), Box-Muller on
// the four output words -- so a column of Xi never has to be kept or drawn twice, and
//       D = Xi L   [1024][p]      s = Xi 1   [1024]
// stay in HBM: a chunk of new samples adds  Xi_new L_new  and  Xi_new 1  (xi_fill_kernel + acc_small_kernel), a check
// is  x = (D - s mu^T) / sqrt(n (n - 1))  -- evaluated by the quantile kernels as they read it (one rank) or written
// by running_draws_kernel for the all-reduce (several) -- and the two quantile kernels.  At every check x has, given the lift vectors, exactly the distribution N(0, C_unbiased / n) the reference
// samples from; successive checks reuse the columns of Xi of the samples they share (the reference redraws: its
// checks are independent given the samples, ours are positively correlated -- each one's distribution is the same).
// With several GPUs each rank holds D and s of its own samples; x is linear in them, so ONE all-reduce of the
// per-rank x (the same buffer and call as before) gives every rank the same draws.
#include "kernels.h"
#include "tiles.h"

namespace lsspa {

constexpr int ND = 1024;  // draws, as in the reference

// ---- running form: counter-based normals ------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11): ten rounds of two 32 x 32 -> 64 multiplies and xors, the key bumped by the
// Weyl constants between rounds.  Pinned by the generator's published known-answer vectors (tests/philox_ref.py).
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t m0 = (uint64_t)0xD2511F53u * c0, m1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(m1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)m1;
    const uint32_t n2 = (uint32_t)(m0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)m0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// The two normals Philox call j of sample `id` yields: u1, u2 = the 53 high bits of (w0, w1), (w2, w3) plus a half, over
// 2^53 -- both in (0, 1) --, then Box-Muller.  They are draws 64 b + r and 64 b + r + 32 of the sample, j = 32 b + r
// (r < 32).
__device__ __forceinline__ void xi_pair(uint64_t seed, uint64_t id, uint32_t j, double& z0, double& z1) {
  uint32_t w[4];
  philox4x32_10((uint32_t)id, (uint32_t)(id >> 32), j, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
  const double u1 = ((double)(((uint64_t)(w[0] >> 5) << 26) | (uint64_t)(w[1] >> 6)) + 0.5) * 0x1p-53;
  const double u2 = ((double)(((uint64_t)(w[2] >> 5) << 26) | (uint64_t)(w[3] >> 6)) + 0.5) * 0x1p-53;
  const double rad = sqrt(-2.0 * log(u1));
  double sn, cs;
  sincospi(2.0 * u2, &sn, &cs);
  z0 = rad * cs;
  z1 = rad * sn;
}

// draws = (Xi H - rowsum(Xi) mean^T) * scale: the thin form with the caller's normals (lsspa_error_draws)
__global__ __launch_bounds__(256, 2) void draws_kernel(const double* __restrict__ Xi, int ldxi,
                                                       const double* __restrict__ H, int ldh, int n_pad,
                                                       const double* __restrict__ mean, double scale, int p,
                                                       double* __restrict__ draws, int ldd) {
  __shared__ __attribute__((aligned(16))) double s_rk[64 * RK_LD];
  __shared__ __attribute__((aligned(16))) double s_kc[16 * KC_LD];
  __shared__ double s_rs[64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int d0 = blockIdx.x * 64;    // draw rows
  const int c0 = blockIdx.y * 128;   // feature columns

  d4 acc[4][2];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) acc[x][y] = d4_zero();
  double rs[4] = {0.0, 0.0, 0.0, 0.0};   // row sums of Xi for rows 16 x + l15, this lane's k share

  const double* srcA = Xi + (int64_t)d0 * ldxi;
  const double* srcB = H + c0;
  const int nch = n_pad / KCH;
  RKRegs<double, 64> ra = {};
  KCRegs<double> rb = {};
  rk_load<double, 64>(ra, srcA, ldxi, tid, 64);
  kc_load<double>(rb, srcB, ldh, tid);
  for (int c = 0; c < nch; ++c) {
    __syncthreads();
    rk_store<double, 64>(ra, s_rk, tid);
    kc_store<double>(rb, s_kc, tid);
    __syncthreads();
    if (c + 1 < nch) {
      rk_load<double, 64>(ra, srcA + (c + 1) * KCH, ldxi, tid, 64);
      kc_load<double>(rb, srcB + (int64_t)(c + 1) * KCH * ldh, ldh, tid);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      double av[4], bv[2];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        av[x] = s_rk[(16 * x + l15) * RK_LD + 4 * kk + l4];
        rs[x] += av[x];
      }
#pragma unroll
      for (int y = 0; y < 2; ++y) bv[y] = s_kc[(4 * kk + l4) * KC_LD + 32 * w + 16 * y + l15];
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = mfma(av[x], bv[y], acc[x][y]);
    }
  }
  // row sums: lanes l4 = 0..3 hold disjoint k shares of row 16 x + l15
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    rs[x] += __shfl_xor(rs[x], 16, 64);
    rs[x] += __shfl_xor(rs[x], 32, 64);
    if (w == 0 && l4 == 0) s_rs[16 * x + l15] = rs[x];
  }
  __syncthreads();
  // x[d][a] = (sum_k Xi[d][k] H[k][a] - (sum_k Xi[d][k]) mean[a]) * scale
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int dl = 16 * x + acc_row(l4, r);
      const double rsum = s_rs[dl];
#pragma unroll
      for (int y = 0; y < 2; ++y) {
        const int a = c0 + 32 * w + 16 * y + l15;
        const double v = (a < p) ? (acc[x][y][r] - rsum * mean[a]) * scale : 0.0;
        draws[(int64_t)(d0 + dl) * ldd + a] = v;
      }
    }
}

// Xi [1024][n_pad] for sample ids first_id + k * stride, k < count; columns k >= count are zero
__device__ __forceinline__ void xi_fill_body(uint64_t seed, int64_t first_id, int64_t stride, int count, int n_pad,
                                             double* __restrict__ Xi) {
  const int k = blockIdx.x * 256 + threadIdx.x;   // column (sample)
  const int j = blockIdx.y;                       // Philox call of the sample
  if (k >= n_pad) return;
  double z0 = 0.0, z1 = 0.0;
  if (k < count) xi_pair(seed, (uint64_t)(first_id + (int64_t)k * stride), (uint32_t)j, z0, z1);
  const int d = 64 * (j >> 5) + (j & 31);
  Xi[(int64_t)d * n_pad + k] = z0;
  Xi[(int64_t)(d + 32) * n_pad + k] = z1;
}

__global__ __launch_bounds__(256) void xi_fill_kernel(uint64_t seed, int64_t first_id, int64_t stride, int count,
                                                      int n_pad, double* __restrict__ Xi) {
  xi_fill_body(seed, first_id, stride, count, n_pad, Xi);
}

// the same for the chunks of a group (blockIdx.z), every chunk with its own [1024][n_pad] block of the workspace
__global__ __launch_bounds__(256) void xi_fill_group_kernel(uint64_t seed, int64_t stride, EstChunks ch,
                                                            double* __restrict__ Xi) {
  const int c = blockIdx.z;
  xi_fill_body(seed, ch.first_id[c], stride, ch.count[c], ch.n_pad[c], Xi + ch.xi_off[c]);
}

// One draw's entry from the running sums; the fused multiply-add spelled out, so that every kernel that forms it
// (one check or a group's) rounds it the same way whatever the compiler would have contracted.
__device__ __forceinline__ double draw_value(double D, double s, double mean, double scale) {
  return __builtin_fma(-s, mean, D) * scale;
}

// x[d][a] = (D[d][a] - s[d] mean[a]) * scale; padding columns zero
__global__ __launch_bounds__(256) void running_draws_kernel(const double* __restrict__ D, const double* __restrict__ s,
                                                            const double* __restrict__ mean, double scale, int p,
                                                            int ld, double* __restrict__ draws) {
  const int d = blockIdx.y;
  const int a = blockIdx.x * 256 + threadIdx.x;
  if (a >= ld) return;
  draws[(int64_t)d * ld + a] = (a < p) ? draw_value(D[(int64_t)d * ld + a], s[d], mean[a], scale) : 0.0;
}

// Where the quantile kernels take the draws from: the draws buffer (after lsspa_error_draws / _running_draws and, with
// several ranks, the all-reduce), or -- one rank, running form -- x = (D - s mean^T) * scale evaluated as it is read,
// so that a check is two launches and the 8 MB of x are never written.
struct DrawSrc {
  const double* draws;     // [1024][ld] or null
  const double* D;         // [1024][ld]
  const double* s;         // [1024]
  const double* mean;      // [p]
  double scale;
  int ld;
  __device__ __forceinline__ double at(int d, int a) const {
    return draws ? draws[(int64_t)d * ld + a] : draw_value(D[(int64_t)d * ld + a], s[d], mean[a], scale);
  }
};

// norms[d] = ||x[d][0..p)||_2, one wave per draw
__global__ __launch_bounds__(256) void row_norms_kernel(DrawSrc src, int p, double* __restrict__ norms) {
  const int lane = threadIdx.x & 63, d = blockIdx.x * 4 + (threadIdx.x >> 6);
  double v = 0.0;
  for (int a = lane; a < p; a += 64) {
    const double x = src.at(d, a);
    v = __builtin_fma(x, x, v);
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
  if (lane == 0) norms[d] = sqrt(v);
}

// numpy.quantile(v, q) with the default 'linear' method on ND values.  out = [feature errors (p), overall error] and,
// with pack_mean, behind them [running mean (p), n]: one copy then brings a whole check to the host.
__device__ __forceinline__ void quantile_body(const DrawSrc& src, const double* __restrict__ norms, int p, double q,
                                              double* __restrict__ out /*[p + 1] or [2 p + 2]*/,
                                              const double* __restrict__ pack_mean,
                                              const double* __restrict__ pack_n) {
  __shared__ double s[ND];
  const int a = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < ND; i += 512) s[i] = (a < p) ? fabs(src.at(i, a)) : norms[i];
  __syncthreads();
  // Bitonic sort, one compare-exchange per thread and stage (512 pairs).  A stage of distance j works inside blocks of
  // 2 j elements; a wave's 64 pairs cover elements [128 w, 128 w + 128), so every stage with j <= 64 stays inside the
  // wave's own elements -- LDS operations of one wave execute in order -- and needs no workgroup barrier: 6 of the 55
  // stages (j = 128, 256, 512) have one on either side, the other 49 none (20.5 -> measured in DESIGN.md).
  for (int lk = 1; (1 << lk) <= ND; ++lk)
    for (int lj = lk - 1; lj >= 0; --lj) {
      const int k = 1 << lk, j = 1 << lj;
      const bool wide = j >= 128;
      if (wide) __syncthreads();
      const int i = ((tid >> lj) << (lj + 1)) + (tid & (j - 1)), ixj = i + j;
      const double x = s[i], y = s[ixj];
      const bool up = (i & k) == 0;
      if ((x > y) == up) {
        s[i] = y;
        s[ixj] = x;
      }
      if (wide) __syncthreads();
      else __builtin_amdgcn_wave_barrier();
    }
  __syncthreads();
  if (tid == 0) {
    const double pos = q * (ND - 1);
    const int lo = (int)floor(pos);
    const int hi = lo + 1 < ND ? lo + 1 : ND - 1;
    const double t = pos - lo;
    const double va = s[lo], vb = s[hi];
    // numpy's _lerp: a + (b - a) t, evaluated from b's side when t >= 0.5
    out[a] = (t >= 0.5) ? vb - (vb - va) * (1.0 - t) : va + (vb - va) * t;
    if (pack_mean) {
      if (a < p) out[p + 1 + a] = pack_mean[a];
      else out[2 * p + 1] = pack_n[0];
    }
  }
}

__global__ __launch_bounds__(512) void quantile_kernel(DrawSrc src, const double* __restrict__ norms, int p, double q,
                                                       double* __restrict__ out, const double* __restrict__ pack_mean,
                                                       const double* __restrict__ pack_n) {
  quantile_body(src, norms, p, q, out, pack_mean, pack_n);
}

// The checks of a group's chunks in one launch (blockIdx.y = the check): each reads the estimator's sums, the running
// mean and n as they stood after ITS chunk (the snapshots the group kernels keep) and writes into its own result slot.
__global__ __launch_bounds__(512) void quantile_group_kernel(EstChecks ck, const double* __restrict__ Dsnap,
                                                             const double* __restrict__ ssnap,
                                                             const double* __restrict__ mean_snap,
                                                             const double* __restrict__ n_snap,
                                                             const double* __restrict__ norms, int p, int ld, double q,
                                                             double* __restrict__ res) {
  const int c = ck.chunk[blockIdx.y];
  const DrawSrc src{nullptr, Dsnap + (int64_t)c * ND * ld, ssnap + (int64_t)c * ND, mean_snap + (int64_t)c * p,
                    ck.scale[blockIdx.y], ld};
  quantile_body(src, norms + (int64_t)c * ND, p, q, res + (int64_t)ck.slot[blockIdx.y] * (2 * p + 2), src.mean,
                n_snap + c);
}

// D[1024][ld] += Xi L and s += Xi 1 for ONE chunk of samples (tens to a few hundred): the 64 x 128-tile kernel above
// has 16 workgroups per feature tile and walks the samples 16 at a time between two barriers -- 22 us at p = 100, 128
// samples, all of it latency.  Here a workgroup owns 16 draws x 128 features (64 workgroups per feature tile), its four
// waves split the samples among them, every lane fetches its own matrix-instruction operands straight from memory (no
// staging, no barrier in the loop), and the four partial tiles meet once in LDS.
// PARTS: the chunk's own product and row sums are stored (a group's chunks side by side, prefix_norms_kernel adds them
// up in order) instead of added to D and s -- the same numbers either way: D + v here, D + P there.
template <bool PARTS>
__device__ __forceinline__ void acc_small_body(const double* __restrict__ Xi, int n_pad, const double* __restrict__ L,
                                               int ldl, int k_valid, int c_valid, int p, double* __restrict__ D, int ld,
                                               double* __restrict__ rowsum_acc) {
  __shared__ double part[4 * 8 * 4 * 64];   // [wave][feature tile][r][lane]
  __shared__ double s_rs[4][16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int d0 = blockIdx.x * 16, c0 = blockIdx.y * 128;
  d4 acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) acc[t] = d4_zero();
  double rs = 0.0;
  const int kq = n_pad / 4;                  // this wave's samples: [w kq, (w + 1) kq), kq a multiple of 4
  const double* arow = Xi + (int64_t)(d0 + l15) * n_pad + w * kq + l4;
  // four k-steps a turn, all 36 loads of a turn issued before its first product (the loop is a chain of memory round
  // trips otherwise: 10 us at 128 samples with two steps in flight)
  for (int st0 = 0; st0 < kq / 4; st0 += 4) {
    double a[4], b[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int st = st0 + u;
      const bool live = st < kq / 4;
      const int k = w * kq + 4 * st + l4;
      a[u] = live ? arow[4 * st] : 0.0;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int c = c0 + 16 * t + l15;
        b[u][t] = (live && k < k_valid && c < c_valid) ? L[(int64_t)k * ldl + c] : 0.0;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      rs += a[u];
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] = mfma(a[u], b[u][t], acc[t]);
    }
  }
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) part[((w * 8 + t) * 4 + r) * 64 + lane] = acc[t][r];
  rs += __shfl_xor(rs, 16, 64);
  rs += __shfl_xor(rs, 32, 64);
  if (l4 == 0) s_rs[w][l15] = rs;
  __syncthreads();
  // wave w adds up feature tiles 2 w and 2 w + 1 and folds them into D
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int t = 2 * w + u;
    const int c = c0 + 16 * t + l15;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double v = 0.0;
#pragma unroll
      for (int ww = 0; ww < 4; ++ww) v += part[((ww * 8 + t) * 4 + r) * 64 + lane];
      if (c < p) {
        double* dst = D + (int64_t)(d0 + acc_row(l4, r)) * ld + c;
        *dst = PARTS ? v : *dst + v;
      }
    }
  }
  if (blockIdx.y == 0 && tid < 16) {
    const double rsum = (s_rs[0][tid] + s_rs[1][tid]) + (s_rs[2][tid] + s_rs[3][tid]);
    rowsum_acc[d0 + tid] = PARTS ? rsum : rowsum_acc[d0 + tid] + rsum;
  }
}

__global__ __launch_bounds__(256) void acc_small_kernel(const double* __restrict__ Xi, int n_pad,
                                                        const double* __restrict__ L, int ldl, int k_valid, int c_valid,
                                                        int p, double* __restrict__ D, int ld,
                                                        double* __restrict__ rowsum_acc) {
  acc_small_body<false>(Xi, n_pad, L, ldl, k_valid, c_valid, p, D, ld, rowsum_acc);
}

// the products of a group's chunks side by side (blockIdx.z = chunk): P[c] = Xi_c L_c [1024][ld], S[c] = Xi_c 1
__global__ __launch_bounds__(256) void acc_group_kernel(EstChunks ch, const double* __restrict__ Xi,
                                                        const double* __restrict__ lifts, int p, int ld,
                                                        double* __restrict__ P, double* __restrict__ S) {
  const int c = blockIdx.z;
  acc_small_body<true>(Xi + ch.xi_off[c], ch.n_pad[c], lifts + (int64_t)ch.first[c] * p, p, ch.count[c], p, p,
                       P + (int64_t)c * ND * ld, ld, S + (int64_t)c * ND);
}

// D and s advanced chunk by chunk from the group's products, in order (D_c = D_{c-1} + P_c, what acc_small_kernel does
// chunk after chunk), every state kept for the check that belongs to it, and the draws' row norms of every state that
// has a check (scale != 0) as row_norms_kernel forms them: one wave per draw, lanes over the features (p <= 128).
__global__ __launch_bounds__(256) void prefix_norms_kernel(EstChunks ch, const double* __restrict__ P,
                                                           const double* __restrict__ S, double* __restrict__ D,
                                                           double* __restrict__ s, double* __restrict__ Dsnap,
                                                           double* __restrict__ ssnap,
                                                           const double* __restrict__ mean_snap, int p, int ld,
                                                           double* __restrict__ norms) {
  const int lane = threadIdx.x & 63, d = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int a0 = lane, a1 = lane + 64;
  double r0 = a0 < p ? D[(int64_t)d * ld + a0] : 0.0, r1 = a1 < p ? D[(int64_t)d * ld + a1] : 0.0;
  double sr = s[d];
  // eight chunks a turn, all their loads issued before the first sum (a chain of dependent memory round trips otherwise:
  // 30 us for sixteen chunks)
  constexpr int CB = 8;
  for (int c0 = 0; c0 < ch.n; c0 += CB) {
    double p0[CB], p1[CB], sc[CB], m0[CB], m1[CB], scl[CB];
#pragma unroll
    for (int u = 0; u < CB; ++u) {
      const int c = c0 + u;
      const bool live = c < ch.n;
      const int64_t o = ((int64_t)(live ? c : c0) * ND + d) * ld;
      p0[u] = (live && a0 < p) ? P[o + a0] : 0.0;
      p1[u] = (live && a1 < p) ? P[o + a1] : 0.0;
      sc[u] = live ? S[(int64_t)c * ND + d] : 0.0;
      scl[u] = live ? ch.scale[c] : 0.0;
      const double* mean = mean_snap + (int64_t)(live ? c : c0) * p;
      m0[u] = (live && a0 < p) ? mean[a0] : 0.0;
      m1[u] = (live && a1 < p) ? mean[a1] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < CB; ++u) {
      const int c = c0 + u;
      if (c < ch.n) {
        const int64_t o = ((int64_t)c * ND + d) * ld;
        if (a0 < p) {
          r0 = r0 + p0[u];
          Dsnap[o + a0] = r0;
        }
        if (a1 < p) {
          r1 = r1 + p1[u];
          Dsnap[o + a1] = r1;
        }
        sr = sr + sc[u];
        if (lane == 0) ssnap[(int64_t)c * ND + d] = sr;
        const double scale = scl[u];
        if (scale != 0.0) {
          double v = 0.0;
          if (a0 < p) {
            const double x = draw_value(r0, sr, m0[u], scale);
            v = __builtin_fma(x, x, v);
          }
          if (a1 < p) {
            const double x = draw_value(r1, sr, m1[u], scale);
            v = __builtin_fma(x, x, v);
          }
#pragma unroll
          for (int o2 = 1; o2 < 64; o2 <<= 1) v += __shfl_xor(v, o2, 64);
          if (lane == 0) norms[(int64_t)c * ND + d] = sqrt(v);
        }
      }
    }
  }
  if (a0 < p) D[(int64_t)d * ld + a0] = r0;
  if (a1 < p) D[(int64_t)d * ld + a1] = r1;
  if (lane == 0) s[d] = sr;
}

hipError_t launch_error_draws(const double* Xi, int ldxi, const double* H, int ldh, int n_pad,
                              const double* mean, double scale, int p, double* draws, int ldd, hipStream_t st) {
  const int n_tiles = (p + 127) / 128;
  if (p < 1 || n_pad < KCH || n_pad % KCH != 0 || ldxi < n_pad || (ldxi & 1) || ldh % 128 != 0 ||
      ldh < n_tiles * 128 || ldd < n_tiles * 128)
    return hipErrorInvalidValue;
  hipLaunchKernelGGL(draws_kernel, dim3(ND / 64, n_tiles), dim3(256), 0, st, Xi, ldxi, H, ldh, n_pad, mean, scale, p,
                     draws, ldd);
  return hipGetLastError();
}

hipError_t launch_error_xi(uint64_t seed, int64_t first_id, int64_t stride, int count, int n_pad, double* Xi,
                           hipStream_t st) {
  if (count < 0 || n_pad < KCH || n_pad % KCH != 0 || count > n_pad || stride < 1 || first_id < 0)
    return hipErrorInvalidValue;
  hipLaunchKernelGGL(xi_fill_kernel, dim3((n_pad + 255) / 256, ND / 2), dim3(256), 0, st, seed, first_id, stride,
                     count, n_pad, Xi);
  return hipGetLastError();
}

hipError_t launch_error_accumulate(uint64_t seed, int64_t first_id, int64_t stride, int count, int n_pad, double* Xi,
                                   const double* L, int ldl, int raw, int ldh, int p, double* D, double* s,
                                   hipStream_t st) {
  const int n_tiles = (p + 127) / 128;
  if (p < 1 || count < 1 || n_pad < KCH || n_pad % KCH != 0 || count > n_pad || stride < 1 || first_id < 0 ||
      ldh % 128 != 0 || ldh < n_tiles * 128 || (raw ? ldl < p : ldl != ldh) || !Xi)
    return hipErrorInvalidValue;
  // the normals by a launch of their own, one thread per Philox call (count x 512 of them: a thousand waves), not inside
  // the GEMM: its 64 waves would make them one after the other -- fp64 log and sincospi are ~300 instructions a call,
  // 28 us against 3 + 5 (measured at p = 100, 128 samples)
  hipLaunchKernelGGL(xi_fill_kernel, dim3((n_pad + 255) / 256, ND / 2), dim3(256), 0, st, seed, first_id, stride, count,
                     n_pad, Xi);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  // the lift vectors as the lift kernels left them ([count][ldl], no padding: guarded reads) or the padded staging
  hipLaunchKernelGGL(acc_small_kernel, dim3(ND / 16, n_tiles), dim3(256), 0, st, (const double*)Xi, n_pad, L, ldl,
                     raw ? count : n_pad, raw ? p : ldh, p, D, ldh, s);
  return hipGetLastError();
}

hipError_t launch_error_running_draws(const double* D, const double* s, const double* mean, double scale, int p,
                                      int ld, double* draws, hipStream_t st) {
  if (p < 1 || ld < p) return hipErrorInvalidValue;
  hipLaunchKernelGGL(running_draws_kernel, dim3((ld + 255) / 256, ND), dim3(256), 0, st, D, s, mean, scale, p, ld,
                     draws);
  return hipGetLastError();
}

static hipError_t launch_quantiles(const DrawSrc& src, int p, double* norms, double* out, const double* pack_mean,
                                   const double* pack_n, hipStream_t st) {
  // (the overall error's workgroup summing its 1024 draws' squares itself, to save the launch: 105 us instead of 4.7 + 21
  // at p = 100 -- one workgroup's dependent shuffle chains against 256 workgroups' -- measured, not kept)
  hipLaunchKernelGGL(row_norms_kernel, dim3(ND / 4), dim3(256), 0, st, src, p, norms);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(quantile_kernel, dim3(p + 1), dim3(512), 0, st, src, (const double*)norms, p, 0.95, out, pack_mean,
                     pack_n);
  return hipGetLastError();
}

hipError_t launch_error_quantiles(const double* draws, int ldd, int p, double* norms, double* out, hipStream_t st,
                                  const double* pack_mean, const double* pack_n) {
  if (p < 1 || ldd < p) return hipErrorInvalidValue;
  return launch_quantiles(DrawSrc{draws, nullptr, nullptr, nullptr, 0.0, ldd}, p, norms, out, pack_mean, pack_n, st);
}

hipError_t launch_error_quantiles_running(const double* D, const double* s, const double* mean, double scale, int ld,
                                          int p, double* norms, double* out, const double* pack_n, hipStream_t st) {
  if (p < 1 || ld < p) return hipErrorInvalidValue;
  return launch_quantiles(DrawSrc{nullptr, D, s, mean, scale, ld}, p, norms, out, mean, pack_n, st);
}

// A group's chunks into the running estimator and their checks, five launches whatever the number of chunks (p <= 128):
// the normals of every chunk, the chunks' products side by side, the ordered sums with a snapshot per chunk and the
// row norms, then all the checks' quantiles.  Xi: the workspace (sum over the chunks of 1024 x n_pad doubles); P, Dsnap:
// [n][1024][ld]; S, ssnap, norms: [n][1024]; mean_snap [n][p] / n_snap [n]: the statistics after every chunk
// (launch_stats_small_multi); res: the result slots ([2 p + 2] each).
hipError_t launch_error_group(uint64_t seed, int64_t stride, const EstChunks& ch, const EstChecks& ck, double* Xi,
                              const double* lifts, int p, int ld, double* P, double* S, double* D, double* s,
                              double* Dsnap, double* ssnap, const double* mean_snap, const double* n_snap,
                              double* norms, double* res, hipStream_t st) {
  if (p < 1 || p > 128 || ld != 128 || ch.n < 1 || ch.n > EstChunks::MAX || ck.n < 0 || ck.n > ch.n || stride < 1)
    return hipErrorInvalidValue;
  int pad_max = 0;
  for (int c = 0; c < ch.n; ++c) {
    if (ch.count[c] < 1 || ch.n_pad[c] < KCH || ch.n_pad[c] % KCH != 0 || ch.count[c] > ch.n_pad[c] ||
        ch.first_id[c] < 0 || ch.first[c] < 0 || ch.xi_off[c] < 0)
      return hipErrorInvalidValue;
    pad_max = ch.n_pad[c] > pad_max ? ch.n_pad[c] : pad_max;
  }
  for (int k = 0; k < ck.n; ++k)
    if (ck.chunk[k] < 0 || ck.chunk[k] >= ch.n || ck.slot[k] < 0 || ch.scale[ck.chunk[k]] != ck.scale[k])
      return hipErrorInvalidValue;
  hipLaunchKernelGGL(xi_fill_group_kernel, dim3((pad_max + 255) / 256, ND / 2, ch.n), dim3(256), 0, st, seed, stride, ch,
                     Xi);
  hipLaunchKernelGGL(acc_group_kernel, dim3(ND / 16, 1, ch.n), dim3(256), 0, st, ch, (const double*)Xi, lifts, p, ld, P,
                     S);
  hipLaunchKernelGGL(prefix_norms_kernel, dim3(ND / 4), dim3(256), 0, st, ch, (const double*)P, (const double*)S, D, s,
                     Dsnap, ssnap, mean_snap, p, ld, norms);
  if (ck.n > 0)
    hipLaunchKernelGGL(quantile_group_kernel, dim3(p + 1, ck.n), dim3(512), 0, st, ck, (const double*)Dsnap,
                       (const double*)ssnap, mean_snap, n_snap, (const double*)norms, p, ld, 0.95, res);
  return hipGetLastError();
}

}  // namespace lsspa
