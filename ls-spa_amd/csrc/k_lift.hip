// Lift extraction, running statistics and the final back-substitution.
//
// Reference counterparts (cvxgrp/ls-spa, ls_spa/ls_spa.py):
//   lift_partial / lift_finish -> costs, R_sq, ediff1d, argsort scatter   (:282-285)
//                                 and the antithetical average            (:205-208)
//   stats_batch / stats_merge  -> merge_sample_mean / merge_sample_cov    (:103-119, :212-216)
//   backsolve                  -> theta = lstsq(R_tr, y_tr)               (:240)
//
// With V = L^-1 RHS (rows j in ordering position, columns c in "test space"),
// z = L^-1 g_pi and y~ the reduced test target, the fitted values of the j-th nested
// model are  N_j = sum_{k<=j} z_k V[k,:]  and its lift is
//     ( |y~ - N_{j-1}|^2 - |y~ - N_j|^2 ) / |y_test|^2 = z_j V[j,:] . (2 y~ - N_j - N_{j-1}) / |y_test|^2
#include "kernels.h"
#include "tiles.h"

namespace lsspa {

// One wave per 64-column strip: lane = column, rows walked in order (the running N is a scan
// down the rows).  Rows are taken 16 at a time: 16 independent coalesced loads, the scan in
// registers, then the 16 per-row dot products are reduced through a small LDS tile.
// VT: the kernel is handed V^T (chunk-major, from the panel launches' X tiles) instead of V: column c of V is row c of
// V^T, whose sixteen entries j0 .. j0 + 15 are one contiguous 128-byte (fp32: 64-byte) piece of chunk j0 / 16; the
// pieces of a wave's 64 columns follow each other, 8 KB in one run.  The wave fetches the run with coalesced 16-byte
// loads and turns it round in LDS (the tile that afterwards carries the dot-product terms) -- a lane fetching its own
// piece with eight 16-byte loads 128 bytes apart touched 64 lines per instruction: 0.49 against 0.23 ms per C3 step.
template <typename T, bool VT>
__global__ __launch_bounds__(256) void lift_partial_kernel(LiftArgs a) {
  constexpr int XT_LD = 17;    // V^T staging tile: [64 columns of V][16 rows j], conflict-free both ways
  __shared__ double s_E[4][VT ? 64 * XT_LD : 16 * TT_LD];
  static_assert(64 * XT_LD >= 16 * TT_LD, "the staging tile also serves as the reduction tile");
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int ord = blockIdx.x;
  const int nstrips = a.m_pad / 64;
  // Strip of wave w.  In tri mode a strip only has the rows below its first column, so its work falls off
  // linearly with its index: deal the strips so that every workgroup gets the same total -- wave 0 walks
  // up from the left, wave 1 down from the middle, wave 2 up from the middle, wave 3 down from the right.
  int strip = blockIdx.y * 4 + w;
  if (a.tri && (nstrips % 4) == 0) {
    const int nwg = nstrips / 4, y = blockIdx.y;
    strip = (w == 0) ? y : (w == 1) ? 2 * nwg - 1 - y : (w == 2) ? 2 * nwg + y : 4 * nwg - 1 - y;
  }
  if (strip >= nstrips) return;  // whole wave leaves; no workgroup barrier below
  const int cs = strip * 64;
  const int p = a.p, p_pad = a.p_pad, m_pad = a.m_pad;
  const int n_iblk = (p + NB - 1) / NB;
  const int64_t ldv = ldv_of(m_pad);
  const T* L = static_cast<const T*>(a.A) + (int64_t)ord * p_pad * p_pad;   // chunk-major: row p is L[cm_off(p_pad, p, j)]
  const T* V = static_cast<const T*>(a.V) + (VT ? (int64_t)ord * p_pad * p_pad : (int64_t)ord * v_rows_of(p) * ldv);
  double* Pp = a.Ppart + ((int64_t)ord * nstrips + strip) * p_pad;
  const int c = cs + lane;
  double yt;
  if (a.tri)
    yt = (c < p) ? (double)static_cast<const T*>(a.At)[(int64_t)ord * p_pad * p_pad + cm_off(p_pad, p, c)] : 0.0;
  else
    yt = a.ytil[c];
  double* E = s_E[w];
  const int r16 = lane & 15, q4 = lane >> 4;

  double run = 0.0;
  const int n_rows = n_iblk * 64;
  for (int j0 = 0; j0 < n_rows; j0 += 16) {
    if (a.tri && j0 + 16 <= cs) {  // V is lower triangular: nothing in this strip yet
      if (lane < 16) Pp[j0 + lane] = 0.0;
      continue;
    }
    // unconditional loads (rows up to n_rows exist and are written by the strip kernel), values selected
    // afterwards: a guarded load becomes a branch, and sixteen of them a latency chain
    T vraw[16];
    if constexpr (VT) {
      typedef typename Tr<T>::vec_t vec_t;
      constexpr int VE = Tr<T>::VE, NQ = 16 / VE;       // vectors per piece = loads per lane
      const vec_t* src = reinterpret_cast<const vec_t*>(V + ((int64_t)(j0 >> 4) * p_pad + cs) * 16);
      vec_t t[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) t[q] = __builtin_nontemporal_load(src + 64 * q + lane);
#pragma unroll
      for (int q = 0; q < NQ; ++q) {       // vector 64 q + lane = piece (column) (64 q + lane) / NQ, entries VE * ((64 q + lane) % NQ) ..
        const int vi = 64 * q + lane, col = vi / NQ, e0 = VE * (vi % NQ);
#pragma unroll
        for (int e = 0; e < VE; ++e) E[col * XT_LD + e0 + e] = (double)t[q][e];
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) vraw[jj] = (T)E[lane * XT_LD + jj];
      __builtin_amdgcn_wave_barrier();    // E is rewritten with the terms below
    } else {
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) vraw[jj] = __builtin_nontemporal_load(V + (j0 + jj) * ldv + c);
    }
    const T zraw = L[cm_off(p_pad, p, min(j0 + r16, p - 1))];
    double v[16];
    // VT: rows c >= p of V^T belong to no feature (and the last ones are not even computed)
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) v[jj] = (j0 + jj < p && (!VT || c < p)) ? (double)vraw[jj] : 0.0;
    const double zl = (j0 + r16 < p) ? (double)zraw : 0.0;  // lane r16 holds z[j0 + r16]
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      const double zj = __shfl(zl, jj, 64);
      const double t = zj * v[jj];
      E[jj * TT_LD + lane] = v[jj] * (2.0 * (yt - run) - t);
      run += t;
    }
    __builtin_amdgcn_wave_barrier();
    // lane (r16, q4) sums columns 16 q4 .. 16 q4 + 15 of row r16, then the four quarters meet
    double s = 0.0;
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) s += E[r16 * TT_LD + 16 * q4 + cc];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (lane < 16) Pp[j0 + lane] = s;
    __builtin_amdgcn_wave_barrier();
  }
}

// lifts[sample][perm[j]] = mean over the sample's orderings of z_j * sum_strips P_j / |y|^2
template <typename T>
__global__ __launch_bounds__(256) void lift_finish_kernel(LiftArgs a) {
  const int sample = blockIdx.x;
  const int nstrips = a.m_pad / 64;
  const int p = a.p, p_pad = a.p_pad;
  const double wgt = 1.0 / (a.per_sample * a.y_norm_sq);
  double* out = a.lifts + (int64_t)sample * p;
  for (int k = 0; k < a.per_sample; ++k) {
    const int ord = sample * a.per_sample + k;
    const int32_t* perm = a.perms + (int64_t)ord * p;
    const T* Lm = static_cast<const T*>(a.A) + (int64_t)ord * p_pad * p_pad;
    const double* Pp = a.Ppart + (int64_t)ord * nstrips * p_pad;
    for (int j = threadIdx.x; j < p; j += 256) {
      double s = 0.0;
      const int nt = a.fused ? j / 128 + 1 : nstrips;
      for (int t = 0; t < nt; ++t) s += Pp[(int64_t)t * p_pad + j];
      const double val = (double)Lm[cm_off(p_pad, p, j)] * s * wgt;
      const int f = perm[j];
      if (k == 0)
        out[f] = val;
      else
        out[f] += val;
    }
    __syncthreads();  // a feature is written by different threads in different orderings
  }
}

// The same for antithetical pairs laid out back to back (ordering 2 s + 1 = ordering 2 s reversed): position j
// of the first ordering and position p - 1 - j of the second are the same feature, so one thread adds both
// contributions and no ordering of writes is needed -- the work spreads over (sample, 256 positions) workgroups.
template <typename T>
__global__ __launch_bounds__(256) void lift_finish_paired_kernel(LiftArgs a) {
  const int sample = blockIdx.x;
  const int j = blockIdx.y * 256 + threadIdx.x;
  const int nstrips = a.m_pad / 64;      // rows of Ppart per ordering (fused: only the first j / 128 + 1 are written)
  const int p = a.p, p_pad = a.p_pad;
  if (j >= p) return;
  const double wgt = 0.5 / a.y_norm_sq;
  const int ord = 2 * sample;
  const int j2 = p - 1 - j;
  const T* L0 = static_cast<const T*>(a.A) + (int64_t)ord * p_pad * p_pad;
  const T* L1 = L0 + (int64_t)p_pad * p_pad;
  const double* P0 = a.Ppart + (int64_t)ord * nstrips * p_pad;
  const double* P1 = P0 + (int64_t)nstrips * p_pad;
  double s0 = 0.0, s1 = 0.0;
  if (a.fused) {      // row block I' of V^T has entries in the columns j >= 128 I' only
    for (int t = 0; t <= j / 128; ++t) s0 += P0[(int64_t)t * p_pad + j];
    for (int t = 0; t <= j2 / 128; ++t) s1 += P1[(int64_t)t * p_pad + j2];
  } else {
    for (int t = 0; t < nstrips; ++t) {
      s0 += P0[(int64_t)t * p_pad + j];
      s1 += P1[(int64_t)t * p_pad + j2];
    }
  }
  const double v0 = (double)L0[cm_off(p_pad, p, j)] * s0 * wgt;
  const double v1 = (double)L1[cm_off(p_pad, p, j2)] * s1 * wgt;
  a.lifts[(int64_t)sample * p + a.perms[(int64_t)ord * p + j]] = v0 + v1;
}

// Every ordering's lifts telescope to the R^2 of the full model (ls_spa/ls_spa.py:284-285: R_sq[p] - R_sq[0]), hence so
// does every sample's antithetical mean: a free end-to-end check of a batch -- a hand-over inside a panel launch that
// delivered stale data, a tile skipped, an ordering read wrong all break it.  One wave per sample; a sum off by more
// than tol (or not a number) raises LSSPA_INFO_SUM, and the largest deviation seen is kept next to the info word.
__global__ __launch_bounds__(256) void sum_check_kernel(const double* __restrict__ lifts, int n_samples, int p, double r2,
                                                        double tol, int32_t* __restrict__ info) {
  const int lane = threadIdx.x & 63, s = blockIdx.x * 4 + (threadIdx.x >> 6);
  double v = r2;                 // (a wave beyond the last sample contributes no deviation)
  if (s < n_samples) {
    v = 0.0;
    for (int a = lane; a < p; a += 64) v += lifts[(int64_t)s * p + a];
  } else {
    v = (lane == 0) ? r2 : 0.0;
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
  __shared__ double s_dev[4];
  double dev = fabs(v - r2);
  if (!(dev == dev)) dev = __longlong_as_double(0x7ff0000000000000ll);   // not a number counts as infinitely far off
  if (lane == 0) s_dev[threadIdx.x >> 6] = dev;
  __syncthreads();
  if (threadIdx.x == 0) {     // one atomic per workgroup, and none while the deviation is exactly zero
    const int nw = min(4, n_samples - blockIdx.x * 4);
    double worst = 0.0;
    for (int k = 0; k < nw; ++k) worst = fmax(worst, s_dev[k]);
    if (!(worst <= tol)) atomicOr(info, 8);                  // LSSPA_INFO_SUM
    // non-negative doubles order like their bit patterns
    if (worst > 0.0)
      atomicMax(reinterpret_cast<unsigned long long*>(info + 2), (unsigned long long)__double_as_longlong(worst));
  }
}

hipError_t launch_sum_check(const double* lifts, int n_samples, int p, double r2, double tol, int32_t* info,
                            hipStream_t st) {
  if (n_samples < 1 || p < 1 || !(tol >= 0.0)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(sum_check_kernel, dim3((n_samples + 3) / 4), dim3(256), 0, st, lifts, n_samples, p, r2, tol, info);
  return hipGetLastError();
}

hipError_t launch_lift(const LiftArgs& a, hipStream_t st) {
  if (a.p < 1 || a.p_pad % NB != 0 || a.m_pad % 128 != 0 || a.n_ord < 1 ||
      (a.per_sample != 1 && a.per_sample != 2) || a.n_ord % a.per_sample != 0 || !(a.y_norm_sq > 0.0))
    return hipErrorInvalidValue;
  const dim3 g1(a.n_ord, (a.m_pad / 64 + 3) / 4), g2(a.n_ord / a.per_sample);
  if (a.vt && a.fused) {      // the X tiles of the panel launches have scanned V^T themselves: only the finish is left
    if (!a.tri || a.m_pad > a.p_pad || a.m_pad / 64 < a.p_pad / 128) return hipErrorInvalidValue;
  } else if (a.vt) {
    if (!a.tri || a.m_pad > a.p_pad) return hipErrorInvalidValue;
    if (a.f32)
      hipLaunchKernelGGL((lift_partial_kernel<float, true>), g1, dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL((lift_partial_kernel<double, true>), g1, dim3(256), 0, st, a);
  } else if (a.f32)
    hipLaunchKernelGGL((lift_partial_kernel<float, false>), g1, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((lift_partial_kernel<double, false>), g1, dim3(256), 0, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (a.per_sample == 2 && a.paired) {
    const dim3 g3(a.n_ord / 2, (a.p + 255) / 256);
    if (a.f32)
      hipLaunchKernelGGL(lift_finish_paired_kernel<float>, g3, dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL(lift_finish_paired_kernel<double>, g3, dim3(256), 0, st, a);
  } else if (a.f32)
    hipLaunchKernelGGL(lift_finish_kernel<float>, g2, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(lift_finish_kernel<double>, g2, dim3(256), 0, st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// pending-batch moments about the running mean mu:  D = lifts - mu,
//   buf[0] (+)= n_s ;  buf[1+a] (+)= sum_s D[s][a] ;  buf[1+p+a*p+b] (+)= sum_s D[s][a] D[s][b]
// 64 x 64 output tile per workgroup, 16 samples staged per step.
// ---------------------------------------------------------------------------------------
// With few 64 x 64 tiles (small p) the samples are cut into gridDim.z slices; slice z writes its moments to
// parts + z * (1 + p + p*p) and stats_reduce_kernel adds the slices in a fixed order (no float atomics).
__global__ __launch_bounds__(256) void stats_batch_kernel(const double* __restrict__ lifts,
                                                          const double* __restrict__ mean,
                                                          double* __restrict__ buf, int n_samples, int p,
                                                          int accumulate, int per_slice) {
  __shared__ double sa[16][65];
  __shared__ double sb[16][65];
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  const int a0 = blockIdx.y * 64, b0 = blockIdx.x * 64;
  double acc[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[u][v] = 0.0;
  double colsum[4] = {0.0, 0.0, 0.0, 0.0};
  const bool do_sum = (blockIdx.x == 0);  // column sums once per a-tile
  const int s_lo = blockIdx.z * per_slice;
  const int s_hi = min(n_samples, s_lo + per_slice);
  if (gridDim.z > 1) {
    buf += (int64_t)blockIdx.z * ((int64_t)1 + p + (int64_t)p * p);
    accumulate = 0;
  }
  lifts += (int64_t)s_lo * p;
  n_samples = max(0, s_hi - s_lo);

  for (int s0 = 0; s0 < n_samples; s0 += 16) {
    __syncthreads();
    for (int idx = tid; idx < 16 * 64; idx += 256) {
      const int s = idx >> 6, c = idx & 63;
      double va = 0.0, vb = 0.0;
      if (s0 + s < n_samples) {
        if (a0 + c < p) va = lifts[(int64_t)(s0 + s) * p + a0 + c] - mean[a0 + c];
        if (b0 + c < p) vb = lifts[(int64_t)(s0 + s) * p + b0 + c] - mean[b0 + c];
      }
      sa[s][c] = va;
      sb[s][c] = vb;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      double av[4], bv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) av[u] = sa[s][ty + 16 * u];
#pragma unroll
      for (int v = 0; v < 4; ++v) bv[v] = sb[s][tx + 16 * v];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] += av[u] * bv[v];
      if (do_sum && tx == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) colsum[u] += av[u];
      }
    }
  }
  double* S = buf + 1;
  double* Q = buf + 1 + p;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int ai = a0 + ty + 16 * u;
    if (ai >= p) continue;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int bi = b0 + tx + 16 * v;
      if (bi >= p) continue;
      const int64_t o = (int64_t)ai * p + bi;
      Q[o] = accumulate ? Q[o] + acc[u][v] : acc[u][v];
    }
    if (do_sum && tx == 0) S[ai] = accumulate ? S[ai] + colsum[u] : colsum[u];
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0)
    buf[0] = accumulate ? buf[0] + (double)n_samples : (double)n_samples;
}

__global__ __launch_bounds__(256) void stats_reduce_kernel(const double* __restrict__ parts, int n_parts,
                                                           int64_t len, double* __restrict__ buf, int accumulate) {
  for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < len; o += (int64_t)gridDim.x * 256) {
    double s = 0.0;
    for (int z = 0; z < n_parts; ++z) s += parts[(int64_t)z * len + o];
    buf[o] = accumulate ? buf[o] + s : s;
  }
}

// Batch moments for small p on the matrix pipe: one wave per lower 16 x 16 tile of Q = D^T D (D = lifts - mean,
// n_samples x p), operands straight from memory (16 consecutive features of one sample = one 128-byte segment),
// the tile and its mirror stored together (Q stays exactly symmetric).  The last workgroup sums the columns (S) and
// writes n_b.  A few microseconds where the 64 x 64-tile kernel above, with a handful of workgroups, takes forty.
// Moments of one 16 x 16 tile by a 256-thread workgroup: four waves, a quarter of the samples each (one wave per tile
// was a chain of 2-3 x 32 matrix instructions, each of which holds its wave for 64 cycles, with no other wave on the
// SIMD to fill them).  The column sums are plain vector adds over a lane's samples plus two
// cross-lane steps; the four partial tiles and column sums meet in LDS and are added in fixed order (bitwise
// reproducible).  Thread tid gets element (row tid / 16, column tid % 16) of the tile: q = sum_s D[s][a] D[s][b],
// sa = sum_s D[s][a], sb = sum_s D[s][b] with D = lifts - mean.  Both small-p statistics kernels go through here, so
// the one-GPU fused merge and the all-reduce path add the same numbers in the same order.
struct TileMoments {
  double q, sa, sb;
};
// (lane_sa / lane_sb, optional: the tile's column sums once more, indexed by the lane's own column l15 -- what a
// workgroup needs to advance the means it holds per lane, stats_small_multi_kernel)
__device__ __forceinline__ TileMoments small_tile_moments_mu(const double* __restrict__ lifts, const double mua,
                                                             const double mub, int n_samples, int p, int ti, int tj,
                                                             double* lane_sa = nullptr, double* lane_sb = nullptr) {
  __shared__ double s_acc[4][256];
  __shared__ double s_sa[4][16], s_sb[4][16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int a = 16 * ti + l15, b = 16 * tj + l15;
  const int ac = a < p ? a : p - 1, bc = b < p ? b : p - 1;     // clamped addresses, value selected afterwards
  // samples of this wave: a contiguous quarter, a multiple of four long
  const int per = ((n_samples + 15) / 16) * 4;
  const int s_lo = wv * per, s_hi = min(n_samples, s_lo + per);
  d4 acc = d4_zero();
  double ca = 0.0, cb = 0.0;
  constexpr int KT = 8;
  for (int s0 = s_lo; s0 < s_hi; s0 += 4 * KT) {
    double ra[KT], rb[KT];
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) {
      const int s = s0 + 4 * kk + l4;
      const int sc = s < s_hi ? s : n_samples - 1;
      ra[kk] = lifts[(int64_t)sc * p + ac];
      rb[kk] = lifts[(int64_t)sc * p + bc];
    }
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) {
      const int s = s0 + 4 * kk + l4;
      const double av = (s < s_hi && a < p) ? ra[kk] - mua : 0.0;
      const double bv = (s < s_hi && b < p) ? rb[kk] - mub : 0.0;
      acc = mfma(av, bv, acc);
      ca += av;
      cb += bv;
    }
  }
  // column sums over the four sample residues (lanes 16 apart)
  ca += __shfl_xor(ca, 16);
  ca += __shfl_xor(ca, 32);
  cb += __shfl_xor(cb, 16);
  cb += __shfl_xor(cb, 32);
#pragma unroll
  for (int r = 0; r < 4; ++r) s_acc[wv][16 * acc_row(l4, r) + l15] = acc[r];
  if (l4 == 0) {
    s_sa[wv][l15] = ca;
    s_sb[wv][l15] = cb;
  }
  __syncthreads();
  const int er = tid >> 4, ec = tid & 15;
  TileMoments m;
  m.q = ((s_acc[0][tid] + s_acc[1][tid]) + s_acc[2][tid]) + s_acc[3][tid];
  m.sa = ((s_sa[0][er] + s_sa[1][er]) + s_sa[2][er]) + s_sa[3][er];
  m.sb = ((s_sb[0][ec] + s_sb[1][ec]) + s_sb[2][ec]) + s_sb[3][ec];
  if (lane_sa) *lane_sa = ((s_sa[0][l15] + s_sa[1][l15]) + s_sa[2][l15]) + s_sa[3][l15];
  if (lane_sb) *lane_sb = ((s_sb[0][l15] + s_sb[1][l15]) + s_sb[2][l15]) + s_sb[3][l15];
  return m;
}

__device__ __forceinline__ TileMoments small_tile_moments(const double* __restrict__ lifts,
                                                          const double* __restrict__ mean, int n_samples, int p,
                                                          int ti, int tj) {
  const int l15 = threadIdx.x & 15;
  const int a = 16 * ti + l15, b = 16 * tj + l15;
  return small_tile_moments_mu(lifts, mean[a < p ? a : p - 1], mean[b < p ? b : p - 1], n_samples, p, ti, tj);
}

__device__ __forceinline__ void tile_of(int t, int& ti, int& tj) {   // lower tiles, row by row
  ti = 0;
  while (t >= ti + 1) {
    t -= ti + 1;
    ++ti;
  }
  tj = t;
}

__global__ __launch_bounds__(256) void stats_small_kernel(const double* __restrict__ lifts,
                                                          const double* __restrict__ mean, double* __restrict__ buf,
                                                          int n_samples, int p, int accumulate) {
  double* S = buf + 1;
  double* Q = buf + 1 + p;
  int ti, tj;
  tile_of(blockIdx.x, ti, tj);
  const bool diag = ti == tj;
  const int tid = threadIdx.x, er = tid >> 4, ec = tid & 15;
  const int ai = 16 * ti + er, bi = 16 * tj + ec;
  const bool live = ai < p && bi < p;
  const int64_t o = live ? (int64_t)ai * p + bi : 0;
  const double q_old = accumulate ? Q[o] : 0.0;                         // fetched before the moments, used after
  const double s_old = (accumulate && diag && ec == 0 && ai < p) ? S[ai] : 0.0;
  const TileMoments m = small_tile_moments(lifts, mean, n_samples, p, ti, tj);
  if (live) {
    const double v = accumulate ? q_old + m.q : m.q;
    Q[o] = v;
    if (!diag) Q[(int64_t)bi * p + ai] = v;
  }
  if (diag && ec == 0 && ai < p) S[ai] = accumulate ? s_old + m.sa : m.sa;
  if (blockIdx.x == 0 && tid == 0) buf[0] = accumulate ? buf[0] + (double)n_samples : (double)n_samples;
}

// The same tiles with the Chan merge folded in, for a single GPU (nothing to all-reduce between the batch moments and
// the merge): the workgroup of tile (ti, tj) also sums the columns of its two blocks of D, so it can add
//     Q + (n nb / (n + nb) - nb) (S_a / nb) (S_b / nb)
// to its tile of M2 at once; the diagonal tiles write the advanced mean.  The mean (and n) are READ by every tile,
// so the new ones go to a second buffer (mean_out, state_out) that the host swaps in afterwards: one launch, no
// pending buffer, no ticket -- the two launches it replaces were a fifth of a p = 100 step.
// Round 3, second form: the tile's moments by four waves (small_tile_moments), thread (row, column) of the 256 applies the
// merge to its element of M2, fetched at the start of the kernel.
__global__ __launch_bounds__(256) void stats_small_fused_kernel(const double* __restrict__ lifts,
                                                                const double* __restrict__ mean,
                                                                const double* __restrict__ state,
                                                                double* __restrict__ mean_out,
                                                                double* __restrict__ state_out, double* __restrict__ M2,
                                                                int n_samples, int p) {
  int ti, tj;
  tile_of(blockIdx.x, ti, tj);
  const bool diag = ti == tj;
  const int tid = threadIdx.x, er = tid >> 4, ec = tid & 15;
  const int ai = 16 * ti + er, bi = 16 * tj + ec;
  const bool live = ai < p && bi < p;
  const int64_t o = live ? (int64_t)ai * p + bi : 0;
  const double m2_old = M2[o];
  const double n = state[0], nb = (double)n_samples;
  const TileMoments m = small_tile_moments(lifts, mean, n_samples, p, ti, tj);
  const double coef = n * nb / (n + nb) - nb;
  const double inv = 1.0 / nb;
  if (live) {
    // coef * (sa' * sb'), not (coef * sa') * sb': element (a, b) and its mirror inside a diagonal tile must round alike
    const double v = m2_old + (m.q + coef * ((m.sa * inv) * (m.sb * inv)));
    M2[o] = v;
    if (!diag) M2[(int64_t)bi * p + ai] = v;
  }
  if (diag && ec == 0 && ai < p) mean_out[ai] = mean[ai] + m.sa * (1.0 / (n + nb));   // as stats_merge_fused_kernel rounds it
  if (blockIdx.x == 0 && tid == 0) {
    state_out[0] = n + nb;
    state_out[1] = 0.0;     // the fused merge kernel's ticket, kept clear in both buffers
  }
}

hipError_t launch_stats_small_fused(const double* lifts, const double* mean, const double* state, double* mean_out,
                                    double* state_out, double* M2, int n_samples, int p, hipStream_t st) {
  if (!stats_small_fusable(n_samples, p)) return hipErrorInvalidValue;
  const int t16 = (p + 15) / 16, n_tiles = t16 * (t16 + 1) / 2;
  hipLaunchKernelGGL(stats_small_fused_kernel, dim3(n_tiles), dim3(256), 0, st, lifts, mean, state, mean_out, state_out,
                     M2, n_samples, p);
  return hipGetLastError();
}

// SEVERAL chunks of samples folded and merged one after the other in ONE launch (round 5): at p = 100 a chunk is 25 us
// of lift kernel and its statistics launch 4.4 us plus the gap of a dependent launch -- a fifth of the step, and not to
// be hidden behind the next group's kernel (a small_reg workgroup pair leaves no room on its CU).  Nothing has to meet
// between two chunks except the running mean and n, and every tile's workgroup can advance those itself: it sums the
// columns of its own two blocks anyway, with the same operands in the same order as the diagonal tiles do, so the means it
// carries per lane are bit for bit the ones the one-chunk kernel would have read back from memory.  The result -- M2, mean,
// n after the last chunk -- is that of the one-chunk launches in sequence, to the last bit (tests).  mean_snap / n_snap:
// the running mean and n after every chunk ([n_chunks][p], [n_chunks]): what the check of that chunk reads.
__global__ __launch_bounds__(256) void stats_small_multi_kernel(const double* __restrict__ lifts,
                                                                const double* __restrict__ mean,
                                                                const double* __restrict__ state,
                                                                double* __restrict__ mean_out,
                                                                double* __restrict__ state_out, double* __restrict__ M2,
                                                                StatsChunks ch, int p, double* __restrict__ mean_snap,
                                                                double* __restrict__ n_snap) {
  int ti, tj;
  tile_of(blockIdx.x, ti, tj);
  const bool diag = ti == tj;
  const int tid = threadIdx.x, er = tid >> 4, ec = tid & 15, l15 = tid & 15;
  const int ai = 16 * ti + er, bi = 16 * tj + ec;
  const bool live = ai < p && bi < p;
  const int64_t o = live ? (int64_t)ai * p + bi : 0;
  double m2 = M2[o];
  double n = state[0];
  const int a = 16 * ti + l15, b = 16 * tj + l15;
  double mua = mean[a < p ? a : p - 1], mub = mean[b < p ? b : p - 1];     // per lane: the columns it subtracts from
  double mu_row = mean[ai < p ? ai : p - 1];                               // per thread: the mean a diagonal tile writes
  for (int c = 0; c < ch.n; ++c) {
    const double nb = (double)ch.count[c];
    double lsa, lsb;
    const TileMoments m = small_tile_moments_mu(lifts + (int64_t)ch.first[c] * p, mua, mub, ch.count[c], p, ti, tj, &lsa,
                                                &lsb);
    const double coef = n * nb / (n + nb) - nb;
    const double inv = 1.0 / nb;
    m2 = m2 + (m.q + coef * ((m.sa * inv) * (m.sb * inv)));       // as stats_small_fused_kernel rounds it
    const double wgt = 1.0 / (n + nb);
    mua = mua + lsa * wgt;
    mub = mub + lsb * wgt;
    mu_row = mu_row + m.sa * wgt;
    n = n + nb;
    if (mean_snap && diag && ec == 0 && ai < p) mean_snap[(int64_t)c * p + ai] = mu_row;
    if (n_snap && blockIdx.x == 0 && tid == 0) n_snap[c] = n;
    __syncthreads();      // the moments' LDS tiles are written again by the next chunk
  }
  if (live) {
    M2[o] = m2;
    if (!diag) M2[(int64_t)bi * p + ai] = m2;
  }
  if (diag && ec == 0 && ai < p) mean_out[ai] = mu_row;
  if (blockIdx.x == 0 && tid == 0) {
    state_out[0] = n;
    state_out[1] = 0.0;
  }
}

hipError_t launch_stats_small_multi(const double* lifts, const double* mean, const double* state, double* mean_out,
                                    double* state_out, double* M2, const StatsChunks& ch, int p, double* mean_snap,
                                    double* n_snap, hipStream_t st) {
  if (ch.n < 1 || ch.n > StatsChunks::MAX) return hipErrorInvalidValue;
  for (int c = 0; c < ch.n; ++c)
    if (!stats_small_fusable(ch.count[c], p) || ch.first[c] < 0) return hipErrorInvalidValue;
  const int t16 = (p + 15) / 16, n_tiles = t16 * (t16 + 1) / 2;
  hipLaunchKernelGGL(stats_small_multi_kernel, dim3(n_tiles), dim3(256), 0, st, lifts, mean, state, mean_out, state_out,
                     M2, ch, p, mean_snap, n_snap);
  return hipGetLastError();
}

bool stats_small_fusable(int n_samples, int p) { return p >= 1 && p <= 128 && n_samples >= 1 && n_samples <= 512; }

int stats_batch_slices(int n_samples, int p) {
  // a handful of tiles and a few hundred samples: one launch beats slices plus their reduction (every launch
  // costs the host more than this kernel runs)
  if (p <= 128 && n_samples <= 512) return 1;
  const int nt = (p + 63) / 64;
  int z = 512 / (nt * nt);                       // aim at ~2 workgroups per CU
  z = z < (n_samples + 63) / 64 ? z : (n_samples + 63) / 64;   // at least 64 samples per slice
  return z < 1 ? 1 : z;
}

hipError_t launch_stats_batch(const double* lifts, const double* mean, double* buf, int n_samples, int p,
                              int accumulate, double* parts, hipStream_t st) {
  if (n_samples < 1 || p < 1) return hipErrorInvalidValue;
  if (p <= 128 && n_samples <= 512) {
    const int t16 = (p + 15) / 16, n_tiles = t16 * (t16 + 1) / 2;
    hipLaunchKernelGGL(stats_small_kernel, dim3(n_tiles), dim3(256), 0, st, lifts, mean, buf, n_samples, p,
                       accumulate);
    return hipGetLastError();
  }
  const int nt = (p + 63) / 64;
  const int nz = parts ? stats_batch_slices(n_samples, p) : 1;
  const int per = (((n_samples + nz - 1) / nz + 15) / 16) * 16;
  hipLaunchKernelGGL(stats_batch_kernel, dim3(nt, nt, nz), dim3(256), 0, st, lifts, mean, nz > 1 ? parts : buf,
                     n_samples, p, accumulate, per);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || nz == 1) return e;
  const int64_t len = (int64_t)1 + p + (int64_t)p * p;
  const int grid = (int)((len + 255) / 256 < 1024 ? (len + 255) / 256 : 1024);
  hipLaunchKernelGGL(stats_reduce_kernel, dim3(grid), dim3(256), 0, st, parts, nz, len, buf, accumulate);
  return hipGetLastError();
}

// Packing of the pending-batch buffer for the all-reduce at large p: Q is symmetric (a sum of outer products,
// computed tile by tile from commutative products in one sample order, so bitwise symmetric), hence only its upper
// triangle travels:  packed = [n_b, S (p), Q[i][i..p-1] for i = 0..p-1]  -- 1 + p + p (p + 1) / 2 elements
// instead of 1 + p + p^2 (100 MB instead of 200 MB at p = 5000).
__device__ __forceinline__ int64_t tri_off(int64_t i, int64_t j, int64_t p) { return i * p - i * (i - 1) / 2 + (j - i); }

__global__ __launch_bounds__(256) void stats_pack_kernel(const double* __restrict__ buf, double* __restrict__ packed,
                                                         int p) {
  const int64_t total = (int64_t)p * p;
  const double* Q = buf + 1 + p;
  double* T = packed + 1 + p;
  for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
    const int64_t i = o / p, j = o - i * p;
    if (j >= i) T[tri_off(i, j, p)] = Q[o];
    if (o <= p) packed[o] = buf[o];   // n_b and S
  }
}

__global__ __launch_bounds__(256) void stats_unpack_kernel(const double* __restrict__ packed, double* __restrict__ buf,
                                                           int p) {
  const int64_t total = (int64_t)p * p;
  double* Q = buf + 1 + p;
  const double* T = packed + 1 + p;
  for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
    const int64_t i = o / p, j = o - i * p;
    Q[o] = (j >= i) ? T[tri_off(i, j, p)] : T[tri_off(j, i, p)];
    if (o <= p) buf[o] = packed[o];
  }
}

int64_t stats_packed_count(int p) { return (int64_t)1 + p + (int64_t)p * (p + 1) / 2; }

hipError_t launch_stats_pack(const double* buf, double* packed, int p, hipStream_t st) {
  if (p < 1) return hipErrorInvalidValue;
  const int64_t total = (int64_t)p * p;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(stats_pack_kernel, dim3(grid), dim3(256), 0, st, buf, packed, p);
  return hipGetLastError();
}

hipError_t launch_stats_unpack(const double* packed, double* buf, int p, hipStream_t st) {
  if (p < 1) return hipErrorInvalidValue;
  const int64_t total = (int64_t)p * p;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(stats_unpack_kernel, dim3(grid), dim3(256), 0, st, packed, buf, p);
  return hipGetLastError();
}

// Chan et al. pairwise merge.  With delta = S / n_b (batch mean minus running mean):
//   M2 += Q - n_b delta delta^T + (n n_b / (n + n_b)) delta delta^T ;  mean += (n_b / (n + n_b)) delta
// M2 is updated by all workgroups first; mean and n by a second launch (stats_advance).
__global__ __launch_bounds__(256) void stats_merge_m2_kernel(const double* __restrict__ buf,
                                                             const double* __restrict__ state_n,
                                                             double* __restrict__ M2, int p) {
  const double nb = buf[0];
  if (!(nb > 0.0)) return;
  const double n = state_n[0];
  const double coef = n * nb / (n + nb) - nb;
  const double inv = 1.0 / nb;
  const double* S = buf + 1;
  const double* Q = buf + 1 + p;
  const int64_t total = (int64_t)p * p;
  for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
    const int ai = (int)(o / p), bi = (int)(o - (int64_t)ai * p);
    M2[o] += Q[o] + coef * ((S[ai] * inv) * (S[bi] * inv));   // symmetric in (ai, bi)
  }
}

__global__ __launch_bounds__(256) void stats_advance_kernel(const double* __restrict__ buf,
                                                            double* __restrict__ state_n,
                                                            double* __restrict__ mean, int p) {
  const double nb = buf[0];
  if (!(nb > 0.0)) return;
  const double n = state_n[0];
  const double wgt = 1.0 / (n + nb);
  for (int i = threadIdx.x; i < p; i += 256) mean[i] += buf[1 + i] * wgt;
  __syncthreads();
  if (threadIdx.x == 0) state_n[0] = n + nb;
}

// The same merge as ONE launch, for small p (where three launches cost the host more than they run): every
// workgroup updates its share of M2 and clears the part of the pending buffer it consumed; the workgroup that
// finishes LAST (ticket in state[1], agent-scope fences around it) advances mean and n and clears n_b and S, which
// all the others have read by then.  state: [n, ticket, ...].
__global__ __launch_bounds__(256) void stats_merge_fused_kernel(double* __restrict__ buf, double* __restrict__ state,
                                                                double* __restrict__ mean, double* __restrict__ M2,
                                                                int p) {
  __shared__ int s_last;
  const double nb = buf[0];
  const double n = state[0];
  double* S = buf + 1;
  double* Q = buf + 1 + p;
  if (nb > 0.0) {
    const double coef = n * nb / (n + nb) - nb;
    const double inv = 1.0 / nb;
    const int64_t total = (int64_t)p * p;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
      const int ai = (int)(o / p), bi = (int)(o - (int64_t)ai * p);
      M2[o] += Q[o] + coef * ((S[ai] * inv) * (S[bi] * inv));   // symmetric in (ai, bi)
      Q[o] = 0.0;
    }
  }
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned int* ticket = reinterpret_cast<unsigned int*>(state + 1);
    s_last = (atomicAdd(ticket, 1u) == gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  if (nb > 0.0) {
    const double wgt = 1.0 / (n + nb);
    for (int i = threadIdx.x; i < p; i += 256) {
      mean[i] += S[i] * wgt;
      S[i] = 0.0;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (nb > 0.0) state[0] = n + nb;
    buf[0] = 0.0;
    *reinterpret_cast<unsigned int*>(state + 1) = 0u;
  }
}

hipError_t launch_stats_merge(double* buf, double* state_n, double* mean, double* M2, int p, hipStream_t st,
                              bool* cleared) {
  if (p < 1) return hipErrorInvalidValue;
  const int64_t total = (int64_t)p * p;
  if (p <= 256) {
    const int grid = (int)((total + 255) / 256 < 64 ? (total + 255) / 256 : 64);
    hipLaunchKernelGGL(stats_merge_fused_kernel, dim3(grid), dim3(256), 0, st, buf, state_n, mean, M2, p);
    if (cleared) *cleared = true;
    return hipGetLastError();
  }
  if (cleared) *cleared = false;
  const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(stats_merge_m2_kernel, dim3(grid), dim3(256), 0, st, buf, state_n, M2, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(stats_advance_kernel, dim3(1), dim3(256), 0, st, buf, state_n, mean, p);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// theta = L^-T z, row-oriented so that every global read is a contiguous row of L:
//   for j = p-1 .. 0 :  theta_j = w_j / L[j][j] ;  w[0:j] -= theta_j * L[j][0:j]
// One workgroup; one-off cost per problem.
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(1024) void backsolve_kernel(const T* __restrict__ L, double* __restrict__ theta,
                                                         int p, int p_pad, double* __restrict__ wg) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  // the running right-hand side: in LDS, or -- p beyond what LDS holds -- in the global workspace wg (one workgroup:
  // its own writes are visible to it after a fence and a barrier)
  double* wv = wg ? wg : reinterpret_cast<double*>(smem_raw);
  __shared__ double s_t;
  const int tid = threadIdx.x;
  for (int i = tid; i < p; i += 1024) wv[i] = (double)L[cm_off(p_pad, p, i)];
  if (wg) __threadfence_block();
  __syncthreads();
  for (int j = p - 1; j >= 0; --j) {
    if (tid == 0) {
      const double t = wv[j] / (double)L[cm_off(p_pad, j, j)];
      s_t = t;
      theta[j] = t;
    }
    __syncthreads();
    const double t = s_t;
    for (int i = tid; i < j; i += 1024) wv[i] -= t * (double)L[cm_off(p_pad, j, i)];
    if (wg) __threadfence_block();
    __syncthreads();
  }
}

// wg: workspace of p doubles, used when p doubles do not fit a CU's LDS (may be null for smaller p)
hipError_t launch_backsolve(const void* A, double* theta, int p, int p_pad, int f32, hipStream_t st, double* wg) {
  if (p < 1 || p_pad <= p) return hipErrorInvalidValue;
  size_t shmem = sizeof(double) * p;
  if (shmem + 64 > LDS_BYTES_PER_CU) {
    if (!wg) return hipErrorInvalidValue;
    shmem = 0;
  } else {
    wg = nullptr;
  }
  static DynLdsGrant grant_f, grant_d;
  hipError_t e = f32 ? grant_f.ensure(reinterpret_cast<const void*>(backsolve_kernel<float>), shmem)
                     : grant_d.ensure(reinterpret_cast<const void*>(backsolve_kernel<double>), shmem);
  if (e != hipSuccess) return e;
  if (f32)
    hipLaunchKernelGGL(backsolve_kernel<float>, dim3(1), dim3(1024), shmem, st, (const float*)A, theta, p, p_pad, wg);
  else
    hipLaunchKernelGGL(backsolve_kernel<double>, dim3(1), dim3(1024), shmem, st, (const double*)A, theta, p,
                       p_pad, wg);
  return hipGetLastError();
}

}  // namespace lsspa
