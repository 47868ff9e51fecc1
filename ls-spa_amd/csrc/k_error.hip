// Device-side error estimator.
//
// The reference draws 1024 samples of N(0, C_unbiased / n) and reports 0.95-quantiles of |x_a| per
// feature and of ||x||_2 (cvxgrp/ls-spa ls_spa/ls_spa.py:321-341), which costs a p x p Cholesky or,
// on the always-singular sample covariance, an SVD.  With H the n x p matrix of lift vectors and
// Xi ~ N(0, I) of shape 1024 x n,   x = Xi (H - 1 mu^T) / sqrt(n (n - 1))   has exactly that
// covariance, so the draws are one thin GEMM.  Xi comes from the host generator (the stream the
// reference's sampler shares); everything else stays in HBM:
//   draws_kernel     : 64 x 128 tiles of Xi * H on the fp64 MFMA, centring and scaling fused into the
//                      store.  With several GPUs each rank multiplies its own samples' rows of H by
//                      the matching columns of Xi; the partial draws are summed by ONE all-reduce.
//   row_norms_kernel : ||x_d||_2, one wave per draw, fixed summation order
//   quantile_kernel  : one workgroup per feature (+ one for the norms): bitonic sort of 1024 values
//                      in LDS, numpy's default linear-interpolation quantile.
#include "kernels.h"
#include "tiles.h"

namespace lsspa {

constexpr int ND = 1024;  // draws, as in the reference

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

// norms[d] = ||draws[d][0..p)||_2, one wave per draw
__global__ __launch_bounds__(256) void row_norms_kernel(const double* __restrict__ draws, int ldd, int p,
                                                        double* __restrict__ norms) {
  const int lane = threadIdx.x & 63, d = blockIdx.x * 4 + (threadIdx.x >> 6);
  double v = 0.0;
  for (int a = lane; a < p; a += 64) {
    const double x = draws[(int64_t)d * ldd + a];
    v += x * x;
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
  if (lane == 0) norms[d] = sqrt(v);
}

// numpy.quantile(v, q) with the default 'linear' method on ND values
__global__ __launch_bounds__(512) void quantile_kernel(const double* __restrict__ draws, int ldd,
                                                       const double* __restrict__ norms, int p, double q,
                                                       double* __restrict__ out /*[p + 1]*/) {
  __shared__ double s[ND];
  const int a = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < ND; i += 512) s[i] = (a < p) ? fabs(draws[(int64_t)i * ldd + a]) : norms[i];
  __syncthreads();
  for (int k = 2; k <= ND; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < ND; i += 512) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const double x = s[i], y = s[ixj];
          const bool up = (i & k) == 0;
          if ((x > y) == up) {
            s[i] = y;
            s[ixj] = x;
          }
        }
      }
      __syncthreads();
    }
  if (tid == 0) {
    const double pos = q * (ND - 1);
    const int lo = (int)floor(pos);
    const int hi = lo + 1 < ND ? lo + 1 : ND - 1;
    const double t = pos - lo;
    const double va = s[lo], vb = s[hi];
    // numpy's _lerp: a + (b - a) t, evaluated from b's side when t >= 0.5
    out[a] = (t >= 0.5) ? vb - (vb - va) * (1.0 - t) : va + (vb - va) * t;
  }
}

hipError_t launch_error_draws(const double* Xi, int ldxi, const double* H, int ldh, int n_pad,
                              const double* mean, double scale, int p, double* draws, int ldd, hipStream_t st) {
  const int n_tiles = (p + 127) / 128;
  if (p < 1 || n_pad < KCH || n_pad % KCH != 0 || ldxi < n_pad || (ldxi & 1) || ldh % 128 != 0 ||
      ldh < n_tiles * 128 || ldd < n_tiles * 128)
    return hipErrorInvalidValue;
  hipLaunchKernelGGL(draws_kernel, dim3(ND / 64, n_tiles), dim3(256), 0, st, Xi, ldxi, H, ldh, n_pad, mean, scale,
                     p, draws, ldd);
  return hipGetLastError();
}

hipError_t launch_error_quantiles(const double* draws, int ldd, int p, double* norms, double* out,
                                  hipStream_t st) {
  if (p < 1 || ldd < p) return hipErrorInvalidValue;
  hipLaunchKernelGGL(row_norms_kernel, dim3(ND / 4), dim3(256), 0, st, draws, ldd, p, norms);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(quantile_kernel, dim3(p + 1), dim3(512), 0, st, draws, ldd, norms, p, 0.95, out);
  return hipGetLastError();
}

}  // namespace lsspa
