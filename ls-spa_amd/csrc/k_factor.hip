// Per-ordering factorisation kernels: permuted gather, blocked left-looking Cholesky
// (diagonal-block and panel steps, with V^T = L_t^T L^-T as extra block rows) and the strip triangular solve.  Templated on the element type
// of the work matrices (double: default; float: fp32 mode), see tiles.h.
//
// What they replace in the reference (cvxgrp/ls-spa, ls_spa/ls_spa.py):
//   gather      -> X_train[:, perm], X_test[:, perm]                    (:275-276)
//   chol_*      -> np.linalg.qr of the permuted train factor            (:275)
//   strip       -> solve_triangular + the X @ T product                 (:279-283)
// in the Gram form  G_pi = L L^T  (L = R^T of the reference's QR up to row signs).
#include "kernels.h"
#include "tiles.h"

namespace lsspa {

// =====================================================================================
// gather:  A[mat][i][j] = S[perm[i]][perm[j]]  (j <= i),  row p = s[perm[.]] | aug,
//          rows > p = identity.  One workgroup walks GROWS output rows; each source row
//          is read once, coalesced, into LDS and the permuted columns are picked there,
//          so both the global read and the global write are contiguous.  The source Gram
//          matrices are fp64 in both modes; fp32 mode rounds on the way out.
// =====================================================================================
constexpr int GROWS = 16;

// PAIRED: ordering 2 s + 1 is ordering 2 s reversed (an antithetical pair).  With pi' = reverse(pi),
// row p-1-i of the second matrix is  S[pi_i][pi_{p-1-j'}],  j' <= p-1-i : the OTHER end of the same source
// row that row i of the first matrix takes its entries from.  One staging of the source row in LDS
// therefore serves both matrices, which halves the row reads.
template <typename T, bool PAIRED, typename ST>
__global__ __launch_bounds__(256) void gather_kernel(GatherArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  ST* rowbuf = reinterpret_cast<ST*>(smem_raw);                                  // [p_pad]
  int32_t* sperm = reinterpret_cast<int32_t*>(smem_raw + sizeof(ST) * a.p_pad);  // [p_pad]
  typedef ST st2 __attribute__((ext_vector_type(2)));
  constexpr int VE = Tr<T>::VE;
  typedef typename Tr<T>::vec_t vec_t;

  const int tid = threadIdx.x;
  const int n_slots = PAIRED ? a.n_ord / 2 : a.n_ord;
  const int src = blockIdx.y / n_slots;
  const int ord = (blockIdx.y - src * n_slots) * (PAIRED ? 2 : 1);
  const int mat = src * a.n_ord + ord;
  const int i0 = blockIdx.x * GROWS;
  const int p = a.p, p_pad = a.p_pad;
  const int32_t* perm = a.perms + (int64_t)ord * p;
  const ST* S;
  if constexpr (sizeof(ST) == 8) S = reinterpret_cast<const ST*>(a.S[src]);
  else S = reinterpret_cast<const ST*>(a.Sf[src]);
  const double* svec = a.s[src];
  T* out = static_cast<T*>(a.A) + (int64_t)mat * p_pad * p_pad;   // chunk-major, see tiles.h
  T* out2 = out + (int64_t)p_pad * p_pad;                           // the reversed ordering's matrix
  double* d0 = a.diag0 + (int64_t)mat * p_pad;
  double* d02 = d0 + p_pad;

  // permutation entries this workgroup can touch: all of them when it also writes the mirror rows
  const int jmax = PAIRED ? p : min(i0 + GROWS, p);
  for (int j = tid; j < jmax; j += 256) sperm[j] = perm[j];
  __syncthreads();

  for (int ii = 0; ii < GROWS; ++ii) {
    const int i = i0 + ii;
    if (i >= p_pad) break;
    const int jend = min(((i + 1 + NB - 1) / NB) * NB, p_pad);  // zero-fill to the block edge
    if (i < p) {
      const ST* srow = S + (int64_t)sperm[i] * a.ld_src;
      __syncthreads();  // previous row's picks are done
      for (int c = 2 * tid; c < p; c += 512) {
        if (c + 1 < p) {
          *reinterpret_cast<st2*>(rowbuf + c) = *reinterpret_cast<const st2*>(srow + c);
        } else {
          rowbuf[c] = srow[c];
        }
      }
      __syncthreads();
      for (int j = VE * tid; j < jend; j += VE * 256) {   // VE consecutive j stay inside one 16-column chunk
        vec_t v;
#pragma unroll
        for (int e = 0; e < VE; ++e) {   // unconditional LDS reads (index clamped), value selected afterwards
          const ST rv = rowbuf[sperm[min(j + e, i)]];
          v[e] = (j + e <= i) ? (T)rv : (T)0;
        }
        __builtin_nontemporal_store(v, reinterpret_cast<vec_t*>(out + cm_off(p_pad, i, j)));
      }
      if (tid == 0) d0[i] = rowbuf[sperm[i]];
      if (PAIRED) {
        const int i2 = p - 1 - i;
        const int jend2 = min(((i2 + 1 + NB - 1) / NB) * NB, p_pad);
        for (int j = VE * tid; j < jend2; j += VE * 256) {
          vec_t v;
#pragma unroll
          for (int e = 0; e < VE; ++e) {
            const ST rv = rowbuf[sperm[p - 1 - min(j + e, i2)]];
            v[e] = (j + e <= i2) ? (T)rv : (T)0;
          }
          __builtin_nontemporal_store(v, reinterpret_cast<vec_t*>(out2 + cm_off(p_pad, i2, j)));
        }
        if (tid == 0) d02[i2] = rowbuf[sperm[i]];
      }
    } else if (i == p) {
      for (int j = tid; j < jend; j += 256) {
        out[cm_off(p_pad, i, j)] = (T)((j < p) ? svec[sperm[j]] : (j == p ? a.aug[src] : 0.0));
        if (PAIRED) out2[cm_off(p_pad, i, j)] = (T)((j < p) ? svec[sperm[p - 1 - j]] : (j == p ? a.aug[src] : 0.0));
      }
      if (tid == 0) {
        d0[i] = a.aug[src];
        if (PAIRED) d02[i] = a.aug[src];
      }
    } else {
      for (int j = tid; j < jend; j += 256) {
        out[cm_off(p_pad, i, j)] = (j == i) ? (T)1 : (T)0;
        if (PAIRED) out2[cm_off(p_pad, i, j)] = (j == i) ? (T)1 : (T)0;
      }
      if (tid == 0) {
        d0[i] = 1.0;
        if (PAIRED) d02[i] = 1.0;
      }
    }
  }
}

// The same gather for feature counts whose source row and ordering no longer fit the 160 KB of LDS of a CU
// (12 B per padded feature: p > 13567).  The source row is staged in SEGMENTS of `seg` entries; for each segment the
// workgroup walks the row's columns j <= i (the ordering is read from memory, coalesced) and writes the entries whose
// source column lies in the segment.  Every output element is written exactly once, by element-wise stores: a slow
// path (several passes over the columns per row) for shapes the fast kernel cannot take, so that p is bounded by HBM,
// not by LDS (the reference has no limit, ls_spa/ls_spa.py:163).  Unpaired: every ordering is gathered on its own.
template <typename T, typename ST>
__global__ __launch_bounds__(256) void gather_seg_kernel(GatherArgs a, int seg) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  ST* rowbuf = reinterpret_cast<ST*>(smem_raw);                                  // [seg]
  const int tid = threadIdx.x;
  const int src = blockIdx.y / a.n_ord;
  const int ord = blockIdx.y - src * a.n_ord;
  const int mat = src * a.n_ord + ord;
  const int i0 = blockIdx.x * GROWS;
  const int p = a.p, p_pad = a.p_pad;
  const int32_t* __restrict__ perm = a.perms + (int64_t)ord * p;
  const ST* S;
  if constexpr (sizeof(ST) == 8) S = reinterpret_cast<const ST*>(a.S[src]);
  else S = reinterpret_cast<const ST*>(a.Sf[src]);
  const double* svec = a.s[src];
  T* out = static_cast<T*>(a.A) + (int64_t)mat * p_pad * p_pad;   // chunk-major, see tiles.h
  double* d0 = a.diag0 + (int64_t)mat * p_pad;
  for (int ii = 0; ii < GROWS; ++ii) {
    const int i = i0 + ii;
    if (i >= p_pad) break;
    const int jend = min(((i + 1 + NB - 1) / NB) * NB, p_pad);  // zero-fill to the block edge
    if (i < p) {
      const int pi = perm[i];
      const ST* srow = S + (int64_t)pi * a.ld_src;
      for (int j = i + 1 + tid; j < jend; j += 256) out[cm_off(p_pad, i, j)] = (T)0;
      for (int seg0 = 0; seg0 < p; seg0 += seg) {
        const int n = min(seg, p - seg0);
        __syncthreads();  // the previous segment's picks are done
        for (int c = tid; c < n; c += 256) rowbuf[c] = srow[seg0 + c];
        __syncthreads();
        for (int j = tid; j <= i; j += 256) {
          const int pj = perm[j] - seg0;
          if (pj >= 0 && pj < n) out[cm_off(p_pad, i, j)] = (T)rowbuf[pj];
        }
      }
      if (tid == 0) d0[i] = (double)srow[pi];
    } else if (i == p) {
      for (int j = tid; j < jend; j += 256)
        out[cm_off(p_pad, i, j)] = (T)((j < p) ? svec[perm[j]] : (j == p ? a.aug[src] : 0.0));
      if (tid == 0) d0[i] = a.aug[src];
    } else {
      for (int j = tid; j < jend; j += 256) out[cm_off(p_pad, i, j)] = (j == i) ? (T)1 : (T)0;
      if (tid == 0) d0[i] = 1.0;
    }
  }
}

// Largest feature count the per-ordering kernels take.  The fast gather keeps one source row (8 B an entry in the fp64
// path) and the ordering (4 B an entry) of p_pad entries in the 160 KB of LDS of a CU: p <= 13567; beyond that the
// segmented gather above takes over, and what bounds p is memory -- and 32-bit element counts of one work matrix
// (p_pad^2 < 2^31).
int max_features() { return 32767; }

template <typename T, typename ST>
static hipError_t launch_gather_seg(const GatherArgs& a, hipStream_t st) {
  static DynLdsGrant grant;   // one per instantiation
  const int seg = (int)(128 * 1024 / sizeof(ST));
  const size_t shmem = sizeof(ST) * (size_t)seg;
  hipError_t e = grant.ensure(reinterpret_cast<const void*>(gather_seg_kernel<T, ST>), shmem);
  if (e != hipSuccess) return e;
  dim3 grid((a.p_pad + GROWS - 1) / GROWS, a.n_ord * a.n_src);
  hipLaunchKernelGGL((gather_seg_kernel<T, ST>), grid, dim3(256), shmem, st, a, seg);
  return hipGetLastError();
}

template <typename T, bool PAIRED, typename ST>
static hipError_t launch_gather_as(const GatherArgs& a, dim3 grid, size_t shmem, hipStream_t st) {
  static DynLdsGrant grant;   // one per instantiation
  hipError_t e = grant.ensure(reinterpret_cast<const void*>(gather_kernel<T, PAIRED, ST>), shmem);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((gather_kernel<T, PAIRED, ST>), grid, dim3(256), shmem, st, a);
  return hipGetLastError();
}

hipError_t launch_gather(const GatherArgs& a, hipStream_t st) {
  if (a.p < 1 || a.p_pad % NB != 0 || a.p_pad <= a.p || a.n_ord < 1 || a.n_src < 1 || a.n_src > 2 ||
      (a.ld_src & 1) || (a.paired && (a.n_ord & 1)))
    return hipErrorInvalidValue;
  const bool srcf = a.f32 && a.Sf[0] != nullptr && (a.n_src == 1 || a.Sf[1] != nullptr);
  const size_t shmem = (srcf ? sizeof(float) : sizeof(double)) * a.p_pad + sizeof(int32_t) * a.p_pad;
  if (shmem > LDS_BYTES_PER_CU) {   // the source row and the ordering do not fit a CU's LDS: segmented gather
    if (srcf) return launch_gather_seg<float, float>(a, st);
    if (a.f32) return launch_gather_seg<float, double>(a, st);
    return launch_gather_seg<double, double>(a, st);
  }
  dim3 grid((a.p_pad + GROWS - 1) / GROWS, (a.paired ? a.n_ord / 2 : a.n_ord) * a.n_src);
  // aug row reads sperm[j] for all j < p: the workgroup holding row p must have them all
  // (jmax = min(i0 + GROWS, p) = p there), so nothing else to arrange.
  if (a.paired) {
    if (srcf) return launch_gather_as<float, true, float>(a, grid, shmem, st);
    if (a.f32) return launch_gather_as<float, true, double>(a, grid, shmem, st);
    return launch_gather_as<double, true, double>(a, grid, shmem, st);
  }
  if (srcf) return launch_gather_as<float, false, float>(a, grid, shmem, st);
  if (a.f32) return launch_gather_as<float, false, double>(a, grid, shmem, st);
  return launch_gather_as<double, false, double>(a, grid, shmem, st);
}

__global__ __launch_bounds__(256) void to_f32_kernel(const double* __restrict__ src, float* __restrict__ dst,
                                                     int64_t count) {
  for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < count; o += (int64_t)gridDim.x * 256)
    dst[o] = (float)src[o];
}

hipError_t launch_to_f32(const double* src, float* dst, int64_t count, hipStream_t st) {
  if (count < 1) return hipErrorInvalidValue;
  const int grid = (int)((count + 255) / 256 < 2048 ? (count + 255) / 256 : 2048);
  hipLaunchKernelGGL(to_f32_kernel, dim3(grid), dim3(256), 0, st, src, dst, count);
  return hipGetLastError();
}

// copy the 64 x 64 block at (r0, c0) of a chunk-major matrix into LDS with stride DI_LD, 256 threads;
// neg != 0 stores the negated block
template <typename T, int NT = 256>
__device__ __forceinline__ void load_block64_cm(T* lds, const T* __restrict__ A, int p_pad, int r0, int c0,
                                                int tid, bool neg) {
  typedef typename Tr<T>::vec_t vec_t;
  constexpr int VE = Tr<T>::VE, VPR = 64 / VE, NV = 4096 / VE;
#pragma unroll
  for (int q = 0; q < NV / NT; ++q) {
    const int idx = tid + NT * q;
    const int row = idx / VPR, cv = idx % VPR;
    vec_t v = *reinterpret_cast<const vec_t*>(A + cm_off(p_pad, r0 + row, c0 + VE * cv));
    if (neg) v = -v;
    Tr<T>::lds_store(lds + row * DI_LD + VE * cv, v);
  }
}

// =====================================================================================
// Blocked Cholesky  G_pi = L L^T  of the per-ordering work matrices, panels of TWO 64-wide block columns
// (128 wide), left-looking:
//   chol_panel2, step Jo : for a 128-row tile I below the panel, one k-loop over the columns left of
//        the panel accumulates  C = A[I, J:J+2] - sum_{K<J} L[I,K] L[J:J+2,K]^T  for both block
//        columns at once (16 flop per operand byte), then, on the accumulators,
//            X1 = C1 L11^-T ;  C2 -= X1 L21^T ;  X2 = C2 L22^-T
//        (transposed: the accumulators are the B operand of the next product).  Tile 0 -- the next
//        panel's diagonal block -- then applies the whole row panel to its diagonal block and factors it.
//   factor_diag128 : L11, L11^-1 by the carried-identity elimination; L21 = A21 L11^-T and
//        A22 -= L21 L21^T on the matrix pipe; L22, L22^-1 by the elimination again.
//   X tiles (round 4, tri mode): V^T = L_t^T L^-T  (V = L^-1 L_t, the quantity the lift scan walks) is computed
//        as EXTRA BLOCK ROWS of the training factorisation.  With B = L_t^T stacked under G_pi the panel step
//            X[:, J] = ( B[:, J] - sum_{K<J} X[:, K] L[J, K]^T ) L[J, J]^-T
//        is the tile body above, with three differences: the tile starts from -B[I', J] = -L_t[J, I']^T (read straight
//        into accumulator layout: no transposition), its k-loop starts at the tile's own row block (X is upper
//        triangular: X[I', K] = 0 for K < I'), and there is no diagonal block to update.  Tile (I', J) needs
//        L_t[J, I'] (final after launch J - 1), L[J, 0:J) and Dinv_J (what the L tiles of launch J need): it runs in
//        launch J, next to them, sharing the panel-row operand through the L2; launch p_pad/128 - 1 has X tiles only.
//        This replaces the strip kernel (a workgroup walking a 128-column strip of V top-down, eight dependent solve
//        epilogues in a row) in tri mode.
// =====================================================================================
// lower tiles (ti >= tj) of the 8 x 8 grid of 16 x 16 tiles of a 128 x 128 block, row by row; wave w owns 9 of them.
// The (ti, tj) of tile t come out of two packed constants (3 bits an entry) with scalar shifts: as a table in
// constant memory every lookup was a memory load whose wait also drained the loads issued before it -- nine
// dependent round trips in front of the diagonal update of every panel tile.
__device__ __forceinline__ void syrk_tile(int t, int& ti, int& tj) {
  constexpr unsigned long long TI_LO = 0x5b6db2491b6d2448ull, TI_HI = 0x1ffffffb6db6ull;   // entries 0..20, 21..35
  constexpr unsigned long long TJ_LO = 0x58d111a21a211040ull, TJ_HI = 0x1f58d11ac688ull;
  const int sh = 3 * (t < 21 ? t : t - 21);
  ti = (int)(((t < 21 ? TI_LO : TI_HI) >> sh) & 7);
  tj = (int)(((t < 21 ? TJ_LO : TJ_HI) >> sh) & 7);
}

// the same enumeration as compile-time functions (row ti of the t-th lower tile, row by row)
__host__ __device__ constexpr int syrk_ti_c(int t) {
  int i = 0;
  while ((i + 1) * (i + 2) / 2 <= t) ++i;
  return i;
}
__host__ __device__ constexpr int syrk_tj_c(int t) { return t - syrk_ti_c(t) * (syrk_ti_c(t) + 1) / 2; }

// acc[4 h + xp][y] <- sign * sum_{x <= xp} D[xp][x] acc[4 h + x][y]  (D lower triangular, 64 x 64 in LDS),
// one column tile y at a time so that only four extra tiles are live
template <typename T>
__device__ __forceinline__ void tri_mult_inplace(typename Tr<T>::acc_t (&acc)[8][2], const int half,
                                                 const T* s_d, const T sign, const int l15, const int l4) {
  typedef typename Tr<T>::acc_t acc_t;
#pragma unroll
  for (int y = 0; y < 2; ++y) {
    acc_t t[4];
#pragma unroll
    for (int xp = 0; xp < 4; ++xp) t[xp] = Tr<T>::zero();
#pragma unroll
    for (int xp = 0; xp < 4; ++xp) {
#pragma unroll
      for (int x = 0; x <= xp; ++x)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const T av = sign * s_d[(16 * xp + l15) * DI_LD + 16 * x + Tr<T>::acc_row(l4, r)];
          t[xp] = Tr<T>::mfma(av, acc[4 * half + x][y][r], t[xp]);
        }
      __builtin_amdgcn_sched_barrier(0);   // keep the scheduler from hoisting every LDS read to the top
    }
#pragma unroll
    for (int xp = 0; xp < 4; ++xp) acc[4 * half + xp][y] = t[xp];
  }
}

// acc[4 + xp][y] += sum_x S[xp][x] acc[x][y]   (S a full 64 x 64 block in LDS)
template <typename T>
__device__ __forceinline__ void full_mult_lower_half(typename Tr<T>::acc_t (&acc)[8][2], const T* s_d,
                                                     const int l15, const int l4) {
#pragma unroll
  for (int xp = 0; xp < 4; ++xp) {
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const T av = s_d[(16 * xp + l15) * DI_LD + 16 * x + Tr<T>::acc_row(l4, r)];
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[4 + xp][y] = Tr<T>::mfma(av, acc[x][y][r], acc[4 + xp][y]);
      }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---------------------------------------------------------------------------------------------
// Blocked factorisation of a 64 x 64 diagonal block with its inverse, 16 x 16 sub-blocks:
//   for kb = 0..3:  (1) one wave factors the 16 x 16 diagonal sub-block in registers (carried-identity
//                       elimination, no workgroup barrier: rows and pivots travel by lane shuffles);
//                   (2) L[ib][kb] = T[ib][kb] Ld^-T (ib > kb)  and  X[kb][jb] = Ld^-1 Y[kb][jb] (jb < kb);
//                   (3) T[ib][jb] -= L[ib][kb] L[jb][kb]^T  and  Y[ib][jb] -= L[ib][kb] X[kb][jb]
//                       ((2), (3): 16 x 16 x 16 tile products on the matrix pipe, one tile per wave and turn),
// i.e. forward substitution on [T | I] block by block.  About a dozen barriers instead of the 64 of the
// column-at-a-time sweep -- this routine sits on the critical path of every panel launch.
// LDS: s_t  64 x DI_LD  the block; its unused upper 16 x 16 blocks (jb, ib) hold Y[ib][jb], ib > jb
//      s_x  4 x 16 x XD_LD  the diagonal sub-blocks of the inverse; then 16 pivots' 1/sqrt(d)
// ---------------------------------------------------------------------------------------------
// Phase stamps of the panel kernel (tools/panel_probe.hip; compiled out of the library): 100 MHz wall clock at the
// phase boundaries of a few workgroups spread over the grid, [workgroup slot][phase].
constexpr int XD_LD = 17;
constexpr int FB_SX_ELEMS = 4 * 16 * XD_LD + 16;

// one wave: factor the 16 x 16 block at s_blk (stride DI_LD, lower part valid) in place (L, zeros above the
// diagonal) and write its inverse to s_inv (16 x XD_LD).  tol_lane: lane l holds the pivot threshold of row l & 15
// (piv_tol x the row's original diagonal entry).
// The elimination runs on the matrix pipe with the block in accumulator layout (tiles.h: factor16_acc); here the
// block is brought into that layout (the full symmetric block from its lower part), and L = (unscaled columns) /
// sqrt(pivots) and L^-1 = (unscaled rows of Y) / sqrt(pivots) are written back.
template <typename T>
__device__ __forceinline__ void wave_factor16(T* s_blk, T* s_inv, T* s_dd, const double tol_lane, int lane,
                                              int& bad) {
  typedef typename Tr<T>::acc_t acc_t;
  const int l15 = lane & 15, l4 = lane >> 4;
  acc_t t, y;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = Tr<T>::acc_row(l4, r);
    t[r] = (l15 <= row) ? s_blk[row * DI_LD + l15] : s_blk[l15 * DI_LD + row];
    y[r] = (row == l15) ? (T)1 : (T)0;
  }
  // The one-pivot-a-step sweep here (SEQ), although the four-pivot form (tiles.h: factor16_acc_b4) is 1.5 x faster in
  // isolation and what the small-problem kernels use: measured in this file's kernels (round 5, alternating processes on
  // one box, C3) it makes the diagonal launch 0.0817 against 0.0925 ms and the panel launches 5.80 against 5.875 ms a
  // step when each kernel runs alone -- and the two-lane step 5.938-5.962 against 5.928-5.934 ms: beside a
  // neighbour's k-loop a phase costs its fp64 VECTOR instruction count (they share the matrix pipe), and the four-pivot
  // form trades matrix instructions for vector ones (about 13 a pivot against 9).
  factor16_acc<T, true>(t, y, tol_lane, lane, bad);
  // 1 / L[j][j] = 1 / sqrt(pivot j): the pivot sits on the diagonal of t.  One reciprocal square root per lane (the
  // square-root-then-divide form, evaluated for every register under a predicate, cost as much as the sweep itself)
  T dj;
  const bool holds = acc_diag<T>(t, l15, l4, dj);
  const T rs_mine = fast_rsqrt<T>(dj);
  if (holds) s_dd[l15] = rs_mine;
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = Tr<T>::acc_row(l4, r);
    const T rs_row = s_dd[row];
    // t holds (row, col = l15): for col >= row the unscaled L[col][row]; y holds (L^-1)[row][col] L[row][row], col <= row
    if (l15 > row) s_blk[l15 * DI_LD + row] = t[r] * rs_row;          // L[col][row], strictly lower
    if (l15 == row) s_blk[row * DI_LD + row] = t[r] * rs_row;          // L[row][row] = sqrt(pivot)
    if (l15 > row) s_blk[row * DI_LD + l15] = (T)0;                    // zeros above the diagonal
    s_inv[row * XD_LD + l15] = (l15 <= row) ? y[r] * rs_row : (T)0;
  }
}

// out(16x16) = sign * A * B (+ C): A[m][k] = a[m * lda + k]; B[k][n] = bt ? b[n * ldb + k] : b[k * ldb + n];
// result / accumulation tile at c (stride ldc).  One wave.
template <typename T>
__device__ __forceinline__ void tile16_mma(const T* a, int lda, const T* b, int ldb, bool bt, T* c, int ldc,
                                           bool accumulate, T sign, int lane) {
  typedef typename Tr<T>::acc_t acc_t;
  const int l15 = lane & 15, l4 = lane >> 4;
  acc_t o = Tr<T>::zero();
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const int kx = 4 * kk + l4;
    const T av = sign * a[l15 * lda + kx];
    const T bv = bt ? b[l15 * ldb + kx] : b[kx * ldb + l15];
    o = Tr<T>::mfma(av, bv, o);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    T* dst = c + Tr<T>::acc_row(l4, r) * ldc + l15;
    *dst = accumulate ? *dst + o[r] : o[r];
  }
}

// Factor the 64 x 64 diagonal block of M at (r0, r0) (chunk-major, lower part meaningful) in place and write
// its inverse to Dg (row-major 64 x 64).  NT threads, the first four waves work.  s_t: >= 64 * DI_LD elements,
// s_x: >= FB_SX_ELEMS elements of LDS.
// first half: the block into LDS (lower 16 x 16 blocks from memory, the rest zero: Y starts as the identity),
// in two moves so that the fetch can be issued long before the LDS region is free
template <typename T, int NT>
struct Block64Regs {
  typename Tr<T>::vec_t v[4096 / Tr<T>::VE / NT];
};

template <typename T, int NT>
__device__ __forceinline__ void factor_block64_fetch(Block64Regs<T, NT>& b, const T* __restrict__ M, int p_pad, int r0,
                                                     int tid) {
  constexpr int VE = Tr<T>::VE, VPR = 64 / VE, NV = 4096 / VE;
#pragma unroll
  for (int qq = 0; qq < NV / NT; ++qq) {
    const int idx = tid + NT * qq;
    const int row = idx / VPR, col = VE * (idx % VPR);   // VE consecutive columns stay inside one 16-column block
    b.v[qq] = *reinterpret_cast<const typename Tr<T>::vec_t*>(M + cm_off(p_pad, r0 + row, r0 + col));
  }
}

template <typename T, int NT>
__device__ __forceinline__ void factor_block64_put(const Block64Regs<T, NT>& b, T* s_t, int tid) {
  constexpr int VE = Tr<T>::VE, VPR = 64 / VE, NV = 4096 / VE;
#pragma unroll
  for (int qq = 0; qq < NV / NT; ++qq) {
    const int idx = tid + NT * qq;
    const int row = idx / VPR, col = VE * (idx % VPR);
    Tr<T>::lds_store(s_t + row * DI_LD + col, ((col >> 4) > (row >> 4)) ? Tr<T>::vzero() : b.v[qq]);
  }
  __syncthreads();
}

template <typename T, int NT>
__device__ __forceinline__ void factor_block64_load(const T* __restrict__ M, int p_pad, int r0, T* s_t, int tid) {
  Block64Regs<T, NT> b;
  factor_block64_fetch<T, NT>(b, M, p_pad, r0, tid);
  factor_block64_put<T, NT>(b, s_t, tid);
}

// second half: factor the block held in s_t, store L to M and the inverse to Dg.  On return s_t still holds L
// (lower blocks) and the off-diagonal blocks of the inverse (block (jb, ib) = X[ib][jb]), s_x its diagonal blocks.
template <typename T, int NT>
__device__ __forceinline__ void factor_block64_core(T* __restrict__ M, int p_pad, int r0, T* __restrict__ Dg,
                                                    const double tol64, int32_t* __restrict__ info, T* s_t, T* s_x,
                                                    int tid) {
  // tol64: lane l holds the pivot threshold of row r0 + l (fetched from memory by the caller BEFORE the block was
  // staged: read where it is needed, the global load sat in front of every one of the four 16 x 16 eliminations)
  const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);   // the wave index as the scalar it is
  T* const s_dd = s_x + 4 * 16 * XD_LD;
  int bad = 0;
  // Block step kb: (1) the 16 x 16 pivot chain of diagonal tile kb on ONE wave; (2) the panel tiles L[ib][kb] and the
  // inverse tiles X[kb][jb]; (3) the trailing tiles.  LOOK-AHEAD (round 4): (3) of step kb - 1 and (1) of step kb run side
  // by side -- the wave that has the chain updates diagonal tile (kb, kb) alone and starts at once, the other three waves
  // share the remaining trailing tiles meanwhile (8 / 6 / 3 of them: three rounds / two / one, as the four waves had
  // before): the 128 x 128 factorisation 52.4 -> 49.5 us alone on its CU, the diagonal launch 0.103 -> 0.096 ms.
  // The pivot chain is a sequence of DEPENDENT matrix instructions (two per pivot, a third of the pipe's time): with
  // priority over the co-resident workgroup's k-loop, which fills the pipe, each issues when it is ready instead of
  // queueing behind independent work that can wait.  The four chains of a 64 x 64 block go to the four waves in turn:
  // the co-resident workgroup's waves meet at two barriers a chunk, so it runs at the pace of its slowest wave -- all
  // four chains on wave 0 slowed ITS SIMD's neighbour, and with it the whole neighbour, four times as long as each
  // SIMD's share does (6.40 -> 6.36 ms a C3 step; priority 0 instead of 3 here: no difference).
#pragma unroll 1
  for (int kb = 0; kb < 4; ++kb) {
    const int wc = kb & 3;                   // this step's chain wave
    if (w == wc) {
      if (kb > 0) {                          // T[kb][kb] -= L[kb][kb-1] L[kb][kb-1]^T: the one trailing tile the chain needs
        const T* const lkk = s_t + (16 * kb) * DI_LD + 16 * (kb - 1);
        tile16_mma<T>(lkk, DI_LD, lkk, DI_LD, true, s_t + (16 * kb) * DI_LD + 16 * kb, DI_LD, true, (T)-1, lane);
        __builtin_amdgcn_wave_barrier();
      }
      __builtin_amdgcn_s_setprio(3);
      wave_factor16<T>(s_t + (16 * kb) * DI_LD + 16 * kb, s_x + kb * 16 * XD_LD, s_dd,
                       __shfl(tol64, 16 * kb + (lane & 15)), lane, bad);
      __builtin_amdgcn_s_setprio(0);
    } else if (kb > 0) {
      // (3) of step kp = kb - 1 without its tile (kb, kb): trailing tiles (ib >= jb > kp) and Y tiles (ib > kp, jb <= kp)
      const int kp = kb - 1;
      const T* const invp = s_x + kp * 16 * XD_LD;
      const int wr = (w - wc - 1) & 3;       // 0, 1, 2 among the three helpers
      int t = 0;
      for (int ib = kp + 1; ib < 4; ++ib) {
        const T* lik = s_t + (16 * ib) * DI_LD + 16 * kp;
        for (int jb = kp + 1; jb <= ib; ++jb) {
          if (ib == kb) continue;            // (kb, kb): the chain's wave has it
          if ((t++ % 3) == wr)               // T[ib][jb] -= L[ib][kp] * L[jb][kp]^T
            tile16_mma<T>(lik, DI_LD, s_t + (16 * jb) * DI_LD + 16 * kp, DI_LD, true,
                          s_t + (16 * ib) * DI_LD + 16 * jb, DI_LD, true, (T)-1, lane);
        }
        for (int jb = 0; jb <= kp; ++jb)
          if ((t++ % 3) == wr) {             // Y[ib][jb] -= L[ib][kp] * X[kp][jb]
            const T* xb = (jb == kp) ? invp : s_t + (16 * jb) * DI_LD + 16 * kp;
            tile16_mma<T>(lik, DI_LD, xb, (jb == kp) ? XD_LD : DI_LD, false,
                          s_t + (16 * jb) * DI_LD + 16 * ib, DI_LD, true, (T)-1, lane);
          }
      }
    }
    __syncthreads();
    T* const inv = s_x + kb * 16 * XD_LD;
    // (2): 3 - kb panel tiles and kb inverse tiles: three tiles in all, one per wave
    if (w < 3) {
      if (w < 3 - kb) {
        const int ib = kb + 1 + w;           // L[ib][kb] = T[ib][kb] * Ld^-T, in place
        T* tb = s_t + (16 * ib) * DI_LD + 16 * kb;
        tile16_mma<T>(tb, DI_LD, inv, XD_LD, true, tb, DI_LD, false, (T)1, lane);
      } else {
        const int jb = w - (3 - kb);         // X[kb][jb] = Ld^-1 * Y[kb][jb], in place (stored at block (jb, kb))
        T* yb = s_t + (16 * jb) * DI_LD + 16 * kb;
        tile16_mma<T>(inv, XD_LD, yb, DI_LD, false, yb, DI_LD, false, (T)1, lane);
      }
    }
    __syncthreads();
  }
  // store L (lower) and its inverse (lower; off-diagonal blocks from the upper block positions), 16-byte vectors
  {
    typedef typename Tr<T>::vec_t vec_t;
    constexpr int VE = Tr<T>::VE, VPR = 64 / VE, NV = 4096 / VE;
#pragma unroll
    for (int qq = 0; qq < NV / NT; ++qq) {
      const int idx = tid + NT * qq;
      const int row = idx / VPR, col = VE * (idx % VPR), rb = row >> 4, cb = col >> 4;
      vec_t xv = Tr<T>::vzero();
      if (cb <= rb) {
        *reinterpret_cast<vec_t*>(M + cm_off(p_pad, r0 + row, r0 + col)) = Tr<T>::lds_load(s_t + row * DI_LD + col);
#pragma unroll
        for (int e = 0; e < VE; ++e)
          xv[e] = (cb < rb) ? s_t[(16 * cb + (row & 15)) * DI_LD + 16 * rb + ((col + e) & 15)]
                            : s_x[(rb * 16 + (row & 15)) * XD_LD + ((col + e) & 15)];
      }
      *reinterpret_cast<vec_t*>(Dg + row * 64 + col) = xv;
    }
  }
  if (bad && lane == 0) atomicOr(&info[0], 1);
}

template <typename T, int NT>
__device__ __forceinline__ void factor_block64(T* __restrict__ M, int p_pad, int r0, T* __restrict__ Dg,
                                               const double* __restrict__ diag0, double piv_tol,
                                               int32_t* __restrict__ info, T* s_t, T* s_x, int tid) {
  const double tol64 = piv_tol * diag0[r0 + (tid & 63)];
  factor_block64_load<T, NT>(M, p_pad, r0, s_t, tid);
  factor_block64_core<T, NT>(M, p_pad, r0, Dg, tol64, info, s_t, s_x, tid);
}

// Factor the 128 x 128 diagonal block at (r0, r0) in place; inverses of its two 64 x 64 diagonal
// factors to Dg[0..4095] and Dg[4096..8191].  256 threads; s_a: >= 64 * DI_LD elements of LDS, s_x: >= FB_SX_ELEMS.
// Everything between the two 64 x 64 factorisations stays on chip: A21 is fetched (into the registers that
// will be the B operand) before the first one starts, L11^-1 is read where that factorisation left it in
// LDS, and A22's update  -= L21 L21^T  is applied to its LDS copy, not to memory.
template <typename T, int NT = 256>
__device__ __forceinline__ void factor_diag128(T* __restrict__ M, int p_pad, int r0, T* __restrict__ Dg,
                                               const double* __restrict__ diag0, double piv_tol,
                                               int32_t* __restrict__ info, T* s_a, T* s_x, int tid) {
  static_assert(NT == 256, "four waves");
  typedef typename Tr<T>::acc_t acc_t;
  const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, l4 = lane >> 4;
  // A21 tile of wave w: rows i = 16 w + l15, loaded as (k = 16 x + acc_row, i) = the B operand of L11^-1 A21^T
  acc_t c[4], o[4];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      c[x][r] = M[cm_off(p_pad, r0 + NB + 16 * w + l15, r0 + 16 * x + Tr<T>::acc_row(l4, r))];
  Block64Regs<T, NT> a22;   // fetched now, needed after the first factorisation: its latency is off the critical path
  factor_block64_fetch<T, NT>(a22, M, p_pad, r0 + NB, tid);
  const double tol64_b = piv_tol * diag0[r0 + NB + (tid & 63)];   // the second block's pivot thresholds, likewise
  factor_block64<T, NT>(M, p_pad, r0, Dg, diag0, piv_tol, info, s_a, s_x, tid);
  // L21^T = L11^-1 A21^T, L11^-1 block (xp, x) read from the upper block (x, xp) of s_a / the diagonal blocks in s_x
#pragma unroll
  for (int xp = 0; xp < 4; ++xp) {
    o[xp] = Tr<T>::zero();
#pragma unroll
    for (int x = 0; x <= xp; ++x)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kx = Tr<T>::acc_row(l4, r);
        const T av = (x < xp) ? s_a[(16 * x + l15) * DI_LD + 16 * xp + kx] : s_x[(16 * xp + l15) * XD_LD + kx];
        o[xp] = Tr<T>::mfma(av, c[x][r], o[xp]);
      }
  }
  __syncthreads();   // L11^-1 has been read by everyone (and stored): the region now takes L21 in operand layout
#pragma unroll
  for (int xp = 0; xp < 4; ++xp)
#pragma unroll
    for (int r = 0; r < 4; ++r) s_a[(16 * w + l15) * DI_LD + 16 * xp + Tr<T>::acc_row(l4, r)] = o[xp][r];
  __syncthreads();
  // store L21 (contiguous 16-column row pieces in the chunk-major layout)
  {
    constexpr int VE = Tr<T>::VE, VPR = 64 / VE, NV = 4096 / VE;
#pragma unroll
    for (int q = 0; q < NV / NT; ++q) {
      const int idx = tid + NT * q;
      const int row = idx / VPR, cv = idx % VPR;
      *reinterpret_cast<typename Tr<T>::vec_t*>(M + cm_off(p_pad, r0 + NB + row, r0 + VE * cv)) =
          Tr<T>::lds_load(s_a + row * DI_LD + VE * cv);
    }
  }
  // L21 L21^T, lower 16 x 16 tiles of the 4 x 4 grid: 10 tiles, wave w keeps tiles w, w + 4, w + 8 in registers
  acc_t u[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int t = w + 4 * q;
    u[q] = Tr<T>::zero();
    if (t < 10) {
      int ti, tj;
      syrk_tile(t, ti, tj);
#pragma unroll
      for (int kk = 0; kk < 16; ++kk)
        u[q] = Tr<T>::mfma(s_a[(16 * ti + l15) * DI_LD + 4 * kk + l4], s_a[(16 * tj + l15) * DI_LD + 4 * kk + l4],
                           u[q]);
    }
  }
  __syncthreads();   // L21 has been read: the region takes A22
  factor_block64_put<T, NT>(a22, s_a, tid);
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int t = w + 4 * q;
    if (t < 10) {
      int ti, tj;
      syrk_tile(t, ti, tj);
#pragma unroll
      for (int r = 0; r < 4; ++r) s_a[(16 * ti + Tr<T>::acc_row(l4, r)) * DI_LD + 16 * tj + l15] -= u[q][r];
    }
  }
  __syncthreads();
  factor_block64_core<T, NT>(M, p_pad, r0 + NB, Dg + 4096, tol64_b, info, s_a, s_x, tid);
}

template <typename T>
__global__ __launch_bounds__(256, 2) void chol_diag2_kernel(T* __restrict__ A, T* __restrict__ Dinv,
                                                            const double* __restrict__ diag0, double piv_tol,
                                                            int32_t* __restrict__ info, int p_pad, int Jo,
                                                            int nblk, int32_t* __restrict__ row_flags) {
  __shared__ __attribute__((aligned(16))) T s_a[64 * DI_LD];
  __shared__ __attribute__((aligned(16))) T s_x[FB_SX_ELEMS];
  const int mt = blockIdx.x;
  // the fused lift scan's "row p of panel J is final" word of this matrix starts the batch at zero (a memset of its own
  // between this launch and the first panel launch took 120-160 us of the lane's time in the two-lane pipeline: a fill
  // kernel queues for slots like any other)
  if (row_flags != nullptr && threadIdx.x == 0) row_flags[mt] = 0;
  T* M = A + (int64_t)mt * p_pad * p_pad;
  factor_diag128<T>(M, p_pad, Jo * 128, Dinv + ((int64_t)mt * nblk + 2 * Jo) * 4096,
                    diag0 + (int64_t)mt * p_pad, piv_tol, info, s_a, s_x, threadIdx.x);
}

// One 128 x 128 tile of panel step Jo (256 or 512 threads).  Two kinds, told apart by the workgroup-uniform `xt`:
//   L tile  (xt == 0): rows I0 = J0 + 128 + 128 tile of the matrix M itself (MJ == M); tile 0 also updates and factors
//                      the next diagonal block.
//   X tile  (xt != 0): rows I0 = 128 tile of X = V^T, which lives in its own chunk-major buffer M; the panel rows, the
//                      diagonal block's inverse (Dm) and L21 come from the ordering's TRAINING matrix MJ, the tile's
//                      start from the factored TEST matrix Bt: -B[i][j] = -L_t[J0 + j][I0 + i].
// Dm / diag0 are the slices of the matrix MJ.
// s_a: >= 2 * 128 * RK_LD elements, s_b: >= 128 * RK_LD elements of LDS.
// What an X tile needs to scan its own block of V^T for the lifts before it leaves the chip (fused lift scan, below);
// for an L tile of the last block row: the flag it raises when row p of its panel (z, or y~ in a test matrix) is final.
struct TileLift {
  int32_t* raise;          // L tile of the last block row: &flags[matrix]; else null
  const int32_t* fz;       // X tile: flag of the ordering's training matrix (z of this panel final)
  const int32_t* fy;       // X tile: flag of the ordering's test matrix (y~ of this panel final; diagonal tile only)
  double* run;             // [p_pad] running N of the ordering, carried from panel to panel
  double* P;               // [blocks][p_pad] partial sums of the ordering, one row per row block of V^T
  int p;                   // features
  int mode;                // 0: no scan (the lift kernel reads V^T back); 1: scan; 2: scan, last panel's V^T not stored
};

// (Rounds 3-4 carried eight timing-only build variants of this function -- one phase compiled out each -- and wall-clock
// stamps at its phase boundaries; the tables they produced are profiles/r04_phase_removal*.log and profiles/r04_probes.log,
// the code is in the history at the end of round 4, DESIGN_HISTORY.md says where.)
template <typename T, int NT, bool XLAST = false>
__device__ __forceinline__ void panel2_tile(T* __restrict__ M, const T* __restrict__ MJ, const T* __restrict__ Bt,
                                            T* __restrict__ Dm, const double* __restrict__ diag0,
                                            double piv_tol, int32_t* __restrict__ info, const int p_pad, const int Jo,
                                            const int tile, const int xt_arg, const int p_live, const TileLift& tl,
                                            T* const s_a, T* const s_b, const int tid) {
  const int xt = XLAST ? 1 : xt_arg;      // the last launch has X tiles only: the L-tile code drops out of its kernel
  typedef typename Tr<T>::acc_t acc_t;
  typedef typename Tr<T>::vec_t vec_t;
  constexpr int VE = Tr<T>::VE;
  constexpr int NW = NT / 64;          // waves
  constexpr int RW = 128 / NW;         // tile rows per wave: 32 / 16
  constexpr int YT = RW / 16;          // 16-row accumulator tiles per wave and panel column block: 2 / 1
  constexpr int NU = (36 + NW - 1) / NW;   // diagonal-update tiles per wave: 9 / 5
  // Region A: the two 128 x 16 operand tiles of the main loop; afterwards the 64 x 64 blocks of the
  // two-level solve and the diagonal factorisation's block.  Region B: output staging tile.
  static_assert(2 * 128 * RK_LD >= 64 * DI_LD, "a 64 x 64 block must fit region A");
  static_assert(128 * RK_LD >= FB_SX_ELEMS, "the diagonal factorisation's side buffer must fit region B");
  T* const s_rkj = s_a;
  T* const s_rki = s_a + 128 * RK_LD;
  T* const s_dinv = s_a;
  T* const s_out = s_b;

  const int lane = tid & 63, w = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int J0 = Jo * 128;
  const int I0 = xt ? tile * 128 : J0 + 128 + tile * 128;
  // Rows of a wave.  The tile's eight 16-row sub-tiles are dealt to the four waves as {w, 7 - w}: wave w owns rows
  // rb[0] = 16 w .. and rb[1] = 16 (7 - w) .. (any assignment would do: rows are independent of each other from the
  // tile's start to its store).  The pairing matters for the X tiles' first k-block, below.
  static_assert(NW == 4 && YT == 2, "the row map is written for four waves of two 16-row sub-tiles");
  const int ws = __builtin_amdgcn_readfirstlane(w);   // the wave index as the scalar it is
  const int rb[2] = {16 * ws, 16 * (7 - ws)};
  // Rows at or beyond p_live (= p + 1 rounded up to 16) are identity padding: left of the diagonal they are exact
  // zeros before, during and after every update, so the 16-row accumulator tiles that consist of them only are
  // left out of every product -- the same bits with fewer instructions (p = 1000 pads to 1024: 1 tile in 64;
  // p = 5000 to 5120: 7 in 320).  (X tiles: those rows of V^T belong to no feature; nobody reads them.)
  // All of these are wave-uniform, and the compiler has to know it (the wave index comes out of threadIdx): a
  // condition it takes for divergent turns every product it guards into an exec-masked region of its own; as
  // scalars they are plain branches (measured: 4.08 -> 4.04 ms of panel time per C3 step).  A condition-free copy
  // of the k-loop for full tiles was measured too: no faster (it costs registers the epilogue then spills).
  const bool live[2] = {I0 + rb[0] < p_live, I0 + rb[1] < p_live};

  // k-loop range: an L tile sums over every column left of the panel; an X tile starts at its own row block
  // (X[I', K] = 0 for K < I').  Inside that first block X[I', I'] is upper triangular: sub-tile s (rows 16 s ..) is zero
  // left of column 16 s, i.e. in the block's chunks c < s, and those products are left out.  With the sub-tiles dealt
  // {w, 7 - w} every wave leaves out 7 of its 16 sub-tile-chunks: the block costs every wave -- and so the workgroup,
  // whose waves meet at two barriers a chunk -- 9/16 of a full one.  (Contiguous rows per wave, as the strip kernel had
  // them, let wave 3 skip six chunks of eight while wave 0 skipped none: the workgroup took the full time.)
  const int cb = xt ? I0 / KCH : 0;
  const int nch = J0 / KCH - cb;
  // live 16-column tiles of the panel.  Only the X tiles of the LAST panel have fewer than eight (L tiles never sit in
  // the last panel: it has no rows below it): columns j >= p_live of V^T belong to the augmented row and the identity
  // padding, no lift reads them, and their accumulator tiles are left out of the k-loop.  XLAST is that launch's own
  // instantiation of the kernel (X tiles only): the guard costs the other launches nothing -- in one kernel for both, the
  // guarded copy of the loop cost the fp64 instance 288 B of scratch.
  const int xlive = XLAST ? min(8, max(1, (p_live - J0 + 15) / 16)) : 8;
  constexpr int NEVER = 0x7fffffff;
  const int cstart[2] = {live[0] ? (xt ? rb[0] / KCH : 0) : NEVER, live[1] ? (xt ? rb[1] / KCH : 0) : NEVER};
  const T* srcJ = MJ + cm_off(p_pad, J0, KCH * cb);
  const T* srcI = M + cm_off(p_pad, I0, KCH * cb);
  const int64_t chunk = (int64_t)p_pad * 16;
  // The k-loop walks the chunks from the panel's side DOWN to the tile's start: the X tiles of one ordering
  // and one panel have k-ranges that all END at the panel and start at their own row blocks; walking down, the tiles that
  // run side by side on an XCD stream the same chunk of the panel rows L[J, .] at the same time and share it through the
  // L2 -- walking up, tile I' was 8 I' chunks ahead of tile 0, further apart than the L2 holds.  (L tiles of a matrix
  // have equal ranges: in step either way.)  Panel time per C3 step, each launch alone: 5.78 -> 5.72 ms; the pipelined step
  // within the noise (6.09 both).
  RKRegs<T, 128, NT> rj = {}, ri = {};
  const int c_first = nch - 1, c_step = -1;
  if (nch > 0) {
    rk_load_full<T, 128, NT>(rj, srcJ + c_first * chunk, CM_LD, tid);
    // (the tile's own rows are read by nobody else at this time: streaming hint -- the panel rows, which the other
    // tiles of the matrix read too, keep the L2 to themselves: 6.09 -> 6.02 ms a C3 step)
    rk_load_full_nt<T, 128, NT>(ri, srcI + c_first * chunk, CM_LD, tid);
  }

  // acc[x][y][r] <-> (panel column j = 16 x + acc_row(l4, r), tile row i = RW w + 16 y + l15); holds -C^T.
  acc_t acc[8][YT];
  constexpr bool T_INIT = true;
  if (xt && T_INIT) {
    // -B[i][j] = -L_t[J0 + j][I0 + i]: sixteen lanes read sixteen consecutive columns of one row of L_t (one 128-byte
    // piece of a chunk): accumulator layout as it stands.  On the diagonal tile (I0 == J0) the entries above L_t's
    // diagonal are not part of the factor (the upper right 64 x 64 block of a diagonal block is never written):
    // unconditional loads, value selected afterwards.
#pragma unroll
    for (int x = 0; x < 8; ++x) {
#pragma unroll
      for (int y = 0; y < YT; ++y)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = J0 + 16 * x + Tr<T>::acc_row(l4, r), col = I0 + rb[y] + l15;
          const T v = __builtin_nontemporal_load(Bt + cm_off(p_pad, row, col));
          acc[x][y][r] = (row >= col) ? -v : (T)0;
        }
      if (x & 1) __builtin_amdgcn_sched_barrier(0);
    }
  } else if (T_INIT) {
    // (Round 3, measured and not kept: -A^T entering through the matrix pipe instead -- the tile's own eight chunks
    // staged like the k-loop's operands and multiplied by -I, no transposition in front of the k-loop: 4.05 against
    // 3.98 ms of panel time per C3 step in alternating processes on one box.)
    // Each wave stages its own rows through its slice of the output buffer (coalesced reads, no
    // workgroup barrier).
    constexpr int VPR = 16 / VE;        // 16-byte vectors per 16-column row piece
    constexpr int RPI = 64 / VPR;       // rows per wave instruction: 8 (fp64) / 16 (fp32)
    constexpr int NQ = RW / RPI;        // passes over the wave's rows: NQ / 2 per sub-tile
    const int rr = lane / VPR, ch = lane % VPR;
#pragma unroll
    for (int xh = 0; xh < 4; ++xh) {
      vec_t t[2][NQ];
#pragma unroll
      for (int xx = 0; xx < 2; ++xx)
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          t[xx][q] = __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(
              M + cm_off(p_pad, I0 + rb[q / (NQ / 2)] + rr + RPI * (q % (NQ / 2)), J0 + 16 * (2 * xh + xx) + VE * ch)));
#pragma unroll
      for (int xx = 0; xx < 2; ++xx) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          Tr<T>::lds_store(s_out + (rb[q / (NQ / 2)] + rr + RPI * (q % (NQ / 2))) * RK_LD + VE * ch, t[xx][q]);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int y = 0; y < YT; ++y)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            acc[2 * xh + xx][y][r] = -s_out[(rb[y] + l15) * RK_LD + Tr<T>::acc_row(l4, r)];
        __builtin_amdgcn_wave_barrier();
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  for (int it = 0, c = c_first; it < nch; ++it, c += c_step) {
    __syncthreads();
    rk_store<T, 128, NT>(rj, s_rkj, tid);
    rk_store<T, 128, NT>(ri, s_rki, tid);
    __syncthreads();
    if (it + 1 < nch) {
      rk_load_full<T, 128, NT>(rj, srcJ + (c + c_step) * chunk, CM_LD, tid);
      rk_load_full_nt<T, 128, NT>(ri, srcI + (c + c_step) * chunk, CM_LD, tid);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      T av[8], bv[YT];
#pragma unroll
      for (int x = 0; x < 8; ++x) av[x] = s_rkj[(16 * x + l15) * RK_LD + 4 * kk + l4];
#pragma unroll
      for (int y = 0; y < YT; ++y) bv[y] = s_rki[(rb[y] + l15) * RK_LD + 4 * kk + l4];
#pragma unroll
      for (int y = 0; y < YT; ++y)
        if (c >= cstart[y]) {
#pragma unroll
          for (int x = 0; x < 8; ++x)
            if (!XLAST || x < xlive) acc[x][y] = Tr<T>::mfma(av[x], bv[y], acc[x][y]);
        }
    }
  }


  // the wave's row bases once more, from a copy of the wave index the compiler cannot see through: otherwise the store
  // loop's LDS addresses are computed ahead of the k-loop and carried -- spilled -- through it
  int ws_e = ws;
  asm volatile("" : "+s"(ws_e));
  const int rb_e[2] = {16 * ws_e, 16 * (7 - ws_e)};
  // the store's and the fused lift scan's bookkeeping (both described at the store, below)
  constexpr int VPR16 = 16 / VE, RPI16 = 64 / VPR16;      // vectors per row piece; rows per wave instruction
  constexpr int SE_LD = 17;
  const bool scan = xt && tl.mode != 0;
  double* const s_e = reinterpret_cast<double*>(s_a) + (tid >> 6) * (32 * SE_LD);      // [32][SE_LD] per wave: the terms e
  double* const s_z = reinterpret_cast<double*>(s_a) + 4 * 32 * SE_LD;                  // [128] z of the panel
  static_assert((4 * 32 * SE_LD + 128) * sizeof(double) <= 2 * 128 * RK_LD * sizeof(T), "scan tiles fit region A");
  static_assert(16 * 128 * sizeof(double) <= 4 * 32 * SE_LD * sizeof(double), "the final sums fit the e tiles");
  static_assert(YT == 2, "the scan takes a wave's 32 rows as lane >> 1");
  // scan: lane -> row lane >> 1 of the wave's 32 (sub-tile (lane >> 5), its row (lane >> 1) & 15), half lane & 1 of the
  // chunk's 16 columns; column sums: lane -> column l15 of the chunk over the rows 8 l4 .. 8 l4 + 7
  const int sy = lane >> 5, si = (lane >> 1) & 15, sh = lane & 1;
  double run_c = 0.0, yt_c = 0.0, z_mine = 0.0, d_run = 0.0;
  double psum[8] = {};
  // two-level solve on the accumulators (they hold -C^T):
  //   X1^T = L11^-1 C1^T ;  -C2^T += L21 X1^T ;  X2^T = L22^-1 C2^T
  // acc[4 h + xp][y] <- -sum_{x <= xp} D[xp][x] acc[4 h + x][y], one column tile at a time (in place)
  auto tri_mult = [&](const int half) {
#pragma unroll
    for (int y = 0; y < YT; ++y) {
      if (!live[y]) continue;   // padding rows: the accumulators are and stay zero
      acc_t t[4];
#pragma unroll
      for (int xp = 0; xp < 4; ++xp) t[xp] = Tr<T>::zero();
#pragma unroll
      for (int xp = 0; xp < 4; ++xp) {
#pragma unroll
        for (int x = 0; x <= xp; ++x)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const T av = -s_dinv[(16 * xp + l15) * DI_LD + 16 * x + Tr<T>::acc_row(l4, r)];
            t[xp] = Tr<T>::mfma(av, acc[4 * half + x][y][r], t[xp]);
          }
        __builtin_amdgcn_sched_barrier(0);   // keep the scheduler from hoisting every LDS read to the top
      }
#pragma unroll
      for (int xp = 0; xp < 4; ++xp) acc[4 * half + xp][y] = t[xp];
    }
  };
  // Each 64 x 64 operand block of the solve is fetched into registers one stage ahead and put into LDS when
  // the previous stage is done with the region.
  const T* Dg = Dm + (int64_t)(2 * Jo) * 4096;
  DenseBlock64Regs<T, NT> nb;
  block64_fetch<T, NT>(nb, Dg, tid);
  __syncthreads();  // every wave is done with the operand tiles that region A now loses
  block64_put<T, NT>(nb, s_dinv, tid);
  __syncthreads();
  block64_fetch_cm<T, NT>(nb, MJ, p_pad, J0 + NB, J0, tid);
  tri_mult(0);
  __syncthreads();
  block64_put<T, NT>(nb, s_dinv, tid);
  __syncthreads();
  block64_fetch<T, NT>(nb, Dg + 4096, tid);
#pragma unroll
  for (int xp = 0; xp < 4; ++xp) {
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const T av = s_dinv[(16 * xp + l15) * DI_LD + 16 * x + Tr<T>::acc_row(l4, r)];
        // no padding test here: the accumulators of padding rows are zeros and add zeros, and a branch around every
        // product put the LDS latency of its fragment in front of it
#pragma unroll
        for (int y = 0; y < YT; ++y) acc[4 + xp][y] = Tr<T>::mfma(av, acc[x][y][r], acc[4 + xp][y]);
      }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (scan) {
    // the flag is polled, and the three vectors are fetched into registers, in front of the solve's last stage: its
    // matrix instructions hide the two memory round trips (the flag has been up for a long time by now: the training
    // matrices' L tiles are the first half of the grid's L tiles, the X tiles come after all of them)
    if (!XLAST && J0 + 128 < p_pad) {      // (the last launch has no L tiles: its row p came out of the launch before)
      if (tid == 0) {
        // The wait ends by the CLOCK, not by a spin count (round 5): half a second of the 100 MHz wall clock -- the
        // longest tile this engine can run (p = 32767: 256 row blocks of eight chunks) takes about ten milliseconds, and
        // the row comes from a workgroup that was dispatched before this one.  A flag that has not come by then never
        // will: LSSPA_INFO_SCAN_WAIT, and the batch's own sum check (LSSPA_INFO_SUM) sees what the scan then made of it.
        const long long give_up = (long long)wall_clock64() + 50000000ll;
        while (__hip_atomic_load(tl.fz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= Jo ||
               (I0 == J0 && __hip_atomic_load(tl.fy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= Jo)) {
          __builtin_amdgcn_s_sleep(32);
          if ((long long)wall_clock64() > give_up) {
            atomicOr(&info[0], 4);
            break;
          }
        }
      }
    }
  }
  __syncthreads();
  block64_put<T, NT>(nb, s_dinv, tid);
  __syncthreads();
  if (scan) {
    const int p = tl.p;
    if (tid < 128) {
      const int j = J0 + tid;
      z_mine = (j < p) ? (double)__hip_atomic_load(MJ + cm_off(p_pad, p, j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                       : 0.0;
    }
    const int c = I0 + rb_e[sy] + si;
    yt_c = (c < p) ? (double)__hip_atomic_load(Bt + cm_off(p_pad, p, c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                   : 0.0;
    run_c = (I0 == J0) ? 0.0 : tl.run[c];
    d_run = 2.0 * (yt_c - run_c);
  }
  tri_mult(1);

  // Store L[I, panel] through the output buffer, one 16-column chunk at a time (a contiguous 128 x 16
  // block in the chunk-major layout).  (Round 4, measured and not kept: the accumulators stored straight to memory --
  // an fp64 register is 32 contiguous bytes over four lanes, sixteen rows an instruction; no barrier, no LDS round
  // trip, and 6.51 against 6.20 ms a C3 step: the 32-byte pieces cost the memory system more than sixteen barriers cost
  // the tile.)
  // Every wave turns its own two 16 x 16 sub-tiles round in its own rows of the buffer: in the chunk-major layout a
  // sub-tile is one contiguous 2 KB block (fp32: 1 KB), so the wave's own lanes store it in full lines and the store
  // needs no workgroup barrier.  (As a change of its own -- workgroup-wide staging with two barriers a chunk before --
  // it left the step where it was, 6.16-6.21 against 6.19-6.21 ms: the store costs its bytes, not its barriers.)
  //
  // FUSED LIFT SCAN (X tiles, tl.mode != 0).  The lifts need, for every row c of V^T (a column of V, "test space") and
  // the columns j in ordering position, the running sum N_{j-1}[c] = sum_{k<j} z_k V^T[c][k] and the terms
  //     e = V^T[c][j] (2 (y~_c - N_{j-1}[c]) - z_j V^T[c][j]),        lift_j = z_j sum_c e / |y_test|^2
  // (k_lift.hip).  A separate kernel used to read all of V^T back for this (1.2 GB and 0.23 ms a C3 step).  Here the
  // tile does it for its own 128 x 128 block while the block passes through LDS on its way out: the 16 columns of a
  // chunk are scanned by two lanes a row (eight columns each and one exchange), the sums of e over the wave's rows go
  // through a small LDS tile, and N is carried from panel to panel in tl.run (tile (I', J) runs in launch J, tile
  // (I', J + 1) in the next).  z of THIS panel is row p of the training matrix's L tile (last block row, J) and y~ of a
  // diagonal tile's rows is row p of the test matrix's: both are written by workgroups of the SAME launch.  Those are
  // dispatched before every X tile (the grid is L tiles first, and each XCD hands out its workgroups in order), they
  // depend on nothing, and they raise a flag when their tile is stored; the X tile polls it in front of its solve's last
  // stage (above), i.e. in practice it never waits: tools/xtile_probe.hip stamps 0.04-2.2 us there, the poll's own round
  // trip (a wait that outlasts ~60 ms sets LSSPA_INFO_SCAN_WAIT and goes on: no hang).  The last launch needs no flag.
  if (scan) {
    __syncthreads();      // every wave is done with the solve's blocks in region A
    if (tid < 128) s_z[tid] = z_mine;
    __syncthreads();
  }
#pragma unroll
  for (int xp = 0; xp < 8; ++xp) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int y = 0; y < YT; ++y)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        s_out[(rb_e[y] + l15) * RK_LD + Tr<T>::acc_row(l4, r)] = acc[xp][y][r];
    __builtin_amdgcn_wave_barrier();
    if (!(XLAST && tl.mode == 2)) {
#pragma unroll
      for (int y = 0; y < YT; ++y)
#pragma unroll
        for (int h = 0; h < 16 / RPI16; ++h) {
          const int rr = rb_e[y] + RPI16 * h + lane / VPR16, cv = VE * (lane % VPR16);
          T* const dst = M + cm_off(p_pad, I0 + rr, J0 + 16 * xp + cv);
          const vec_t val = Tr<T>::lds_load(s_out + rr * RK_LD + cv);
          if (tl.raise != nullptr && I0 + rr == tl.p) {
            // row p of an L tile (z, or y~ in a test matrix): X tiles of THIS launch read it -- written through (sc1)
#pragma unroll
            for (int e = 0; e < VE; ++e)
              __hip_atomic_store(dst + e, val[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } else {
            __builtin_nontemporal_store(val, reinterpret_cast<vec_t*>(dst));   // read again a launch later at the earliest
          }
        }
    }
    if (scan) {
      const int p = tl.p;
      const int c = I0 + rb_e[sy] + si;
      // Few fp64 VECTOR instructions: beside a workgroup in its k-loop every one of them waits for the matrix pipe it
      // shares with that wave's matrix instructions -- 265 cycles an instruction beside a pure stream of them, 8.5 alone
      // (tools/mfma_cap_probe.hip).  d_run = 2 (y~ - N) is what is carried; with the lane's inclusive partial sums pi of
      // t = z v the term of column q is  e = v (d0 - (pi[q-1] + pi[q])).
      double v[8], pi[8];
      // (16-byte reads, all of them before the first select: a guarded element read becomes a branch with a full wait
      // behind it, and eight of them a chain of eight LDS round trips)
      vec_t rawv[8 / VE];
#pragma unroll
      for (int q = 0; q < 8 / VE; ++q) rawv[q] = Tr<T>::lds_load(s_out + (rb_e[sy] + si) * RK_LD + 8 * sh + VE * q);
      asm volatile("" ::: "memory");
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int jl = 16 * xp + 8 * sh + q;
        const double raw = (double)rawv[q / VE][q % VE];
        v[q] = (J0 + jl < p && c < p) ? raw : 0.0;
        const double t = s_z[jl] * v[q];
        pi[q] = q ? pi[q - 1] + t : t;
      }
      // the other half of the row (the neighbouring lane): the second half starts behind the first one's total
      const double other = __shfl_xor(pi[7], 1, 64);
      const double d0 = sh ? fma(-2.0, other, d_run) : d_run;
#pragma unroll
      for (int q = 0; q < 8; ++q)
        s_e[(lane >> 1) * SE_LD + 8 * sh + q] = v[q] * (d0 - (q ? pi[q - 1] + pi[q] : pi[0]));
      d_run = fma(-2.0, pi[7] + other, d_run);
      __builtin_amdgcn_wave_barrier();
      // column l15 of the chunk over eight of the wave's rows; the four row groups and the four waves meet at the end
      double sum = 0.0;
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) sum += s_e[(8 * l4 + rr) * SE_LD + l15];
      psum[xp] = sum;
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (scan) {
    if (sh == 0) tl.run[I0 + rb_e[sy] + si] = fma(-0.5, d_run, yt_c);      // N = y~ - d_run / 2
    __syncthreads();      // every wave is done with its e tile: the region takes the 16 partial sums of each column
    double* const s_f = reinterpret_cast<double*>(s_a);      // [4 waves x 4 row groups][128]
#pragma unroll
    for (int xp = 0; xp < 8; ++xp) s_f[((tid >> 6) * 4 + l4) * 128 + 16 * xp + l15] = psum[xp];
    __syncthreads();
    if (tid < 128) {
      double sum = 0.0;
#pragma unroll
      for (int g = 0; g < 16; ++g) sum += s_f[g * 128 + tid];
      tl.P[(int64_t)(I0 / 128) * p_pad + J0 + tid] = sum;
    }
  }
  // row p of this panel is final (z in a training matrix, y~ in a test matrix): tell the X tiles of this launch
  // (the hand-off needs no fence -- an agent-scope fence writes back or invalidates a whole L2: with one in every tile the
  // step took 7.8 instead of 6.2 ms --: the 128 values are stored and loaded with sc1 accesses, which go through to
  // memory and past the reader's L1, every storing wave waits for its stores, and one lane raises the flag behind a
  // workgroup barrier; the reader polls with sc1 loads and reads after a barrier of its own)
  if (!xt && tl.raise != nullptr && I0 + 128 == p_pad) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(tl.raise, Jo + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }

  // A[I,I] -= L[I,0:J0+128) L[I,0:J0+128)^T on the diagonal block that tile 0 owns and factors next: 36 lower
  // 16 x 16 tiles over the waves.  The accumulators of the solve are dead by now; the eight chunks just written are
  // staged once more (they come back from L2) instead of keeping both sets of accumulators alive through the
  // store loop.
  __threadfence_block();
  // The tiles of a wave -- t = ws + NW q of the row-by-row enumeration -- are COMPILE-TIME constants in four (eight)
  // copies of the loop, one per wave index: with run-time tile indices every product had its own pair of LDS reads,
  // its own full wait for them and a branch around it (a padding test per tile), i.e. the LDS latency in front of each
  // of the 36 matrix instructions of a chunk.  Here the (at most eight) row-block fragments of a k-step are fetched
  // together and the products follow back to back.  Padding row blocks (rows at or beyond p_live) are exact zeros in
  // L[I, panel]: their products add zeros, so they are no longer tested for.
  auto diag_update = [&](auto ws_tag) {
    constexpr int WS = decltype(ws_tag)::value;
    acc_t upd[NU];
#pragma unroll
    for (int q = 0; q < NU; ++q) {
      const int t = WS + NW * q;
      const int ti = syrk_ti_c(t < 36 ? t : 0), tj = syrk_tj_c(t < 36 ? t : 0);
      // start from -A[I,I] (unconditional loads: the block's upper triangle exists, its content is never stored):
      // the reads are in flight under the re-staging loop instead of being a dependent read-modify-write at the end
#pragma unroll
      for (int r = 0; r < 4; ++r)
        upd[q][r] = -M[cm_off(p_pad, I0 + 16 * ti + Tr<T>::acc_row(l4, r), I0 + 16 * tj + l15)];
    }
    // LEFT-looking, once per diagonal block (round 3): only the tile-0 workgroup comes here, for the block it is about
    // to factor, and applies the whole row panel L[I, 0 : J0 + 128) -- the chunks of the earlier panel steps and the
    // eight just written -- in one loop.  Before, every tile of every panel step read its diagonal block, subtracted its
    // own 128 columns' worth and wrote it back (28 read-modify-writes of a 128 x 128 block per matrix instead of 7, and
    // 28 short loops with their ramps instead of 7 long ones): panel launches 3.86 -> 3.72 ms per C3 step in alternating
    // processes on one box.  Measured on top of it and not kept: (a) the columns of all but the last two panel steps
    // applied one launch ahead by a workgroup of its own per matrix, to shorten this one (the launch's long pole) --
    // 3.83 against 3.72 ms: the block is then read and written twice, and the extra workgroups delay the tiles;
    // (b) tile-0 workgroups interleaved with the others in dispatch order instead of first -- 4.43 against 3.70 ms.
    const T* srcP = srcI;
    const int ndc = nch + 8;
    RKRegs<T, 128, NT> rp = {};
    __syncthreads();   // all stores above are issued and fenced
    auto products = [&](const T* buf) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        T z[8];
#pragma unroll
        for (int bk = 0; bk < 8; ++bk) z[bk] = buf[(16 * bk + l15) * RK_LD + 4 * kk + l4];
#pragma unroll
        for (int q = 0; q < NU; ++q) {
          const int t = WS + NW * q;
          if (t < 36) upd[q] = Tr<T>::mfma(z[syrk_ti_c(t)], z[syrk_tj_c(t)], upd[q]);
        }
      }
    };
    rk_load_full_nt<T, 128, NT>(rp, srcP, CM_LD, tid);      // (streaming hint, as for the tile's own rows in the k-loop)
    // (two staging buffers and one barrier per chunk instead of two: 3.76-3.77 against 3.77-3.80 ms in alternating
    // processes, within the noise, and 28 spilled registers in the fp32 instance -- not kept)
    for (int c = 0; c < ndc; ++c) {
      __syncthreads();
      rk_store<T, 128, NT>(rp, s_out, tid);
      __syncthreads();
      if (c + 1 < ndc) rk_load_full_nt<T, 128, NT>(rp, srcP + (c + 1) * chunk, CM_LD, tid);
      products(s_out);
    }
    // the block's addresses are formed again here, from a copy of p_pad the compiler cannot see through: kept from the
    // initial loads above they are carried (five of them spilled) through the loop
    int pp_e = p_pad;
    asm volatile("" : "+s"(pp_e));
#pragma unroll
    for (int q = 0; q < NU; ++q) {
      const int t = WS + NW * q;
      if (t >= 36) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * syrk_ti_c(t) + Tr<T>::acc_row(l4, r), col = 16 * syrk_tj_c(t) + l15;
        if (col <= row) M[cm_off(pp_e, I0 + row, I0 + col)] = -upd[q][r];
      }
    }
  };
  if (tile == 0 && !xt)
  switch (ws) {     // scalar: the wave index
    case 0: diag_update(std::integral_constant<int, 0>()); break;
    case 1: diag_update(std::integral_constant<int, 1>()); break;
    case 2: diag_update(std::integral_constant<int, 2>()); break;
    case 3: diag_update(std::integral_constant<int, 3>()); break;
    case 4: if constexpr (NW > 4) diag_update(std::integral_constant<int, 4>()); break;
    case 5: if constexpr (NW > 4) diag_update(std::integral_constant<int, 5>()); break;
    case 6: if constexpr (NW > 4) diag_update(std::integral_constant<int, 6>()); break;
    default: if constexpr (NW > 4) diag_update(std::integral_constant<int, 7>()); break;
  }


  // Tile 0 is the next panel's diagonal block and has just received its last update: factor it here,
  // the latency-bound sweep overlaps with the other workgroups' MFMA work.
  if (tile == 0 && !xt) {
    __threadfence_block();
    __syncthreads();
    factor_diag128<T, NT>(M, p_pad, I0, Dm + (int64_t)(2 * (Jo + 1)) * 4096, diag0, piv_tol, info, s_a, s_b, tid);
  }
}

// Workgroup of NT threads (NT / 64 waves): wave w owns tile rows RW w .. RW w + RW - 1 (RW = 128 / waves)
// for all 128 panel columns, i.e. 8 x YT accumulator tiles.
// fp32, 256 threads: capped at 168 registers for a third workgroup per CU (as the fp32 strip kernel); fp64 needs all
// 256 for its accumulators
struct Panel2Args {
  void* A;                 // [n_mats] chunk-major work matrices: training matrices first, then (tri) the test matrices
  void* Dinv;              // [n_mats][nblk][64][64]
  void* X;                 // [n_ord] chunk-major p_pad x p_pad, V^T (n_x > 0 only)
  const double* diag0;
  double piv_tol;
  int32_t* info;
  int p_pad, Jo, nblk, n_mats;
  int n_ord;               // orderings = training matrices (X tiles: the test matrix of ordering o is matrix n_ord + o)
  int n_lt;                // L tiles per matrix in this launch: p_pad / 128 - 1 - Jo (0 in the X-only last launch)
  int n_x;                 // X tiles per ordering in this launch: Jo + 1, or 0 (rect mode / strip-kernel cross-check)
  int grouped, p_live;
  int32_t* flags;          // [n_mats] fused lift scan: panels whose row p is final, per matrix (zeroed before launch 0)
  double* run;             // [n_ord][p_pad] running N of the scan
  double* Ppart;           // [n_ord][pstride] partial sums, row block I' at [I' * p_pad]
  int64_t pstride;
  int p, lift;             // features; 0: no fused scan, 1: scan, 2: scan and the last panel's V^T is not stored
  int mute;                // fault injection (developer flag 4096): the row flags are never raised
};

template <typename T, int NT, bool XLAST = false>
__global__ __launch_bounds__(NT, (sizeof(T) == 4 && NT == 256) ? 3 : NT / 128) void chol_panel2_kernel(Panel2Args a) {
  __shared__ __attribute__((aligned(16))) T s_a[2 * 128 * RK_LD];
  __shared__ __attribute__((aligned(16))) T s_b[128 * RK_LD];
  // Dispatch order (1-D grid).  The tile-0 workgroups, which also update and factor the next diagonal block, come
  // first, one per matrix.  The other L tiles follow in groups of eight matrices: consecutive ids walk the
  // eight matrices (id % 8 = matrix % 8 = XCD, workgroups go round-robin over the XCDs), then the tiles,
  // so the tiles of one matrix run at about the same time on ONE XCD and share the panel-row operand
  // L[J, 0:J0] through its L2.  The X tiles come last, KIND BY KIND: tile 0 (the longest k-loop) of every ordering,
  // then tile 1 of every ordering, ...  Workgroups of very different length must not alternate with a short period:
  // the dispatcher hands workgroup n to XCD n % 8 and, inside the XCD, to its shader engines in turn, strictly in
  // order -- a workgroup whose engine has no free slot holds up everything behind it.  With the eight X tiles of an
  // ordering interleaved (period 64 = 8 XCDs x 8 kinds) each engine saw two kinds only, and the whole chip waited
  // for the engine that held the long ones: 848 us for launch 7, its slots idle 40 % of the time
  // (tools/xtile_probe.hip: tiles that ran 25 us were not replaced until 120 us).
  const int n_mats = a.n_mats, n_ord = a.n_ord;
  int mt, tile, xt = 0;
  {
    const int id = blockIdx.x;
    const int n0 = a.n_lt > 0 ? n_mats : 0;
    const int lt1 = a.n_lt > 0 ? a.n_lt - 1 : 0;      // L tiles of a matrix after its tile 0
    const int n_l = n0 + n_mats * lt1;                // all L tiles
    if (id < n0) {
      mt = id;
      tile = 0;
    } else if (id < n_l) {
      const int rem = id - n0;
      if (a.grouped) {
        const int per = 8 * lt1;
        const int g = rem / per, within = rem - g * per;
        tile = 1 + (within >> 3);
        mt = 8 * g + (within & 7);
      } else {
        tile = 1 + rem / n_mats;
        mt = rem % n_mats;
      }
    } else {            // X tile `tile` of ordering mt (= training matrix mt)
      const int rem = id - n_l;
      tile = rem / n_ord;
      mt = rem - tile * n_ord;
      xt = 1;
    }
  }
  T* const A = static_cast<T*>(a.A);
  const int64_t pp2 = (int64_t)a.p_pad * a.p_pad;
  T* const MJ = A + (int64_t)mt * pp2;
  T* const Dm = static_cast<T*>(a.Dinv) + (int64_t)mt * a.nblk * 4096;
  const double* const d0 = a.diag0 + (int64_t)mt * a.p_pad;
  TileLift tl;
  tl.p = a.p;
  tl.mode = a.lift;
  if (XLAST || xt) {
    tl.raise = nullptr;
    tl.fz = a.flags + mt;
    tl.fy = a.flags + n_ord + mt;
    tl.run = a.run + (int64_t)mt * a.p_pad;
    tl.P = a.Ppart + (int64_t)mt * a.pstride;
    panel2_tile<T, NT, XLAST>(static_cast<T*>(a.X) + (int64_t)mt * pp2, MJ, A + (int64_t)(n_ord + mt) * pp2, Dm, d0,
                              a.piv_tol, a.info, a.p_pad, a.Jo, tile, 1, a.p_live, tl, s_a, s_b, threadIdx.x);
  } else {
    tl.raise = (a.lift && !a.mute) ? a.flags + mt : nullptr;
    tl.fz = tl.fy = nullptr;
    tl.run = tl.P = nullptr;
    panel2_tile<T, NT, false>(MJ, MJ, MJ, Dm, d0, a.piv_tol, a.info, a.p_pad, a.Jo, tile, 0, a.p_live, tl, s_a, s_b,
                              threadIdx.x);
  }
}

// whole factorisation of n_mats matrices: one diagonal launch + (p_pad / 128 - 1) panel launches (+ one more, X tiles
// only, when V^T is computed alongside)
hipError_t launch_chol2_diag(void* A, void* Dinv, const double* diag0, double piv_tol, int32_t* info, int p_pad,
                             int n_mats, int f32, hipStream_t st, int32_t* row_flags) {
  if (p_pad % 128 != 0 || n_mats < 1) return hipErrorInvalidValue;
  const int nblk = p_pad / NB;
  if (f32)
    hipLaunchKernelGGL(chol_diag2_kernel<float>, dim3(n_mats), dim3(256), 0, st, (float*)A, (float*)Dinv, diag0,
                       piv_tol, info, p_pad, 0, nblk, row_flags);
  else
    hipLaunchKernelGGL(chol_diag2_kernel<double>, dim3(n_mats), dim3(256), 0, st, (double*)A, (double*)Dinv,
                       diag0, piv_tol, info, p_pad, 0, nblk, row_flags);
  return hipGetLastError();
}

hipError_t launch_chol2_panel(void* A, void* Dinv, const double* diag0, double piv_tol, int32_t* info, int p_pad,
                              int Jo, int n_mats, int f32, hipStream_t st, int flags, int p_live, void* X,
                              int n_ord, const PanelLift* pl) {
  if (p_live <= 0 || p_live > p_pad) p_live = p_pad;
  const int n_panel = p_pad / 128 - 1;
  // with X tiles (X != null) the matrices are [n_ord training][n_ord test] and there is one more launch, Jo = n_panel
  if (p_pad % 128 != 0 || Jo < 0 || n_mats < 1 || Jo > n_panel || (Jo == n_panel && !X)) return hipErrorInvalidValue;
  if (X && (n_ord < 1 || n_mats != 2 * n_ord)) return hipErrorInvalidValue;
  Panel2Args a;
  a.A = A;
  a.Dinv = Dinv;
  a.X = X;
  a.diag0 = diag0;
  a.piv_tol = piv_tol;
  a.info = info;
  a.p_pad = p_pad;
  a.Jo = Jo;
  a.nblk = p_pad / NB;
  a.n_mats = n_mats;
  a.n_ord = X ? n_ord : n_mats;
  a.n_lt = n_panel - Jo;
  a.n_x = X ? Jo + 1 : 0;
  a.p_live = p_live;
  a.flags = nullptr;
  a.run = a.Ppart = nullptr;
  a.pstride = 0;
  a.p = 0;
  a.lift = 0;
  if (pl && pl->mode) {
    if (!X || !pl->flags || !pl->run || !pl->Ppart || pl->p < 1 || pl->p >= p_pad ||
        pl->pstride < (int64_t)(p_pad / 128) * p_pad)
      return hipErrorInvalidValue;
    a.flags = pl->flags;
    a.run = pl->run;
    a.Ppart = pl->Ppart;
    a.pstride = pl->pstride;
    a.p = pl->p;
    a.lift = pl->mode;
  }
  a.mute = (flags & 4096) ? 1 : 0;
  const int64_t total = (int64_t)n_mats * a.n_lt + (int64_t)a.n_ord * a.n_x;
  if (total < 1 || total > 0x7fffffff) return hipErrorInvalidValue;
  a.grouped = (n_mats % 8 == 0 && a.n_lt > 1) ? 1 : 0;
  const dim3 grid((unsigned)total);
  // 256 threads: 512-thread workgroups (16 rows per wave, twice the waves per SIMD) were measured
  // slower in both precisions -- the epilogue is bound by its memory traffic, not by latency
  const bool xlast = a.n_lt == 0 && p_live < p_pad - 15;     // X tiles only, and dead columns in the last panel
  if (f32) {
    if (xlast) hipLaunchKernelGGL((chol_panel2_kernel<float, 256, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((chol_panel2_kernel<float, 256, false>), grid, dim3(256), 0, st, a);
  } else {
    if (xlast) hipLaunchKernelGGL((chol_panel2_kernel<double, 256, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((chol_panel2_kernel<double, 256, false>), grid, dim3(256), 0, st, a);
  }
  return hipGetLastError();
}

// =====================================================================================
// strip2:  V = L^-1 * RHS  by 128-column strips of the right-hand side, advancing 128 ROWS per step, top to bottom:
//     V[I] = L_II^-1 ( RHS[I] - sum_{K<I} L[I,K] V[K] ).
//   rect : RHS = rows perm[i] of Ft (p x m, fp64), a plain row gather -- the shipped path for M < p (the test Gram
//          matrix is singular there, no test Cholesky).
//   tri  : RHS = L_t (Cholesky factor of the permuted test Gram), lower triangular, so a strip starts at its own
//          diagonal block.  Since round 4 tri mode computes V^T inside the panel launches (X tiles above); this
//          kernel is then the independent cross-check (developer flag 128).
// Strips are independent: no inter-workgroup traffic.  The k-loop of a step covers both 64-row halves with one pass
// over the V rows above them (16 flop per operand byte).  The step ends with a two-level triangular solve on the
// accumulators:
//     X1 = Dinv_i C1 ;  C2 -= L[i+1][i] X1 ;  X2 = Dinv_{i+1} C2
// in which X1, still in registers, is the B operand of the middle product.
// =====================================================================================
// NT threads = NT / 64 waves, each owning 32 columns: the strip is CW = NT / 2 columns wide.  The wider
// strip (NT = 512) reads L half as often; the accumulators per wave, and so the waves per SIMD, are the same.
// fp32, 256 threads: capped at 168 registers (172 otherwise) for a third workgroup per CU -- 105.9 -> 101.6 ms at the
// C5 shape; fp64 needs all 256 registers for its accumulators and stays at two.
template <typename T, int NT, int OCC = (sizeof(T) == 4 && NT == 256) ? 3 : 2>
__global__ __launch_bounds__(NT, OCC) void strip2_kernel(StripArgs a) {
  constexpr int CW = NT / 2;
  typedef KCWRegs<T, CW, NT> KR;
  typedef typename Tr<T>::acc_t acc_t;
  __shared__ __attribute__((aligned(16))) T s_rk[128 * RK_LD];
  __shared__ __attribute__((aligned(16))) T s_kc[16 * KR::LD];
  __shared__ __attribute__((aligned(16))) T s_dinv[64 * DI_LD];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int ws = __builtin_amdgcn_readfirstlane(w);   // the wave index as the scalar it is
  const int ord = blockIdx.x;
  const int c0 = blockIdx.y * CW;
  const int cols_valid = min(CW, a.m_pad - c0);   // the last strip of a wide layout may be half empty
  const bool wactive = 32 * (threadIdx.x >> 6) < cols_valid;
  const int p = a.p, p_pad = a.p_pad, m_pad = a.m_pad;
  const int nblk = p_pad / NB;
  const int n_iblk = (p + NB - 1) / NB;
  const int64_t ldv = ldv_of(m_pad);
  const int64_t chunk = (int64_t)p_pad * 16;
  const T* L = static_cast<const T*>(a.A) + (int64_t)ord * p_pad * p_pad;   // chunk-major
  const T* Lt = a.tri ? static_cast<const T*>(a.rhs) + (int64_t)ord * p_pad * p_pad : nullptr;
  const int32_t* perm = a.tri ? nullptr : a.perms + (int64_t)ord * p;
  T* V = static_cast<T*>(a.V) + (int64_t)ord * v_rows_of(p) * ldv;
  const T* Dv = static_cast<const T*>(a.Dinv) + (int64_t)ord * nblk * 4096;

  const int ib0 = a.tri ? c0 / NB : 0;       // even: strips are a multiple of 128 wide
  const int kstart = a.tri ? c0 : 0;
  // tri: the rows above the strip's first diagonal block are structurally zero and nobody reads them -- this
  // kernel's k-loops start at row c0 and lift_partial skips a 64-column strip's rows above its first column -- so
  // they are not written either (they were: 0.9 GB of zeros per 256 orderings at p = 1000); lsspa_debug_factor
  // masks them on the host.

  for (int ib = ib0; ib < n_iblk; ib += 2) {
    const int I0 = ib * NB;
    const bool two = ib + 1 < n_iblk;          // the last step of an odd block count has one half
    acc_t acc[8][2];
#pragma unroll
    for (int x = 0; x < 8; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y) acc[x][y] = Tr<T>::zero();
    // padding: rows at or beyond row_live (identity rows of L: zeros left of the diagonal) and columns at or beyond
    // col_live (zero columns of the right-hand side, hence of V) only ever add exact zeros in the k-loop: the 16 x 16
    // tiles made of them are left out of it
    const int xlive = min(8, max(0, (a.row_live - I0 + 15) / 16));
    const int ylive = min(2, max(0, (a.col_live - (c0 + 32 * ws) + 15) / 16));   // wave-uniform, and known to be

    const T* srcL = L + cm_off(p_pad, I0, kstart);
    const T* srcV = V + kstart * ldv + c0;
    const int nch = (I0 - kstart) / KCH;
    RKRegs<T, 128, NT> rl = {};
    KR rv = {};
    if (nch > 0) {
      if (two) rk_load_full<T, 128, NT>(rl, srcL, CM_LD, tid);
      else rk_load<T, 128, NT>(rl, srcL, CM_LD, tid, 64);
      kcw_load<T, CW, NT>(rv, srcV, ldv, tid, cols_valid);
    }
    for (int c = 0; c < nch; ++c) {
      __syncthreads();
      rk_store<T, 128, NT>(rl, s_rk, tid);
      kcw_store<T, CW, NT>(rv, s_kc, tid, cols_valid);
      __syncthreads();
      if (c + 1 < nch) {
        if (two) rk_load_full<T, 128, NT>(rl, srcL + (c + 1) * chunk, CM_LD, tid);
        else rk_load<T, 128, NT>(rl, srcL + (c + 1) * chunk, CM_LD, tid, 64);
        kcw_load<T, CW, NT>(rv, srcV + (c + 1) * KCH * ldv, ldv, tid, cols_valid);
      }
      // tri: V[k][col] = 0 for col > k: wave w (columns c0 + 32 w ..) sees only zeros while k < c0 + 32 w
      if (a.tri && c < 2 * ws) continue;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        T av[8], bv[2];
#pragma unroll
        for (int x = 0; x < 8; ++x) av[x] = s_rk[(16 * x + l15) * RK_LD + 4 * kk + l4];
#pragma unroll
        for (int y = 0; y < 2; ++y) bv[y] = s_kc[(4 * kk + l4) * KR::LD + 32 * w + 16 * y + l15];
#pragma unroll
        for (int x = 0; x < 8; ++x)
          if (x < xlive) {
#pragma unroll
            for (int y = 0; y < 2; ++y)
              if (y < ylive) acc[x][y] = Tr<T>::mfma(av[x], bv[y], acc[x][y]);
          }
      }
    }

    // Strides re-materialised AFTER the k-loop: otherwise every row address of the epilogue (32 of them, 64 bit)
    // is computed ahead of the k-loop and carried -- spilled -- through it.
    int64_t ldv_e = ldv;
    int pp_e = p_pad;
    asm volatile("" : "+s"(ldv_e), "+s"(pp_e));

    __syncthreads();  // s_dinv is still being read by slower waves of the previous step
    load_block64<T, NT>(s_dinv, Dv + (int64_t)ib * 4096, tid);

    // C = RHS[I] - acc  (direct global reads: 16 lanes cover one contiguous row segment), one half at a
    // time so that only 32 loads are in flight
    // Every load below is unconditional (clamped address, value selected afterwards): a guarded load
    // becomes a branch per element, and the wait at each join turns the 32 loads into a latency chain.
    auto rhs_minus_acc = [&](const int half) {
      const int blk_end = I0 + (half + 1) * NB;    // end of the rows' own diagonal block
      if (a.tri) {
#pragma unroll
        for (int xx = 0; xx < 4; ++xx) {
          const int x = 4 * half + xx;
          T t[4][2];
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int y = 0; y < 2; ++y)   // column clamped into the rows' own block: always inside the matrix
              t[r][y] = __builtin_nontemporal_load(Lt + cm_off(pp_e, I0 + 16 * x + Tr<T>::acc_row(l4, r),
                                                               min(c0 + 32 * w + 16 * y + l15, blk_end - 1)));
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int y = 0; y < 2; ++y) {
              const int c = c0 + 32 * w + 16 * y + l15;
              acc[x][y][r] = ((c < blk_end) ? t[r][y] : (T)0) - acc[x][y][r];
            }
          __builtin_amdgcn_sched_barrier(0);   // 8 loads in flight, not 32
        }
      } else {
#pragma unroll
        for (int xx = 0; xx < 4; ++xx) {
          const int x = 4 * half + xx;
          double t[4][2];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = I0 + 16 * x + Tr<T>::acc_row(l4, r);
            const int64_t srow = (int64_t)perm[min(i, p - 1)] * m_pad;
#pragma unroll
            for (int y = 0; y < 2; ++y) t[r][y] = a.Ft[srow + (wactive ? c0 + 32 * w + 16 * y + l15 : 0)];
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = I0 + 16 * x + Tr<T>::acc_row(l4, r);
#pragma unroll
            for (int y = 0; y < 2; ++y) acc[x][y][r] = ((i < p) ? (T)t[r][y] : (T)0) - acc[x][y][r];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    // acc[4 h + xp][y] <- sum_{x <= xp} D[xp][x] acc[4 h + x][y], one column tile at a time (in place)
    auto tri_solve = [&](const int half) {
#pragma unroll
      for (int y = 0; y < 2; ++y) {
        acc_t t[4];
#pragma unroll
        for (int xp = 0; xp < 4; ++xp) t[xp] = Tr<T>::zero();
#pragma unroll
        for (int xp = 0; xp < 4; ++xp)
#pragma unroll
          for (int x = 0; x <= xp; ++x)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const T av = s_dinv[(16 * xp + l15) * DI_LD + 16 * x + Tr<T>::acc_row(l4, r)];
              t[xp] = Tr<T>::mfma(av, acc[4 * half + x][y][r], t[xp]);
            }
        __builtin_amdgcn_sched_barrier(0);   // keep the scheduler from hoisting every LDS read to the top
#pragma unroll
        for (int xp = 0; xp < 4; ++xp) acc[4 * half + xp][y] = t[xp];
      }
    };
    auto store_half = [&](const int half) {
#pragma unroll
      for (int xp = 0; xp < 4; ++xp)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = I0 + half * NB + 16 * xp + Tr<T>::acc_row(l4, r);
#pragma unroll
          for (int y = 0; y < 2; ++y)
            if (wactive) V[i * ldv_e + c0 + 32 * w + 16 * y + l15] = acc[4 * half + xp][y][r];
        }
    };

    rhs_minus_acc(0);
    __syncthreads();
    tri_solve(0);     // X1 = Dinv_i * C1
    store_half(0);
    __builtin_amdgcn_sched_barrier(0);

    if (two) {
      rhs_minus_acc(1);
      // C2 -= L[i+1][i] * X1  (X1 straight from its accumulators)
      __syncthreads();
      load_block64_cm<T, NT>(s_dinv, L, p_pad, I0 + NB, I0, tid, true);
      __syncthreads();
#pragma unroll
      for (int xp = 0; xp < 4; ++xp) {
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const T av = s_dinv[(16 * xp + l15) * DI_LD + 16 * x + Tr<T>::acc_row(l4, r)];
#pragma unroll
            for (int y = 0; y < 2; ++y) acc[4 + xp][y] = Tr<T>::mfma(av, acc[x][y][r], acc[4 + xp][y]);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
      load_block64<T, NT>(s_dinv, Dv + (int64_t)(ib + 1) * 4096, tid);
      __syncthreads();
      tri_solve(1);   // X2 = Dinv_{i+1} * C2
      store_half(1);
    }
    // the next step's k-loop reads these rows back (written by other waves of this workgroup)
    __threadfence_block();
    __syncthreads();
  }
}

hipError_t launch_strip(const StripArgs& a_in, hipStream_t st) {
  StripArgs a = a_in;
  if (a.row_live <= 0 || a.row_live > a.p_pad) a.row_live = a.p_pad;
  if (a.col_live <= 0 || a.col_live > a.m_pad) a.col_live = a.m_pad;
  if (a.p < 1 || a.p_pad % NB != 0 || a.m_pad % 128 != 0 || a.n_ord < 1) return hipErrorInvalidValue;
  if (a.tri && a.m_pad > a.p_pad + 127) return hipErrorInvalidValue;
  if (a.tri ? (a.rhs == nullptr) : (a.perms == nullptr || a.Ft == nullptr)) return hipErrorInvalidValue;
  dim3 grid(a.n_ord, a.m_pad / 128);
  // 128-column strips, 256 threads (256-column strips with 512 threads: a third less L traffic, measured 5 % slower at
  // p = 1000 and 3 % faster at p = 5000 in fp32, rounds 1-2; dropped with the other unshipped variants in round 4)
  if (a.f32)
    hipLaunchKernelGGL((strip2_kernel<float, 256>), grid, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((strip2_kernel<double, 256>), grid, dim3(256), 0, st, a);
  return hipGetLastError();
}

// =====================================================================================
// probe: one MFMA through the traits, used by the unit test that pins the operand / result lane
// maps of v_mfma_f64_16x16x4_f64 and v_mfma_f32_16x16x4_f32
// =====================================================================================
template <typename T>
__global__ void mfma_probe_kernel(const double* A, const double* B, double* D) {
  const int l = threadIdx.x, l15 = l & 15, l4 = l >> 4;
  typename Tr<T>::acc_t acc = Tr<T>::zero();
  acc = Tr<T>::mfma((T)A[l15 * 4 + l4], (T)B[l4 * 16 + l15], acc);
#pragma unroll
  for (int r = 0; r < 4; ++r) D[Tr<T>::acc_row(l4, r) * 16 + l15] = (double)acc[r];
}

hipError_t launch_mfma_probe(const double* A, const double* B, double* D, int f32, hipStream_t st) {
  if (f32)
    hipLaunchKernelGGL(mfma_probe_kernel<float>, dim3(1), dim3(64), 0, st, A, B, D);
  else
    hipLaunchKernelGGL(mfma_probe_kernel<double>, dim3(1), dim3(64), 0, st, A, B, D);
  return hipGetLastError();
}

}  // namespace lsspa
