// Tile-level building blocks shared by the LS-SPA kernels (gfx950 / CDNA4 only).
//
// All dense inner products run on the matrix pipe, one wave per 16x16 tile with K = 4 per
// instruction: v_mfma_f64_16x16x4_f64 (fp64 path) or v_mfma_f32_16x16x4_f32 (fp32 path).
//   A operand : one element per lane, lane l holds A[i = l & 15][k = l >> 4]
//   B operand : one element per lane, lane l holds B[k = l >> 4][j = l & 15]
//   C/D       : four elements per lane, register r of lane l holds
//                 f64:  D[row = (l >> 4) + 4 r][col = l & 15]
//                 f32:  D[row = 4 (l >> 4) + r][col = l & 15]
// (cdna_hip_programming.md section 3; the f64 map differs from every other dtype).  Both maps are
// wrapped in Tr<T>::acc_row, and every use below goes through it.
//
// Consequence used throughout: register r of an accumulator tile is, as it stands, the B operand
// of one k-step of a following product that sums over the tile's ROW index -- the step whose k
// values are acc_row(l >> 4, r).  The triangular solves "X = Dinv * C" therefore take C straight
// from the accumulators, with no LDS round trip; the A operand is read at the matching k.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace lsspa {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));  // 16-byte staging unit, fp64
typedef float f4 __attribute__((ext_vector_type(4)));    // 16-byte staging unit / accumulator, fp32
typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int NB = 64;        // factorisation block (diagonal blocks are NB x NB)
constexpr int KCH = 16;       // k-chunk staged per main-loop iteration
// LDS row strides, in ELEMENTS, the same numbers for both element sizes:
constexpr int RK_LD = 18;     // [rows][16 k] tile: conflict-free for the (row = l&15, k = l>>4) fragment
                              // read (fp64: 144-B rows, 16-B aligned; fp32: 72-B rows, 8-B aligned)
constexpr int KC_LD = 144;    // [16 k][128 cols] tile: +16 puts the two k rows of a 32-lane group on
                              // disjoint bank halves
constexpr int DI_LD = 66;     // 64 x 64 block read as an A operand
constexpr int TT_LD = 65;     // small reduction tiles walked by column

// Layout of the per-ordering work matrices in HBM: CHUNK-MAJOR.  A p_pad x p_pad matrix is stored as
// p_pad/16 column chunks of 16 columns; inside a chunk the rows follow each other:
//     element (r, c)  ->  ((c >> 4) * p_pad + r) * 16 + (c & 15)
// Every operand tile the factorisation kernels stage -- R rows x 16 k -- is then ONE contiguous block
// instead of R separate row segments a full row apart.  The tile loaders below take it as a row-major
// tile with row stride CM_LD = 16.
constexpr int CM_LD = 16;
__host__ __device__ inline int64_t cm_off(int p_pad, int r, int c) {
  return ((int64_t)(c >> 4) * p_pad + r) * 16 + (c & 15);
}
// V (solve result) stays row-major: its tiles are 16 rows x 128 columns.  32 extra elements per row
// keep a tile's rows from landing on the same few memory channels (m_pad is a multiple of 128).
constexpr int LD_PAD = 32;
__host__ __device__ inline int64_t ldv_of(int m_pad) { return (int64_t)m_pad + LD_PAD; }
// rows of a V matrix: the ordering's row blocks rounded up to 128
__host__ __device__ inline int64_t v_rows_of(int p) { return (int64_t)((p + 127) / 128) * 128; }

// Note on v_mfma_f64_4x4x4_4b_f64: in a dependent-free register loop it issues every 16.5 cycles (tools/mfma_bench3.hip)
// and a 16x16x4 step can be built from four of them (lane maps probed in tools/mfma_probe4.hip: lane l = 16 q + 4 g + t
// holds A_g[i=t][k=q], B_g[k=q][j=t], D_g[i=q][j=t]; CBSZ/ABID broadcast is ignored for f64).  It buys nothing in the
// kernels: in the isolated k-loop (tools/mfma_bench5.hip) both forms sustain the same rate, and kernel variants written
// with the 4x4x4 form (git history: "Experimental 4x4x4-MFMA kernel variants") were correct and slower.
// What this loop shape (128 x 128 tile, 16-wide k-chunks through LDS, 2 workgroups per CU) delivers on random operands
// at the steady-state clock (tools/mfma_bench6.hip, profiles/r02_kloop_ceiling.log): ~60 TFLOP/s fed from HBM, ~67 from
// the Infinity Cache or L2 -- the vendor library's own 128 x 128 x 16 GEMM kernel reaches 69-71 at 8192^3 and 59-67 batched
// at 1024^3.  Variations of the loop (fragment prefetch, double-buffered LDS, one operand straight to registers, deeper
// global prefetch) measured equal or slower: the loop is not what holds the factorisation kernels at 46-50.

// ---- per-element-type traits ---------------------------------------------------------------------
template <typename T>
struct Tr;

template <>
struct Tr<double> {
  typedef d4 acc_t;
  typedef v2d vec_t;
  static constexpr int VE = 2;  // elements per 16 bytes
  static __device__ __forceinline__ acc_t mfma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int acc_row(int l4, int r) { return l4 + 4 * r; }
  static __device__ __forceinline__ acc_t zero() {
    acc_t z = {0.0, 0.0, 0.0, 0.0};
    return z;
  }
  static __device__ __forceinline__ vec_t vzero() {
    vec_t z = {0.0, 0.0};
    return z;
  }
  // LDS addresses of staged vectors are 16-byte aligned for fp64 (all strides are even)
  static __device__ __forceinline__ void lds_store(double* p, vec_t v) { *reinterpret_cast<vec_t*>(p) = v; }
  static __device__ __forceinline__ vec_t lds_load(const double* p) { return *reinterpret_cast<const vec_t*>(p); }
};

template <>
struct Tr<float> {
  typedef f4 acc_t;
  typedef f4 vec_t;
  static constexpr int VE = 4;
  static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int acc_row(int l4, int r) { return 4 * l4 + r; }
  static __device__ __forceinline__ acc_t zero() {
    acc_t z = {0.f, 0.f, 0.f, 0.f};
    return z;
  }
  static __device__ __forceinline__ vec_t vzero() { return zero(); }
  // fp32 tile rows are 72 B / 264 B apart: staged vectors are only 8-byte aligned in LDS
  static __device__ __forceinline__ void lds_store(float* p, vec_t v) {
    f2 lo = {v[0], v[1]}, hi = {v[2], v[3]};
    *reinterpret_cast<f2*>(p) = lo;
    *reinterpret_cast<f2*>(p + 2) = hi;
  }
  static __device__ __forceinline__ vec_t lds_load(const float* p) {
    const f2 lo = *reinterpret_cast<const f2*>(p), hi = *reinterpret_cast<const f2*>(p + 2);
    vec_t v = {lo[0], lo[1], hi[0], hi[1]};
    return v;
  }
};

// ---- register staging of a [R rows][16 k] tile (R = 64 or 128), NT threads ---------------------
template <typename T, int R, int NT = 256>
struct RKRegs {
  static constexpr int VPR = 16 / Tr<T>::VE;   // 16-byte vectors per row: 8 (fp64) / 4 (fp32)
  static constexpr int RPP = NT / VPR;         // rows per pass of the workgroup: 32 / 64 at 256 threads
  static constexpr int NP = R / RPP;           // passes
  static_assert(NP >= 1 && NP * RPP == R, "tile rows must be a multiple of the rows per pass");
  typename Tr<T>::vec_t v[NP];
};

template <typename T, int R, int NT = 256>
__device__ __forceinline__ void rk_load(RKRegs<T, R, NT>& r, const T* __restrict__ src, int64_t ld, int tid,
                                        int rows_valid) {
  typedef RKRegs<T, R, NT> RR;
  const int c = tid % RR::VPR, row = tid / RR::VPR;
#pragma unroll
  for (int q = 0; q < RR::NP; ++q) {
    const int rr = row + RR::RPP * q;
    if (rr < rows_valid)
      r.v[q] = *reinterpret_cast<const typename Tr<T>::vec_t*>(src + (int64_t)rr * ld + Tr<T>::VE * c);
    else
      r.v[q] = Tr<T>::vzero();
  }
}

// the same for a tile whose R rows all exist: no per-row guard (the guard compiles to an exec-masked branch
// around every load even when the bound is the literal tile height)
template <typename T, int R, int NT = 256>
__device__ __forceinline__ void rk_load_full(RKRegs<T, R, NT>& r, const T* __restrict__ src, int64_t ld, int tid) {
  typedef RKRegs<T, R, NT> RR;
  const int c = tid % RR::VPR, row = tid / RR::VPR;
#pragma unroll
  for (int q = 0; q < RR::NP; ++q)
    r.v[q] = *reinterpret_cast<const typename Tr<T>::vec_t*>(src + (int64_t)(row + RR::RPP * q) * ld + Tr<T>::VE * c);
}

// the same with the streaming hint (an operand no other workgroup reads at this time)
template <typename T, int R, int NT = 256>
__device__ __forceinline__ void rk_load_full_nt(RKRegs<T, R, NT>& r, const T* __restrict__ src, int64_t ld, int tid) {
  typedef RKRegs<T, R, NT> RR;
  const int c = tid % RR::VPR, row = tid / RR::VPR;
#pragma unroll
  for (int q = 0; q < RR::NP; ++q)
    r.v[q] = __builtin_nontemporal_load(
        reinterpret_cast<const typename Tr<T>::vec_t*>(src + (int64_t)(row + RR::RPP * q) * ld + Tr<T>::VE * c));
}

template <typename T, int R, int NT = 256>
__device__ __forceinline__ void rk_store(const RKRegs<T, R, NT>& r, T* lds, int tid) {
  typedef RKRegs<T, R, NT> RR;
  const int c = tid % RR::VPR, row = tid / RR::VPR;
#pragma unroll
  for (int q = 0; q < RR::NP; ++q) Tr<T>::lds_store(lds + (row + RR::RPP * q) * RK_LD + Tr<T>::VE * c, r.v[q]);
}

// ---- register staging of a [16 k][128 cols] tile, 256 threads -----------------------------------
template <typename T>
struct KCRegs {
  static constexpr int VPR = 128 / Tr<T>::VE;  // vectors per k row: 64 / 32
  static constexpr int RPP = 256 / VPR;        // k rows per pass: 4 / 8
  static constexpr int NP = 16 / RPP;          // passes: 4 / 2
  typename Tr<T>::vec_t v[NP];
};

template <typename T>
__device__ __forceinline__ void kc_load(KCRegs<T>& r, const T* __restrict__ src, int64_t ld, int tid) {
  typedef KCRegs<T> KR;
  const int c = tid % KR::VPR, k = tid / KR::VPR;
#pragma unroll
  for (int q = 0; q < KR::NP; ++q)
    r.v[q] = *reinterpret_cast<const typename Tr<T>::vec_t*>(src + (int64_t)(k + KR::RPP * q) * ld + Tr<T>::VE * c);
}

template <typename T>
__device__ __forceinline__ void kc_store(const KCRegs<T>& r, T* lds, int tid) {
  typedef KCRegs<T> KR;
  const int c = tid % KR::VPR, k = tid / KR::VPR;
#pragma unroll
  for (int q = 0; q < KR::NP; ++q) Tr<T>::lds_store(lds + (k + KR::RPP * q) * KC_LD + Tr<T>::VE * c, r.v[q]);
}

// ---- the same for a [16 k][CW cols] tile and NT threads (row stride CW + 16 in LDS).  Columns at or
// beyond cols_valid are not read: the lane loads column 0 instead (an address select, no branch -- a
// branch would put a wait for the loads at its join and serialise the prefetch) and stores zeros. ------
template <typename T, int CW, int NT>
struct KCWRegs {
  static constexpr int VPR = CW / Tr<T>::VE;   // vectors per k row
  static constexpr int RPP = NT / VPR;         // k rows per pass
  static constexpr int NP = 16 / RPP;          // passes
  static constexpr int LD = CW + 16;
  static_assert(RPP >= 1 && NP >= 1 && NP * RPP == 16, "tile shape does not divide over the workgroup");
  typename Tr<T>::vec_t v[NP];
};

template <typename T, int CW, int NT>
__device__ __forceinline__ void kcw_load(KCWRegs<T, CW, NT>& r, const T* __restrict__ src, int64_t ld, int tid,
                                         int cols_valid) {
  typedef KCWRegs<T, CW, NT> KR;
  const int c = tid % KR::VPR, k = tid / KR::VPR;
  const int cc = (Tr<T>::VE * c < cols_valid) ? Tr<T>::VE * c : 0;
  // one 32-bit per-thread offset for all passes; the pass's row step is uniform and goes into the scalar base --
  // the 64-bit address of every pass kept per thread was four register pairs carried (spilled) through the kernel
  const unsigned toff = (unsigned)(k * (int)ld + cc);
#pragma unroll
  for (int q = 0; q < KR::NP; ++q) {
    const T* row = src + (int64_t)(KR::RPP * q) * ld;
    r.v[q] = *reinterpret_cast<const typename Tr<T>::vec_t*>(row + toff);
  }
}

template <typename T, int CW, int NT>
__device__ __forceinline__ void kcw_store(const KCWRegs<T, CW, NT>& r, T* lds, int tid, int cols_valid) {
  typedef KCWRegs<T, CW, NT> KR;
  const int c = tid % KR::VPR, k = tid / KR::VPR;
  const bool ok = Tr<T>::VE * c < cols_valid;
#pragma unroll
  for (int q = 0; q < KR::NP; ++q)
    Tr<T>::lds_store(lds + (k + KR::RPP * q) * KR::LD + Tr<T>::VE * c, ok ? r.v[q] : Tr<T>::vzero());
}

// copy a dense 64 x 64 block (row-major, ld 64) from global into LDS with stride DI_LD, NT threads
template <typename T, int NT = 256>
__device__ __forceinline__ void load_block64(T* lds, const T* __restrict__ g, int tid) {
  constexpr int VE = Tr<T>::VE, VPR = 64 / VE, NV = 4096 / VE;
#pragma unroll
  for (int q = 0; q < NV / NT; ++q) {
    const int idx = tid + NT * q;
    const int row = idx / VPR, cv = idx % VPR;
    Tr<T>::lds_store(lds + row * DI_LD + VE * cv,
                     *reinterpret_cast<const typename Tr<T>::vec_t*>(g + row * 64 + VE * cv));
  }
}

// the same in two moves -- fetch into registers, put into LDS later -- so that the global latency of a block
// needed by the next stage hides behind the current stage's arithmetic
template <typename T, int NT = 256>
struct DenseBlock64Regs {
  typename Tr<T>::vec_t v[4096 / Tr<T>::VE / NT];
};

template <typename T, int NT = 256>
__device__ __forceinline__ void block64_fetch(DenseBlock64Regs<T, NT>& b, const T* __restrict__ g, int tid) {
  constexpr int VE = Tr<T>::VE, VPR = 64 / VE, NV = 4096 / VE;
#pragma unroll
  for (int q = 0; q < NV / NT; ++q) {
    const int idx = tid + NT * q;
    b.v[q] = *reinterpret_cast<const typename Tr<T>::vec_t*>(g + (idx / VPR) * 64 + VE * (idx % VPR));
  }
}

// source: the 64 x 64 block at (r0, c0) of a chunk-major matrix
template <typename T, int NT = 256>
__device__ __forceinline__ void block64_fetch_cm(DenseBlock64Regs<T, NT>& b, const T* __restrict__ A, int p_pad, int r0,
                                                 int c0, int tid) {
  constexpr int VE = Tr<T>::VE, VPR = 64 / VE, NV = 4096 / VE;
#pragma unroll
  for (int q = 0; q < NV / NT; ++q) {
    const int idx = tid + NT * q;
    b.v[q] = *reinterpret_cast<const typename Tr<T>::vec_t*>(A + cm_off(p_pad, r0 + idx / VPR, c0 + VE * (idx % VPR)));
  }
}

template <typename T, int NT = 256>
__device__ __forceinline__ void block64_put(const DenseBlock64Regs<T, NT>& b, T* lds, int tid) {
  constexpr int VE = Tr<T>::VE, VPR = 64 / VE, NV = 4096 / VE;
#pragma unroll
  for (int q = 0; q < NV / NT; ++q) {
    const int idx = tid + NT * q;
    Tr<T>::lds_store(lds + (idx / VPR) * DI_LD + VE * (idx % VPR), b.v[q]);
  }
}

// ---- lane-level helpers of the in-register 16 x 16 eliminations (k_factor.hip, k_small.hip) ----------------
template <typename T>
__device__ __forceinline__ T bcast_lane(T v, int src) {   // value of v in lane src (uniform src)
  if constexpr (sizeof(T) == 8) {
    const int lo = __builtin_amdgcn_readlane(__double2loint((double)v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint((double)v), src);
    return (T)__hiloint2double(hi, lo);
  } else {
    return (T)__int_as_float(__builtin_amdgcn_readlane(__float_as_int((float)v), src));
  }
}

// 1 / d to working precision without the division sequence: hardware reciprocal + two Newton steps
// (the elimination's critical path runs through this once per pivot).  Measured (tools/rcp_probe.hip, 2^20 arguments over
// 40 binades): v_rcp_f64 4.6e-8 relative, one step 2.2e-15 (10 ulp), two steps 0.5 ulp; v_rsq_f64 5.2e-8 / 4.1e-15 / 0.6 ulp:
// one step would save two of a pivot's nine fp64 vector instructions and cost the factors an order of magnitude.
template <typename T>
__device__ __forceinline__ T fast_recip(T d) {
  if constexpr (sizeof(T) == 8) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
  } else {
    float r = __builtin_amdgcn_rcpf(d);
    r = fmaf(fmaf(-d, r, 1.0f), r, r);
    return r;
  }
}

// 1 / sqrt(d) to working precision without the square-root and division sequences (each a dozen dependent
// instructions): hardware reciprocal square root + two Newton steps  r <- r + r (1 - d r^2) / 2
template <typename T>
__device__ __forceinline__ T fast_rsqrt(T d) {
  if constexpr (sizeof(T) == 8) {
    double r = __builtin_amdgcn_rsq(d);
    r = fma(0.5 * r, fma(-d * r, r, 1.0), r);
    r = fma(0.5 * r, fma(-d * r, r, 1.0), r);
    return r;
  } else {
    float r = __builtin_amdgcn_rsqf(d);
    r = fmaf(0.5f * r, fmaf(-d * r, r, 1.0f), r);
    return r;
  }
}

// the diagonal element of a 16 x 16 block in accumulator layout that this lane holds, if it holds one (lanes whose
// rows include row l15); one select chain instead of a predicated computation per register
template <typename T>
__device__ __forceinline__ bool acc_diag(const typename Tr<T>::acc_t& t, int l15, int l4, T& d) {
  d = t[0];
  bool mine = false;
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (Tr<T>::acc_row(l4, r) == l15) {
      d = t[r];
      mine = true;
    }
  return mine;
}

// value of v in the lane whose byte address (4 * lane) is addr
template <typename T>
__device__ __forceinline__ T bperm(int addr, T v) {
  if constexpr (sizeof(T) == 8) {
    const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint((double)v));
    const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint((double)v));
    return (T)__hiloint2double(hi, lo);
  } else {
    return (T)__int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int((float)v)));
  }
}

// ---- in-register factorisation of a 16 x 16 block on the matrix pipe (round 3) --------------------------------------
// The block T (symmetric, positive definite) and the carried identity Y sit in MFMA ACCUMULATOR layout: element
// (row, col) in lane (l15 = col, l4), register r with row = acc_row(l4, r).  Gaussian elimination step k,
//     T[i][:] -= (T[i][k] / d) T[k][:],   Y[i][:] -= (T[i][k] / d) Y[k][:]      for the rows i > k,
// is a rank-1 update, i.e. ONE matrix instruction per matrix whose only non-zero k-slot is the one the lanes of row
// k feed: those lanes (l4 = lk4) hold T[k][j] in register rk -- as it stands the B operand (row k) and, the trailing
// block being symmetric, also T[j][k], the column the multipliers come from.  No lane permutes, no LDS traffic, two
// matrix instructions and a dozen vector ones per pivot (the permute-based sweep it replaces: 18 ds_bpermute and ~60
// instructions).  Rows <= k are left alone (their multiplier is zero), so after the sweep
//     T[k][j], j >= k  =  L[j][k] L[k][k]   (row k never changes after step k: the unscaled column k of L),
//     Y[i][j], j <= i  =  (L^-1)[i][j] L[i][i].
// tol_lane: lane l holds the pivot threshold of row l & 15.  A pivot at or below it (or NaN) is replaced by 1 and
// flagged, as before.
// The pivot chain.  Taking pivot k + 1 out of the result of matrix instruction k (wait for it, read a lane, test,
// reciprocal with two Newton steps, multiply, issue) made a pivot cost 337-377 cycles of which the instruction itself
// is 64 + 15 (tools/lat_probe.hip).  Pivot k + 1 does not need that result:
//     d_{k+1} = T_k[k+1][k+1] - T_k[k][k+1]^2 / d_k
// is made of two elements of T_k, read off the accumulators BEFORE instruction k takes them, so the threshold test
// and the reciprocal of pivot k + 1 no longer wait for instruction k's result; what is left between the result and the
// next instruction is a select and one multiplication.  Measured: 377 -> 349 cycles per pivot, not the ~140 of the
// dependence graph: an fp64 matrix instruction holds its own wave for its 64 cycles (nothing of the wave issues behind
// it; the SIMD's other wave is free to), so a pivot is the two instructions plus the chain's vector instructions in
// series.  (The diagonal element the instruction writes and the d the multipliers were made with may differ in the last
// bit: rounding-level, like any reordering.)
// The pivot row enters the B operand without its own diagonal element (column k of the rows below is never read again,
// and a NaN there would spread over the whole column through 0 * NaN); a pivot at or below its threshold is replaced by
// 1 for the multipliers at once and on the block's diagonal after the sweep.
// A two-pivots-per-step form (2 x 2 pivot blocks: half the matrix instructions and reciprocals, one lane-pair exchange
// per step) was written against the old chain and measured slower, 755 against 674 cycles per pair, and not kept.
__device__ __forceinline__ void factor16_acc_b4(d4& t, d4& y, const double tol_lane, const int lane, int& bad);

// SEQ: the one-pivot-a-step sweep also for fp64 (the panel kernels of k_factor.hip, see there; tools/factor16_probe.hip);
// default for fp64: the four-pivot form below (the small-problem kernels)
template <typename T, bool SEQ = false>
__device__ __forceinline__ void factor16_acc(typename Tr<T>::acc_t& t, typename Tr<T>::acc_t& y, const double tol_lane,
                                             const int lane, int& bad) {
  constexpr bool F64 = sizeof(T) == 8;
  if constexpr (F64 && !SEQ) {
    // round 5: four pivots a step (factor16_acc_b4): 3844 against 5686 ticks a block alone on its SIMD, 4120 against
    // 7184 with every SIMD of the chip at it (tools/factor16_probe.hip)
    factor16_acc_b4(t, y, tol_lane, lane, bad);
    return;
  }
  const int l15 = lane & 15, l4 = lane >> 4;
  unsigned replaced = 0;      // bit k: pivot k was not positive (or NaN)
  auto accept = [&](T d, const int k) -> T {      // pivot k against its threshold; 1 / pivot
    const double tol = bcast_lane<double>(tol_lane, k);
    if (!((double)d > tol)) {
      d = (T)1;
      replaced |= 1u << k;
    }
    return fast_recip<T>(d);
  };
  T rinv = accept(bcast_lane<T>(t[0], 0), 0);
  // row k sits in the lanes l4 = lk4, register rk: (k & 3, k >> 2) in the f64 result layout, (k >> 2, k & 3) in the f32
  // one; row k + 1 in (lkn, rkn).  The register indices must be static, the lane groups need not: the loops over the
  // lane group stay run-time loops (sixteen unrolled steps cost instruction cache and scalar registers for nothing).
  auto step = [&](const int k, const int lk4, auto rk_tag, const int lkn, auto rkn_tag, auto last_tag) {
    constexpr int rk = decltype(rk_tag)::value, rkn = decltype(rkn_tag)::value;
    constexpr bool last = decltype(last_tag)::value;
    const bool rowk = (l4 == lk4);
    const T vt = (rowk && l15 > k) ? t[rk] : (T)0;          // T[k][l15], l15 > k, in the one live k-slot ...
    const T vy = rowk ? y[rk] : (T)0;                       // Y[k][l15]
    const T a = -vt * rinv;                                 // ... and, T_k being symmetric, -T[l15][k] / d as well
    T sx = (T)0, sdn = (T)1;
    if constexpr (!last) {
      sx = bcast_lane<T>(t[rk], k + 1 + 16 * lk4);          // T_k[k][k+1]
      sdn = bcast_lane<T>(t[rkn], k + 1 + 16 * lkn);        // T_k[k+1][k+1]
    }
    t = Tr<T>::mfma(a, vt, t);
    y = Tr<T>::mfma(a, vy, y);
    if constexpr (!last) rinv = accept((T)fma(-(sx * rinv), sx, sdn), k + 1);   // the same operations as the instruction's
  };
  using std::integral_constant;
  typedef integral_constant<bool, false> more_t;
  if constexpr (F64) {      // k = 4 rk + lk4: pivots in order when the register index is the outer loop
    auto sweep = [&](auto rk_tag, auto rkn_tag, auto last_tag) {
      constexpr int rk = decltype(rk_tag)::value;
#pragma unroll 1
      for (int lk4 = 0; lk4 < 3; ++lk4) step(4 * rk + lk4, lk4, rk_tag, lk4 + 1, rk_tag, more_t());
      step(4 * rk + 3, 3, rk_tag, 0, rkn_tag, last_tag);
    };
    sweep(integral_constant<int, 0>(), integral_constant<int, 1>(), more_t());
    sweep(integral_constant<int, 1>(), integral_constant<int, 2>(), more_t());
    sweep(integral_constant<int, 2>(), integral_constant<int, 3>(), more_t());
    sweep(integral_constant<int, 3>(), integral_constant<int, 3>(), integral_constant<bool, true>());
  } else {                  // k = 4 lk4 + rk: the lane group is the outer loop
    auto group = [&](const int lk4, auto last_tag) {
      step(4 * lk4 + 0, lk4, integral_constant<int, 0>(), lk4, integral_constant<int, 1>(), more_t());
      step(4 * lk4 + 1, lk4, integral_constant<int, 1>(), lk4, integral_constant<int, 2>(), more_t());
      step(4 * lk4 + 2, lk4, integral_constant<int, 2>(), lk4, integral_constant<int, 3>(), more_t());
      step(4 * lk4 + 3, lk4, integral_constant<int, 3>(), lk4 + 1, integral_constant<int, 0>(), last_tag);
    };
#pragma unroll 1
    for (int lk4 = 0; lk4 < 3; ++lk4) group(lk4, more_t());
    group(3, integral_constant<bool, true>());
  }
  if (replaced) {             // numerically not positive definite: flagged, unit pivots on the diagonal
    bad = 1;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (Tr<T>::acc_row(l4, r) == l15 && ((replaced >> l15) & 1u)) t[r] = (T)1;
  }
}

// ---- the same elimination four pivots at a time (fp64 result layout; round 5) -----------------------------------------
// In the f64 result layout the four rows 4 q .. 4 q + 3 are the four lane groups of register q: as it stands the B operand
// of a matrix instruction whose four k-slots are those rows.  With S the 4 x 4 diagonal block of those rows, S = M D M^T
// (M unit lower triangular) and W = M^-1, the four sequential steps are
//     P = W T[R, :]      the four pivot rows as the sequential sweep leaves them      (one instruction: A = W, B = t[q])
//     Q = W Y[R, :]                                                                   (one instruction)
//     T[m, :] -= sum_i (P_i[m] / d_i) P_i[:],   Y[m, :] -= sum_i (P_i[m] / d_i) Q_i[:]   for the rows m beyond the block
// -- two more instructions whose A operand is -P / d read off the result registers of the first (an accumulator tile
// read as an A operand is its own transpose) and whose B operands are P and Q as they stand.  Four instructions per four
// pivots instead of eight, and -- what the chain is made of -- the 4 x 4 factorisation runs on ten values every lane
// holds (read off the accumulators once per block), four reciprocals and ~35 dependent vector instructions per FOUR
// pivots instead of nine per pivot behind a matrix instruction each.  Same contract as factor16_acc (what t and y hold
// afterwards, thresholds, replaced pivots); the results differ from the sequential sweep's by rounding only.
__device__ __forceinline__ void factor16_acc_b4(d4& t, d4& y, const double tol_lane, const int lane, int& bad) {
  const int l15 = lane & 15, l4 = lane >> 4;
  unsigned replaced = 0;
  auto block = [&](auto q_tag, auto last_tag) {
    constexpr int q = decltype(q_tag)::value, c0 = 4 * q;
    constexpr bool last = decltype(last_tag)::value;
    double s[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) s[i][j] = bcast_lane<double>(t[q], (c0 + j) + 16 * i);
    double rinv[4], m[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double d = s[j][j];
      const double tol = bcast_lane<double>(tol_lane, c0 + j);
      if (!(d > tol)) {
        d = 1.0;
        replaced |= 1u << (c0 + j);
      }
      rinv[j] = fast_recip<double>(d);
#pragma unroll
      for (int i = j + 1; i < 4; ++i) m[i][j] = s[i][j] * rinv[j];
#pragma unroll
      for (int i = j + 1; i < 4; ++i)
#pragma unroll
        for (int k = j + 1; k <= i; ++k) s[i][k] = fma(-m[i][j], s[k][j], s[i][k]);
    }
    // W = M^-1 (unit lower triangular): w[i][j] = -sum_{k = j .. i-1} m[i][k] w[k][j]
    double w[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      w[i][i] = 1.0;
#pragma unroll
      for (int j = i - 1; j >= 0; --j) {
        double v = -m[i][j];
#pragma unroll
        for (int k = j + 1; k < i; ++k) v = fma(-m[i][k], w[k][j], v);
        w[i][j] = v;
      }
    }
    // A operand of the two transforms: lane (l15 = row i of the result, l4 = k-slot) = W[i][l4] for i < 4, k <= i
    double aw = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int k = 0; k <= i; ++k)
        if (l15 == i && l4 == k) aw = w[i][k];
    const double bt = (l15 >= c0) ? t[q] : 0.0;      // columns left of the block are never read again (and may be anything)
    const d4 z4 = {0.0, 0.0, 0.0, 0.0};
    const d4 pa = Tr<double>::mfma(aw, bt, z4);       // pa[0]: lane (l15 = n, l4 = i) = P_i[n]
    const d4 qa = Tr<double>::mfma(aw, y[q], z4);     // qa[0]: Q_i[n]
    if (l15 >= c0) t[q] = pa[0];
    y[q] = qa[0];
    if constexpr (!last) {
      const double ri = (l4 == 0) ? rinv[0] : (l4 == 1) ? rinv[1] : (l4 == 2) ? rinv[2] : rinv[3];
      const double a2 = (l15 > c0 + 3) ? -pa[0] * ri : 0.0;           // -P_i[m] / d_i for the rows m beyond the block
      const double b2 = (l15 > c0 + l4) ? pa[0] : 0.0;                // P_i[n] right of its own diagonal
      t = Tr<double>::mfma(a2, b2, t);
      y = Tr<double>::mfma(a2, qa[0], y);
    }
  };
  using std::integral_constant;
  block(integral_constant<int, 0>(), integral_constant<bool, false>());
  block(integral_constant<int, 1>(), integral_constant<bool, false>());
  block(integral_constant<int, 2>(), integral_constant<bool, false>());
  block(integral_constant<int, 3>(), integral_constant<bool, true>());
  if (replaced) {             // numerically not positive definite: flagged, unit pivots on the diagonal
    bad = 1;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (Tr<double>::acc_row(l4, r) == l15 && ((replaced >> l15) & 1u)) t[r] = 1.0;
  }
}

// fp64 helpers kept for the Gram kernel (always fp64 accumulation)
__device__ __forceinline__ d4 mfma(double a, double b, d4 c) { return Tr<double>::mfma(a, b, c); }
__device__ __forceinline__ int acc_row(int l4, int r) { return Tr<double>::acc_row(l4, r); }
__device__ __forceinline__ d4 d4_zero() { return Tr<double>::zero(); }

}  // namespace lsspa
