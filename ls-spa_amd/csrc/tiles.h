// Tile-level building blocks shared by the LS-SPA kernels (gfx950 / CDNA4 only).
//
// All dense inner products run on the fp64 matrix pipe, v_mfma_f64_16x16x4_f64:
// one wave computes a 16x16 tile of D = A*B + C with K = 4 per instruction.
//   A operand : one double per lane, lane l holds A[i = l & 15][k = l >> 4]
//   B operand : one double per lane, lane l holds B[k = l >> 4][j = l & 15]
//   C/D       : four doubles per lane, register r of lane l holds
//               D[row = (l >> 4) + 4 r][col = l & 15]
// (cdna_hip_programming.md section 3, "f64 MFMA does NOT use these maps").
//
// Consequence used throughout: register r of an accumulator tile is, as it stands,
// the B operand of k-step r of a following product that sums over the tile's ROW
// index (k = 4 r + (l >> 4)).  The triangular solves "X = Dinv * C" therefore take
// C straight from the accumulators, with no LDS round trip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lsspa {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));  // 16-byte staging unit

constexpr int NB = 64;        // factorisation block (diagonal blocks are NB x NB)
constexpr int KCH = 16;       // k-chunk staged per main-loop iteration
constexpr int RK_LD = 18;     // LDS row stride (doubles) of a [rows][16 k] tile: 144 B, 16-B aligned,
                              // conflict-free for the (row = l&15, k = l>>4) fragment read
constexpr int KC_LD = 144;    // LDS row stride (doubles) of a [16 k][128 cols] tile: +16 doubles puts
                              // the two k rows of a 32-lane group on disjoint bank halves
constexpr int DI_LD = 66;     // LDS row stride of a 64 x 64 block read as an A operand
constexpr int TT_LD = 65;     // LDS row stride of the in-LDS elimination tiles (column walks)
// Layout of the per-ordering work matrices in HBM: CHUNK-MAJOR.  A p_pad x p_pad matrix is stored as
// p_pad/16 column chunks of 16 columns; inside a chunk the rows follow each other (128 B per row):
//     element (r, c)  ->  ((c >> 4) * p_pad + r) * 16 + (c & 15)
// Every operand tile the factorisation kernels stage -- R rows x 16 k -- is then ONE contiguous
// R * 128-byte block (8-16 KB) instead of R separate 128-byte row segments 8 KB apart, which is what
// a row-major matrix gives and what held the k-loops at ~3.9 TB/s.
// The tile loaders below take it as a row-major tile with row stride CM_LD = 16.
constexpr int CM_LD = 16;
__host__ __device__ inline int64_t cm_off(int p_pad, int r, int c) {
  return ((int64_t)(c >> 4) * p_pad + r) * 16 + (c & 15);
}
// V (solve result) stays row-major: its tiles are 16 rows x 1 KB.  32 extra doubles per row keep a
// tile's rows from landing on the same few memory channels (m_pad is a multiple of 128).
constexpr int LD_PAD = 32;
__host__ __device__ inline int64_t ldv_of(int m_pad) { return (int64_t)m_pad + LD_PAD; }
// rows of a V matrix: the ordering's row blocks rounded up to the 128-row strip step
__host__ __device__ inline int64_t v_rows_of(int p) { return (int64_t)((p + 127) / 128) * 128; }

__device__ __forceinline__ d4 mfma(double a, double b, d4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
// row of an accumulator element inside its 16 x 16 tile
__device__ __forceinline__ int acc_row(int l4, int r) { return l4 + 4 * r; }

// Note on v_mfma_f64_4x4x4_4b_f64: it issues every 16.5 cycles (32 flop/clk/SIMD, 70-76 TFLOP/s in
// tools/mfma_bench3.hip / mfma_bench4.hip) against ~99 cycles per 2048-flop 16x16x4 instruction
// (47 TFLOP/s), and a 16x16x4 step can be built from four of them (lane maps probed in
// tools/mfma_probe4.hip: lane l = 16 q + 4 g + t holds A_g[i=t][k=q], B_g[k=q][j=t], D_g[i=q][j=t];
// CBSZ/ABID broadcast is ignored for f64).  Panel and strip kernels written that way (git history:
// "Experimental 4x4x4-MFMA kernel variants") were correct but not faster in situ -- 12 LDS operand
// reads per 32 MFMAs instead of 6 per 8, lane rotations in the solve stage, lower occupancy -- so
// the 16x16x4 form is kept for now.
__device__ __forceinline__ d4 d4_zero() {
  d4 z = {0.0, 0.0, 0.0, 0.0};
  return z;
}

// ---- register staging of a [R rows][16 k] tile (R = 64 or 128), 256 threads ----
template <int R>
struct RKRegs {
  v2d v[R / 32];
};

template <int R>
__device__ __forceinline__ void rk_load(RKRegs<R>& r, const double* __restrict__ src, int64_t ld,
                                        int tid, int rows_valid) {
  const int c = tid & 7, row = tid >> 3;
#pragma unroll
  for (int q = 0; q < R / 32; ++q) {
    const int rr = row + 32 * q;
    if (rr < rows_valid)
      r.v[q] = *reinterpret_cast<const v2d*>(src + (int64_t)rr * ld + 2 * c);
    else
      r.v[q] = v2d{0.0, 0.0};
  }
}

template <int R>
__device__ __forceinline__ void rk_store(const RKRegs<R>& r, double* lds, int tid) {
  const int c = tid & 7, row = tid >> 3;
#pragma unroll
  for (int q = 0; q < R / 32; ++q)
    *reinterpret_cast<v2d*>(lds + (row + 32 * q) * RK_LD + 2 * c) = r.v[q];
}

// ---- register staging of a [16 k][128 cols] tile, 256 threads ----
struct KCRegs {
  v2d v0, v1, v2, v3;
};

__device__ __forceinline__ void kc_load(KCRegs& r, const double* __restrict__ src, int64_t ld, int tid) {
  const int c = tid & 63, k = tid >> 6;
  const double* s = src + (int64_t)k * ld + 2 * c;
  r.v0 = *reinterpret_cast<const v2d*>(s);
  r.v1 = *reinterpret_cast<const v2d*>(s + 4 * ld);
  r.v2 = *reinterpret_cast<const v2d*>(s + 8 * ld);
  r.v3 = *reinterpret_cast<const v2d*>(s + 12 * ld);
}

__device__ __forceinline__ void kc_store(const KCRegs& r, double* lds, int tid) {
  const int c = tid & 63, k = tid >> 6;
  double* d = lds + k * KC_LD + 2 * c;
  *reinterpret_cast<v2d*>(d) = r.v0;
  *reinterpret_cast<v2d*>(d + 4 * KC_LD) = r.v1;
  *reinterpret_cast<v2d*>(d + 8 * KC_LD) = r.v2;
  *reinterpret_cast<v2d*>(d + 12 * KC_LD) = r.v3;
}

// copy a dense 64 x 64 block (row-major, ld 64) from global into LDS with stride DI_LD
template <int NT = 256>
__device__ __forceinline__ void load_block64(double* lds, const double* __restrict__ g, int tid) {
#pragma unroll
  for (int q = 0; q < 2048 / NT; ++q) {
    const int idx = tid + NT * q;  // 16-byte piece index, 2048 in all
    const int row = idx >> 5, c2 = idx & 31;
    *reinterpret_cast<v2d*>(lds + row * DI_LD + 2 * c2) =
        *reinterpret_cast<const v2d*>(g + row * 64 + 2 * c2);
  }
}

}  // namespace lsspa
