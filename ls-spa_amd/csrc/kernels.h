// Host-side launchers of the LS-SPA HIP kernels.  Every launcher checks its shape
// assumptions before launching (a faulting kernel can take the whole node down).
//
// The per-ordering kernels exist in two element types: fp64 (default; parity with the reference
// to ~1e-15) and fp32 (work matrices, factors and V in float; Gram reduction, lift accumulation and
// running statistics stay fp64).  `f32 != 0` selects the fp32 instantiation; the `void*` work
// buffers then hold floats.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <vector>

namespace lsspa {

// Dynamic LDS beyond the 64 KB default needs hipFuncAttributeMaxDynamicSharedMemorySize, which is set on the CURRENT
// device's copy of the function: remember the size granted per device (a second engine on another GPU of the same
// process must set it again), under a lock (two engines may launch from two host threads).
struct DynLdsGrant {
  static constexpr int MAX_DEVICES = 64;
  std::mutex mu;
  size_t granted[MAX_DEVICES] = {0};
  hipError_t ensure(const void* fn, size_t bytes) {
    if (bytes <= 64 * 1024) return hipSuccess;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    if (dev >= 0 && dev < MAX_DEVICES && granted[dev] >= bytes) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && dev >= 0 && dev < MAX_DEVICES) granted[dev] = bytes;
    return e;
  }
};
constexpr size_t LDS_BYTES_PER_CU = 160 * 1024;   // gfx950

struct GatherArgs {
  const double* S[2];      // source Gram matrices (train, test), fp64, row-major, stride ld_src
  const float* Sf[2];      // optional fp32 copies of S (same stride): read instead of S when the work matrices are
                           // fp32 -- same values after rounding, half the source traffic
  const double* s[2];      // source right-hand sides (g, h)
  double aug[2];           // diagonal value of the augmented row
  int64_t ld_src;
  const int32_t* perms;    // [n_ord][p]
  int p, p_pad, n_ord, n_src;  // n_src = 1 (train only) or 2
  void* A;                 // [n_src * n_ord] chunk-major p_pad x p_pad matrices, lower triangles written
  double* diag0;           // [n_src * n_ord][p_pad]: the permuted diagonals before any update (pivot scale)
  int f32;
  int paired;              // orderings 2 s and 2 s + 1 are each other's reverse (antithetical pairs): one pass over
                           // the source rows writes both matrices
};
hipError_t launch_gather(const GatherArgs& a, hipStream_t st);
int max_features();   // largest p the per-ordering kernels take (32-bit element counts of one work matrix)
// dst[i] = (float)src[i]
hipError_t launch_to_f32(const double* src, float* dst, int64_t count, hipStream_t st);

// Fused small-p path (k_small.hip): one workgroup per ordering, gather -> two Choleskys -> V -> lifts in LDS.
// fp64, tri mode, p + 1 <= 128.  With per_sample == 2 orderings 2 s and 2 s + 1 each add half their lift vector
// to sample s (lifts must be zero beforehand); with 1 the lift vector is stored.
struct SmallArgs {
  const double* S[2];      // source Gram matrices (train, test), row-major, stride ld_src
  const double* s[2];      // source right-hand sides (g, h)
  double aug[2];           // diagonal value of the augmented row
  int64_t ld_src;
  const int32_t* perms;    // [n_ord][p]; with fwd_only [n_ord / 2][p]: ordering 2 s + 1 is sample s's read backwards
  int fwd_only;            // (per_sample == 2 only) the host stages and uploads half as much
  int p, nb, n_ord, per_sample;   // nb = ceil((p + 1) / 16)
  double* lifts;           // [n_ord / per_sample][p]
  double y_norm_sq;
  double piv_tol;
  int32_t* info;
  int variant;             // 0: the register-resident kernel where it applies (nb <= 7); 1: the LDS-resident kernel
  // the register-resident kernel checks every ordering's sum of lifts against the full model's R^2 itself (sum_tol >= 0;
  // off: -1): LSSPA_INFO_SUM into info[0] beyond sum_tol, the largest deviation above sum_quiet into info[2..3]
  double r2, sum_tol, sum_quiet;
};
bool small_p_checks_sum(const SmallArgs& a);
// host_perms.cpp: every row of perms [B][p] a permutation of 0..p-1?  (sets by AVX2 for 8 <= p <= 128, stamps otherwise;
// _plain: the stamp loop alone, the test's comparison)
bool all_permutations(const int32_t* perms, int B, int p, std::vector<int32_t>& mark);
bool all_permutations_plain(const int32_t* perms, int B, int p, std::vector<int32_t>& mark);
// host_perms.cpp: out [B][p] = argsort of the rows of keys [B][p] on up to `threads` native threads; redo [B] = 1 for the
// rows left to the caller (equal keys or a NaN: numpy's order there is its own); returns their number
int64_t argsort_rows_host(const double* keys, int64_t B, int p, int32_t* out, uint8_t* redo, int threads);
// host_perms.cpp: the 'argsort' ordering source as a native thread (Sobol' points by SciPy's recurrence + row argsort)
struct SobolSampler;
SobolSampler* sobol_sampler_create(int p, int bits, const uint64_t* sv_pb, const uint64_t* q0, double scale, int64_t limit,
                                   int block, int64_t ahead, int64_t ahead_unasked, int threads, int rank, int world);
void sobol_sampler_destroy(SobolSampler* s);
int sobol_sampler_take(SobolSampler* s, int64_t count, int32_t* out, int64_t cap, int64_t* n_taken, int64_t* n_own,
                       int64_t* redo_pos, int64_t* redo_id, int64_t* n_redo, const char** err);
bool small_p_eligible(int p);
size_t small_p_lds_bytes(int nb);
hipError_t launch_small_p(const SmallArgs& a, hipStream_t st);

// Blocked Cholesky, 128-wide panels (p_pad a multiple of 128).  diag0 / piv_tol: a pivot d counts as non-positive
// (LSSPA_INFO_NOT_PD) when d <= piv_tol * diag0.  One diagonal launch (block 0), then panel steps Jo = 0 ..
// p_pad/128 - 2: step Jo computes L[I, Jo] for the tiles below and its tile-0 workgroups update and factor diagonal
// block Jo + 1.
hipError_t launch_chol2_diag(void* A, void* Dinv, const double* diag0, double piv_tol, int32_t* info, int p_pad,
                             int n_mats, int f32, hipStream_t st,
                             int32_t* row_flags = nullptr);
// p_live: rows at or beyond it are identity padding (p + 1 rounded up to 16; 0 = none known): their all-zero
// accumulator tiles are left out of the products.
// X != null (tri mode): the matrices are [n_ord training][n_ord test] and step Jo also computes block column Jo of
// X = V^T = L_t^T L^-T ([n_ord] chunk-major p_pad x p_pad matrices, upper block triangle written) as extra tiles of
// the training factorisation; there is then one more step, Jo = p_pad/128 - 1, with X tiles only.
// pl (with X tiles): the lift scan of V^T is done by the X tiles themselves, block by block, before the block leaves
// the chip -- the lift kernel's pass over V^T falls away (launch_lift with fused = 1 only finishes).  flags must be
// zero before launch 0 of a batch; mode 2 also leaves the last panel's V^T unstored (nobody reads it then).
struct PanelLift {
  int32_t* flags;          // [n_mats]
  double* run;             // [n_ord][p_pad]
  double* Ppart;           // [n_ord][pstride], row block I' of V^T at [I' * p_pad]
  int64_t pstride;
  int p;                   // features
  int mode;                // 0 off, 1 scan, 2 scan + last panel of V^T not stored
};
hipError_t launch_chol2_panel(void* A, void* Dinv, const double* diag0, double piv_tol, int32_t* info, int p_pad,
                              int Jo, int n_mats, int f32, hipStream_t st, int flags = 0, int p_live = 0,
                              void* X = nullptr, int n_ord = 0, const PanelLift* pl = nullptr);

struct StripArgs {
  const void* A;           // factored train matrices
  const void* Dinv;        // [n_mats][nblk][64][64]; the first n_ord entries belong to A
  const void* rhs;         // tri: factored test matrices (same layout as A); rect: unused
  const double* Ft;        // rect: transposed test factor [p][m_pad], fp64
  const int32_t* perms;    // rect only
  void* V;                 // [n_ord][v_rows][m_pad + 32]
  int p, p_pad, m_pad, n_ord, tri;
  int flags;               // developer A/B switches
  int f32;
  int row_live = 0;        // rows at or beyond it are identity padding of L (p + 1 rounded up to 16; 0 = unknown)
  int col_live = 0;        // columns at or beyond it are zero columns of the right-hand side (0 = unknown)
};
hipError_t launch_strip(const StripArgs& a, hipStream_t st);

struct LiftArgs {
  const void* A;           // factored train matrices (row p holds z)
  const void* At;          // tri: factored test matrices (row p holds y-tilde); rect: null
  const double* ytil;      // rect: [m_pad]
  const void* V;           // vt == 0: V row-major [v_rows][m_pad + 32]; vt != 0: V^T chunk-major p_pad x p_pad (tri)
  int vt = 0;
  const int32_t* perms;    // [n_ord][p]
  double* Ppart;           // [n_ord][m_pad/64][p_pad]
  double* lifts;           // [n_samples][p]
  double y_norm_sq;
  int p, p_pad, m_pad, n_ord, per_sample, tri;  // per_sample = 1 or 2 orderings per sample
  int f32;
  int paired;              // per_sample == 2 and ordering 2 s + 1 is ordering 2 s reversed
  int fused = 0;           // vt: Ppart already holds the X tiles' own sums, one row per 128-row block of V^T (PanelLift)
};
hipError_t launch_lift(const LiftArgs& a, hipStream_t st);
// |sum of a sample's lifts - r2| <= tol for every sample, else bit 8 (LSSPA_INFO_SUM) of info[0]; the largest deviation
// of all launches since the last reset as a double in info[2..3]
hipError_t launch_sum_check(const double* lifts, int n_samples, int p, double r2, double tol, int32_t* info,
                            hipStream_t st);

// pending-batch moments about the current running mean: buf = [n_b, S (p), Q (p x p)]
// parts: workspace of stats_batch_slices(n_samples, p) * (1 + p + p*p) doubles (or NULL: one slice)
int stats_batch_slices(int n_samples, int p);
hipError_t launch_stats_batch(const double* lifts, const double* mean, double* buf, int n_samples, int p,
                              int accumulate, double* parts, hipStream_t st);
// single GPU, small p: batch moments AND merge in one launch (no pending buffer): reads (mean, state[0] = n), writes
// the advanced ones to (mean_out, state_out) -- the caller swaps the buffers -- and updates M2 in place
bool stats_small_fusable(int n_samples, int p);
// the same for up to 32 chunks of samples (first sample and count of each, in `lifts`), folded and merged one after the
// other in one launch; mean_snap [n][p] / n_snap [n] (may be null): mean and n after every chunk
struct StatsChunks {
  static constexpr int MAX = 32;
  int n;
  int first[MAX], count[MAX];
};
hipError_t launch_stats_small_multi(const double* lifts, const double* mean, const double* state, double* mean_out,
                                    double* state_out, double* M2, const StatsChunks& ch, int p, double* mean_snap,
                                    double* n_snap, hipStream_t st);
hipError_t launch_stats_small_fused(const double* lifts, const double* mean, const double* state, double* mean_out,
                                    double* state_out, double* M2, int n_samples, int p, hipStream_t st);
// Chan merge of the pending batch into (n, mean, M2); n lives in state[0] (state[1] is the fused kernel's ticket and
// must start at zero).  *cleared: the launch also zeroed the pending buffer (small p: one fused kernel)
hipError_t launch_stats_merge(double* buf, double* state_n, double* mean, double* M2, int p, hipStream_t st,
                              bool* cleared);

// upper-triangle packing of the pending-batch buffer for the all-reduce (Q is symmetric):
// packed = [n_b, S (p), Q[i][i..p-1] ...], stats_packed_count(p) = 1 + p + p (p + 1) / 2 elements
int64_t stats_packed_count(int p);
hipError_t launch_stats_pack(const double* buf, double* packed, int p, hipStream_t st);
hipError_t launch_stats_unpack(const double* packed, double* buf, int p, hipStream_t st);

// theta = L^-T z for the factor stored in A (identity ordering), single workgroup; theta is fp64
// wg: workspace of p doubles for p beyond what a CU's LDS holds (may be null below that)
hipError_t launch_backsolve(const void* A, double* theta, int p, int p_pad, int f32, hipStream_t st,
                            double* wg = nullptr);

// Gram contraction  C = Z^T Z, Z = [X | y]  (rows n, P1 = p + 1 columns), fp64 MFMA, split over rows
struct GramArgs {
  const void* X;           // [n][ld] row-major (device)
  const void* y;           // [n]
  int64_t n, ld;
  int p;                   // features; Z has p + 1 columns
  int is_f32;              // element type of X / y
  double* slabs;           // workspace [n_split][n_pairs][128][128]
  int n_split;
  double* C;               // out: [P1pad][P1pad] full symmetric, P1pad = round_up(p + 1, 128)
  int accumulate;          // C += (row chunks of a streamed matrix) instead of C =
  int variant = 0;         // developer A/B switches (bit 0: workgroup id = unit, no XCD-contiguous map)
};
size_t gram_workspace_bytes(int p, int n_split);
int gram_default_split(int64_t n, int p);
hipError_t launch_gram(const GramArgs& a, hipStream_t st);
// G[a][b] = C[a][b] * scale + (a == b) * reg ; g[a] = C[p][a] * scale ; scalars[0] = C[p][p] * scale
hipError_t launch_gram_finalize(const double* C, int p, double scale, double reg, double* G, int64_t ldg,
                                double* g, double* scalar_out, hipStream_t st);

// Device-side error estimator.  launch_error_draws: draws[d][a] = sum_k Xi[d][k] (H[k][a] - mean[a]) * scale for
// the 1024 draws d (Xi [1024][ldxi], H [n_pad][ldh], both zero-padded to n_pad samples, n_pad a multiple of 16;
// ldh and ldd cover ceil(p/128)*128 columns).  launch_error_quantiles: out[a] = 0.95-quantile of |draws[:, a]|,
// out[p] = 0.95-quantile of the row 2-norms (norms [1024] is a workspace).
constexpr int ERR_DRAWS = 1024;
hipError_t launch_error_draws(const double* Xi, int ldxi, const double* H, int ldh, int n_pad,
                              const double* mean, double scale, int p, double* draws, int ldd, hipStream_t st);
// pack_mean / pack_n (optional): out is [2 p + 2] and also receives the running mean and n behind the p + 1 quantiles.
hipError_t launch_error_quantiles(const double* draws, int ldd, int p, double* norms, double* out,
                                  hipStream_t st, const double* pack_mean = nullptr, const double* pack_n = nullptr);
// the same on x = (D - s mean^T) * scale evaluated as it is read (one rank: no draws buffer); always packed
hipError_t launch_error_quantiles_running(const double* D, const double* s, const double* mean, double scale, int ld,
                                          int p, double* norms, double* out, const double* pack_n, hipStream_t st);
// Running form of the estimator (k_error.hip): Xi[d][k] = standard normal made by Philox4x32-10 from (seed, sample id
// first_id + k stride, draw d), k < count, zero up to n_pad (a multiple of 16).  launch_error_xi stores Xi [1024][n_pad]
// (also the test hook); launch_error_accumulate does that into the workspace Xi [1024][n_pad] and adds
// D[1024][ldh] += Xi L, s[1024] += Xi 1 for the chunk's lift vectors L; launch_error_running_draws:
// x = (D - s mean^T) * scale.
// the chunks of a group folded into the running estimator by launch_error_group, and the checks that belong to them
struct EstChunks {
  static constexpr int MAX = 32;
  int n;
  int first[MAX], count[MAX], n_pad[MAX];   // first sample in the lane's lift buffer, samples, samples padded to 16
  long long xi_off[MAX];                    // the chunk's normals in the workspace (doubles)
  long long first_id[MAX];                  // global sample number of the chunk's first sample
  double scale[MAX];                        // 1 / sqrt(n (n - 1)) of the chunk's check, 0 = no check
};
struct EstChecks {
  int n;
  int chunk[EstChunks::MAX], slot[EstChunks::MAX];
  double scale[EstChunks::MAX];
};
hipError_t launch_error_group(uint64_t seed, int64_t stride, const EstChunks& ch, const EstChecks& ck, double* Xi,
                              const double* lifts, int p, int ld, double* P, double* S, double* D, double* s,
                              double* Dsnap, double* ssnap, const double* mean_snap, const double* n_snap,
                              double* norms, double* res, hipStream_t st);
hipError_t launch_error_xi(uint64_t seed, int64_t first_id, int64_t stride, int count, int n_pad, double* Xi,
                           hipStream_t st);
// L: raw == 0: [n_pad][ldl = ldh], padded and zero-filled; raw != 0: [count][ldl >= p] as the lift kernels wrote it
hipError_t launch_error_accumulate(uint64_t seed, int64_t first_id, int64_t stride, int count, int n_pad, double* Xi,
                                   const double* L, int ldl, int raw, int ldh, int p, double* D, double* s,
                                   hipStream_t st);
hipError_t launch_error_running_draws(const double* D, const double* s, const double* mean, double scale, int p,
                                      int ld, double* draws, hipStream_t st);

// unit test hook: D = A(16x4) * B(4x16) on one wave through Tr<T>::mfma / acc_row (fp64 or fp32)
hipError_t launch_mfma_probe(const double* A, const double* B, double* D, int f32, hipStream_t st);

}  // namespace lsspa
