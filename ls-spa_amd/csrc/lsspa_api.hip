// C ABI of the LS-SPA engine (include/lsspa.h): context, device memory and the
// orchestration of the per-batch kernel sequence
//     gather -> { chol_diag, chol_panel } x nblk -> strip -> lift -> stats
#include "../../include/lsspa.h"

#include <hip/hip_runtime.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "comm.h"
#include "kernels.h"
#include "tiles.h"

using namespace lsspa;

namespace {

thread_local std::string g_create_error;

struct ProfRec {
  int cls;
  hipEvent_t beg, end;
};

template <typename T>
struct DevBuf {
  T* ptr = nullptr;
  size_t count = 0;
};

}  // namespace

// One lane = everything a batch of orderings needs while its kernels run: work matrices, solve results, staged
// orderings, its lift vectors.  With one lane (default) every batch runs on the context's stream, one after the
// other.  With two lanes (lsspa_set_lanes) successive batches alternate between two workspaces on two streams: the
// kernels of batch k+1 are in flight while batch k drains its launch tails, runs its memory-bound stages or waits
// for its collective -- the statistics (pending buffer, all-reduce, merge) stay in batch order on the context's
// stream.  A lane's batch starts when the other lane's batch is about half done (ev_mid), so the two are in
// different stages (gather / early panels of one beside the late panels / strip of the other) instead of in step.
struct Lane {
  hipStream_t st = nullptr;          // own stream (two-lane mode)
  int cap_ord = 0;                   // orderings the workspace can hold
  int cap_samples = 0;               // samples the lifts buffer can hold
  DevBuf<char> A, V, Dinv;           // raw bytes: esz() per element
  DevBuf<double> Ppart, lifts, diag0;
  DevBuf<double> run;                // [cap_ord][p_pad] running sums of the fused lift scan (tri mode)
  DevBuf<int32_t> row_flags;         // [2 cap_ord] "row p of panel J is final" per work matrix (fused lift scan)
  DevBuf<int32_t> perms_d;
  // pinned staging of the orderings: two buffers in turn, each guarded by the event of its last
  // H2D copy, so the host can prepare batch k+1 while the GPU still runs batch k (no stream sync)
  int32_t* perms_h[2] = {nullptr, nullptr};
  hipEvent_t perms_ev[2] = {nullptr, nullptr};
  bool perms_busy[2] = {false, false};
  int perms_turn = 0;
  size_t perms_h_count = 0;
  // the device side is double-buffered too ([2][cap_ord][p]) and fed by a copy stream, so the H2D copy of
  // batch k+1 runs under the kernels of batch k instead of between two batches on the compute stream
  hipStream_t copy_stream = nullptr;
  hipEvent_t perms_used[2] = {nullptr, nullptr};   // the kernels that read device slot b have run
  bool perms_used_valid[2] = {false, false};
  const int32_t* perms_cur = nullptr;              // device slot of the batch being launched
  bool sum_checked = false;                        // the batch's kernel checked the orderings' sums itself
  bool perms_fwd_only = false;                     // ... which holds the samples' orderings only, not their reverses
  // second stream for the two-slice schedule of a batch (developer flag 32)
  // two-lane hand-offs
  hipEvent_t ev_mid = nullptr, ev_done = nullptr, ev_consumed = nullptr, ev_main = nullptr;
  bool mid_valid = false, done_valid = false, consumed_valid = false;
  bool mid_armed = false;            // the batch being launched still has to record ev_mid
  int64_t problem_seen = -1;         // ctx->problem_epoch this lane's stream has been ordered behind
  int B = 0, per = 1, taken = 0;     // the batch in flight (launched, `taken` of its B samples collected so far)
  bool in_flight = false;
};

struct lsspa_ctx {
  int device = 0;
  int n_cu = 256;                // compute units of the device (hipDeviceProp_t::multiProcessorCount)
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;

  // reduced problem
  bool have_problem = false;
  int p = 0, p_pad = 0, m = 0, m_pad = 0, tri = 0;
  double aug_train = 0.0, y_norm_sq = 0.0;
  double r2 = 0.0;               // R^2 of the full model (lsspa_full_fit): what every lift vector must sum to
  bool r2_valid = false;
  bool r2_f32 = false;           // ... computed while the per-ordering work was fp32 (good to 1e-4, not 1e-9)
  DevBuf<double> G, g, H, h, Ft, ytil, scal;
  DevBuf<float> Gf, Hf;        // fp32 copies of G / H for the fp32 gather, made on first use
  bool src_f32_valid = false;

  // per-batch workspace: one or two LANES (struct Lane above), used in turn by successive batches
  int f32 = 0;          // element type of the per-ordering work matrices (0: double, 1: float)
  size_t esz() const { return f32 ? 4 : 8; }
  DevBuf<int32_t> info_d;
  Lane lanes[2];
  int n_lanes = 1;      // lsspa_set_lanes
  int lane_turn = 0;    // lane of the next batch
  int lane_last = 0;    // lane of the most recent batch (its lifts are what lsspa_full_fit etc. read)
  hipStream_t lane_stream(const Lane& L) const { return n_lanes == 1 ? stream : L.st; }
  // two lanes: what the lanes' streams must see of the context's stream -- the loaded problem and its fp32 copies.
  // An event recorded there whenever those change (problem_epoch counts the changes).
  hipEvent_t ev_problem = nullptr;
  int64_t problem_epoch = 0;

  // running statistics
  DevBuf<double> mean, M2, pend, state_n, stat_parts;
  DevBuf<double> mean_alt, state_alt;   // the other halves of the (mean, n) pair: lsspa_lift_collect(accumulate = 2)
  bool pend_dirty = false;
  // history of lift vectors + device-side error estimator
  DevBuf<double> hist, xi_d, draws, err_out;
  // row-sharded reduction in progress: summed Gram buffers [2][P1pad][P1pad]
  DevBuf<double> Cred;
  bool reduce_open = false;
  int64_t hist_cap = 0, hist_n = 0;
  int ldh() const { return ((p + 127) / 128) * 128; }
  // running form of the estimator (lsspa_error_running_*): D = Xi L, s = Xi 1 of the samples folded in so far; the
  // history buffer then only stages the lift vectors between a collect and the next lsspa_error_advance
  bool run_on = false;
  uint64_t run_seed = 0;
  DevBuf<double> Dacc, sacc;
  static constexpr int RES_SLOTS = 64;
  double* res_h = nullptr;            // pinned [RES_SLOTS][2 p + 2]: feature errors, overall error, mean, n
  double* res_hd = nullptr;           // the same memory as the device sees it: the quantile kernel writes a check's
                                      // results there itself (no copy in the chain of a check)
  size_t res_h_count = 0;
  hipEvent_t res_ev[RES_SLOTS] = {nullptr};
  bool res_valid[RES_SLOTS] = {false};
  int res_via[RES_SLOTS] = {0};       // the slot whose event covers this one (itself, or the last check of its group)
  int flags = 0;
  // collectives (RCCL), see comm.h
  Comm* comm = nullptr;
  DevBuf<double> pack, xfer;     // packed moments; staging of host-side all-gathers
  DevBuf<double> theta_d;        // lsspa_full_fit's back-substitution
  DevBuf<double> mean_snap, n_snap;   // running mean / n after every chunk of a group folded in one launch (small p)
  DevBuf<double> grp_P, grp_S, grp_D, grp_s, grp_norms;   // launch_error_group: products, sums and their snapshots
  // the streamed reduction's staging (two row chunks in flight), its copy stream and events: kept between calls
  DevBuf<char> red_x[2], red_y[2];
  hipStream_t red_cs = nullptr;
  hipEvent_t red_copied[2] = {nullptr, nullptr}, red_consumed[2] = {nullptr, nullptr};
  DevBuf<int64_t> ibuf;
  int pack_from_p = 2048;        // the moments travel as an upper triangle from this p on
  std::vector<int32_t> perm_mark;   // scratch of the ordering validation
  bool general_path_once = false;   // set while the factors themselves are wanted (full_fit, get_factors, debug_factor)
  int fail_alloc_in = 0;   // test hook (lsspa_debug_fail_alloc): the n-th device allocation from now fails

  // host seconds of the last reduction's parts (lsspa_reduce_timing): streamed copies + Gram kernels, finalize
  // (scaling, Cholesky-side set-up of the statistics, sync)
  double red_stream_s = 0.0, red_finalize_s = 0.0;

  // profiling
  bool prof_on = false;
  std::vector<ProfRec> prof_recs;
  double prof_ms[LSSPA_K_COUNT] = {0};
  int64_t prof_n[LSSPA_K_COUNT] = {0};

  int fail(int code, const char* what, hipError_t e = hipSuccess) {
    char buf[512];
    if (e != hipSuccess)
      snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    else
      snprintf(buf, sizeof buf, "%s", what);
    err = buf;
    return code;
  }
};

namespace {

#define HIPCHK(expr)                                              \
  do {                                                            \
    hipError_t e_ = (expr);                                       \
    if (e_ != hipSuccess) return ctx->fail(LSSPA_ERR_HIP, #expr, e_); \
  } while (0)

template <typename T>
int dev_alloc(lsspa_ctx* ctx, DevBuf<T>& b, size_t count) {
  if (b.count >= count && b.ptr) return LSSPA_OK;
  if (b.ptr) (void)hipFree(b.ptr);
  b.ptr = nullptr;
  b.count = 0;
  if (ctx->fail_alloc_in > 0 && --ctx->fail_alloc_in == 0)
    return ctx->fail(LSSPA_ERR_NOMEM, "hipMalloc: out of memory (injected by lsspa_debug_fail_alloc)");
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&b.ptr), count * sizeof(T));
  if (e != hipSuccess) {
    b.ptr = nullptr;
    (void)hipGetLastError();
    return ctx->fail(LSSPA_ERR_NOMEM, "hipMalloc", e);
  }
  b.count = count;
  return LSSPA_OK;
}

template <typename T>
void dev_free(DevBuf<T>& b) {
  if (b.ptr) (void)hipFree(b.ptr);
  b.ptr = nullptr;
  b.count = 0;
}

#define TRY(expr)                     \
  do {                                \
    int rc_ = (expr);                 \
    if (rc_ != LSSPA_OK) return rc_;  \
  } while (0)

inline int round_up(int x, int q) { return ((x + q - 1) / q) * q; }

struct ProfScope {
  lsspa_ctx* ctx;
  ProfRec rec;
  bool live;
  hipStream_t st;
  ProfScope(lsspa_ctx* c, int cls, hipStream_t on = nullptr) : ctx(c), live(false), st(on) {
    if (!c || !c->prof_on) return;
    if (!st) st = c->stream;
    rec.cls = cls;
    if (hipEventCreate(&rec.beg) != hipSuccess) return;
    if (hipEventCreate(&rec.end) != hipSuccess) {
      (void)hipEventDestroy(rec.beg);
      return;
    }
    (void)hipEventRecord(rec.beg, st);
    live = true;
  }
  ~ProfScope() {
    if (!live) return;
    (void)hipEventRecord(rec.end, st);
    try {
      ctx->prof_recs.push_back(rec);
    } catch (...) {   // out of host memory: lose the sample, not the process (a throwing destructor terminates)
      (void)hipEventDestroy(rec.beg);
      (void)hipEventDestroy(rec.end);
    }
  }
};

int sync_all(lsspa_ctx* ctx);

int prof_collect(lsspa_ctx* ctx) {
  if (ctx->prof_recs.empty()) return LSSPA_OK;
  TRY(sync_all(ctx));
  for (auto& r : ctx->prof_recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.beg, r.end) == hipSuccess) {
      ctx->prof_ms[r.cls] += ms;
      ctx->prof_n[r.cls] += 1;
    }
    (void)hipEventDestroy(r.beg);
    (void)hipEventDestroy(r.end);
  }
  ctx->prof_recs.clear();
  return LSSPA_OK;
}

// ---- small helper kernels that only the API layer needs ---------------------------------
template <typename T>
__global__ void transpose_test_kernel(const T* __restrict__ X, const T* __restrict__ y, int64_t M, int64_t ld,
                                      int p, int m_pad, double* __restrict__ Ft, double* __restrict__ ytil) {
  // Ft[f][r] = X[r][f]; tiny (M < p), clarity over speed
  const int64_t total = (int64_t)p * m_pad;
  for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
    const int f = (int)(o / m_pad), r = (int)(o - (int64_t)f * m_pad);
    Ft[o] = (r < M) ? (double)X[(int64_t)r * ld + f] : 0.0;
  }
  if (blockIdx.x == 0)
    for (int r = threadIdx.x; r < m_pad; r += 256) ytil[r] = (r < M) ? (double)y[r] : 0.0;
}

__global__ void sumsq_kernel(const double* __restrict__ v, int n, double* __restrict__ out) {
  __shared__ double s[256];
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) a += v[i] * v[i];
  s[threadIdx.x] = a;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (threadIdx.x < k) s[threadIdx.x] += s[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = s[0];
}

// ---- problem set-up -----------------------------------------------------------------------
// Drop the per-batch workspace (it is re-created on demand).  The capacities are zeroed FIRST, so that a context
// whose next allocation fails is left with "no workspace", never with a capacity that points at freed buffers.
void free_lane_workspace(Lane& L) {
  L.cap_ord = 0;
  L.cap_samples = 0;
  L.perms_used_valid[0] = L.perms_used_valid[1] = false;
  dev_free(L.A);
  dev_free(L.V);
  dev_free(L.Dinv);
  dev_free(L.diag0);
  dev_free(L.Ppart);
  dev_free(L.run);
  dev_free(L.row_flags);
  dev_free(L.perms_d);
  dev_free(L.lifts);
}

void free_workspace(lsspa_ctx* ctx) {
  for (Lane& L : ctx->lanes) free_lane_workspace(L);
  dev_free(ctx->stat_parts);
}

// wait for everything this context has in flight: the lanes' streams and the context's own
int sync_all(lsspa_ctx* ctx) {
  for (Lane& L : ctx->lanes) {
    if (L.st) HIPCHK(hipStreamSynchronize(L.st));
    if (L.copy_stream) HIPCHK(hipStreamSynchronize(L.copy_stream));
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return LSSPA_OK;
}

int set_dims(lsspa_ctx* ctx, int p, int m, int tri) {
  if (p < 1 || m < 1) return ctx->fail(LSSPA_ERR_ARG, "p and m must be positive");
  if (tri && m != p) return ctx->fail(LSSPA_ERR_ARG, "tri mode needs m == p");
  if (p > max_features()) {
    // the reference has no limit (ls_spa/ls_spa.py:163); here one work matrix of p_pad^2 elements is indexed with 32
    // bits in places.  Refused now, by name, rather than at the first batch's launch.  (Up to round 3 the limit was
    // 13567: a source row and the ordering had to fit the LDS of a CU; the segmented gather lifted that.)
    char msg[160];
    snprintf(msg, sizeof msg, "p = %d features exceeds the %d this engine supports (32-bit element counts of one "
             "p x p work matrix)", p, max_features());
    return ctx->fail(LSSPA_ERR_ARG, msg);
  }
  TRY(sync_all(ctx));  // buffers below may be re-allocated
  ctx->have_problem = false;
  ctx->r2_valid = false;
  ctx->src_f32_valid = false;
  // a workspace sized for another shape is released now: kept, it would count as unavailable memory when the
  // new one is sized from hipMemGetInfo (and its layout depends on p_pad / m_pad / tri anyway)
  for (Lane& L : ctx->lanes) L.in_flight = false;   // a batch launched on the previous problem is void
  if (p != ctx->p || m != ctx->m || tri != ctx->tri) {
    free_workspace(ctx);
    dev_free(ctx->Gf);
    dev_free(ctx->Hf);
  }
  ctx->p = p;
  ctx->m = m;
  ctx->tri = tri;
  ctx->p_pad = round_up(p + 1, 128);   // the two-level factorisation walks 128-wide panels
  ctx->m_pad = round_up(m, 128);
  const size_t pp = (size_t)ctx->p_pad;
  TRY(dev_alloc(ctx, ctx->G, (size_t)p * pp));
  TRY(dev_alloc(ctx, ctx->g, pp));
  TRY(dev_alloc(ctx, ctx->scal, 8));
  if (tri) {
    TRY(dev_alloc(ctx, ctx->H, (size_t)p * pp));
    TRY(dev_alloc(ctx, ctx->h, pp));
  } else {
    TRY(dev_alloc(ctx, ctx->Ft, (size_t)p * ctx->m_pad));
    TRY(dev_alloc(ctx, ctx->ytil, (size_t)ctx->m_pad));
  }
  TRY(dev_alloc(ctx, ctx->mean, (size_t)p));
  TRY(dev_alloc(ctx, ctx->M2, (size_t)p * p));
  TRY(dev_alloc(ctx, ctx->pend, (size_t)1 + p + (size_t)p * p));
  TRY(dev_alloc(ctx, ctx->state_n, 8));
  TRY(dev_alloc(ctx, ctx->mean_alt, (size_t)p));
  TRY(dev_alloc(ctx, ctx->state_alt, 8));
  TRY(dev_alloc(ctx, ctx->info_d, 8));
  // so does the lift history (row stride): it has to be enabled again for the new problem
  ctx->hist_cap = 0;
  ctx->hist_n = 0;
  ctx->run_on = false;
  return LSSPA_OK;
}

int stats_reset(lsspa_ctx* ctx) {
  const int p = ctx->p;
  HIPCHK(hipMemsetAsync(ctx->mean.ptr, 0, sizeof(double) * p, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->M2.ptr, 0, sizeof(double) * (size_t)p * p, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->pend.ptr, 0, sizeof(double) * ((size_t)1 + p + (size_t)p * p), ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->state_n.ptr, 0, sizeof(double) * 8, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->state_alt.ptr, 0, sizeof(double) * 8, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->info_d.ptr, 0, sizeof(int32_t) * 8, ctx->stream));
  ctx->pend_dirty = false;
  ctx->hist_n = 0;
  if (ctx->run_on) {
    HIPCHK(hipMemsetAsync(ctx->Dacc.ptr, 0, ctx->Dacc.count * 8, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->sacc.ptr, 0, ctx->sacc.count * 8, ctx->stream));
    for (bool& v : ctx->res_valid) v = false;
  }
  return LSSPA_OK;
}

// append rows [rows][p] (device or host, row stride p) to the history
int hist_append(lsspa_ctx* ctx, const double* src, int64_t rows, hipMemcpyKind kind) {
  const size_t ldh = ctx->ldh();
  if (ctx->hist_n + rows > ctx->hist_cap) {
    // grow geometrically; the capacity given to lsspa_history_enable is only the first allocation
    const int64_t cap = std::max<int64_t>(2 * ctx->hist_cap, ctx->hist_n + rows);
    const int64_t cap_rows = ((cap + KCH - 1) / KCH) * KCH;
    DevBuf<double> bigger;
    TRY(dev_alloc(ctx, bigger, (size_t)cap_rows * ldh));
    HIPCHK(hipMemsetAsync(bigger.ptr, 0, bigger.count * 8, ctx->stream));
    if (ctx->hist_n > 0)
      HIPCHK(hipMemcpyAsync(bigger.ptr, ctx->hist.ptr, (size_t)ctx->hist_n * ldh * 8, hipMemcpyDeviceToDevice,
                            ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    dev_free(ctx->hist);
    ctx->hist = bigger;
    ctx->hist_cap = cap;
  }
  HIPCHK(hipMemcpy2DAsync(ctx->hist.ptr + (size_t)ctx->hist_n * ldh, ldh * 8, src, (size_t)ctx->p * 8,
                          (size_t)ctx->p * 8, (size_t)rows, kind, ctx->stream));
  ctx->hist_n += rows;
  return LSSPA_OK;
}

// elements of one ordering's V buffer: V row-major (strip kernel) or, in tri mode, V^T as a chunk-major p_pad x p_pad
// matrix (X tiles of the panel launches) -- room for either, so that developer flag 128 can switch between them
size_t v_elems_per_ordering(const lsspa_ctx* ctx) {
  const size_t rowmajor = (size_t)v_rows_of(ctx->p) * (size_t)ldv_of(ctx->m_pad);
  const size_t pp = ctx->p_pad;
  return ctx->tri ? std::max(rowmajor, pp * pp) : rowmajor;
}

// tri mode computes V^T inside the panel launches; rect mode (and developer flag 128) uses the strip kernel
static inline bool vt_path(const lsspa_ctx* ctx) { return ctx->tri && !(ctx->flags & 128); }
// the X tiles scan their own blocks of V^T for the lifts (developer flag 512: the lift kernel reads V^T back instead)
static inline bool fused_scan(const lsspa_ctx* ctx) { return vt_path(ctx) && !(ctx->flags & 512); }

size_t bytes_per_ordering(const lsspa_ctx* ctx) {
  const size_t pp = ctx->p_pad, nblk = pp / NB;
  const size_t nm = ctx->tri ? 2 : 1;
  const size_t es = ctx->esz();
  // work matrices, V / V^T, inverse diagonal blocks, initial diagonals, partial sums, lift vector; the fused scan's
  // running sums and row flags; the two device copies of the ordering
  return nm * pp * pp * es + v_elems_per_ordering(ctx) * es +
         nm * nblk * 4096 * es + nm * pp * 8 +
         (size_t)(ctx->m_pad / 64) * pp * 8 + (size_t)ctx->p * 8 +
         pp * 8 + 2 * sizeof(int32_t) + 2 * (size_t)ctx->p * sizeof(int32_t);
}

int ensure_workspace(lsspa_ctx* ctx, Lane& L, int want_ord, int want_samples) {
  want_ord = (want_ord + 1) & ~1;  // antithetical pairs stay together
  if (want_ord > L.cap_ord || want_samples > L.cap_samples)
    TRY(sync_all(ctx));  // work in flight still uses the old buffers
  if (want_ord > L.cap_ord) {
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    // memory already held by the old workspace comes back when it is re-allocated
    const size_t per = bytes_per_ordering(ctx);
    size_t budget = (size_t)(0.85 * (double)free_b) + (size_t)L.cap_ord * per;
    if (ctx->n_lanes == 2 && &L == &ctx->lanes[0] && ctx->lanes[1].cap_ord == 0)
      budget /= 2;   // leave the second lane its share
    int cap = (int)std::min<size_t>((size_t)want_ord, budget / per);
    cap = std::min(cap, 4096);
    cap &= ~1;  // keep antithetical pairs together
    if (cap < 2) return ctx->fail(LSSPA_ERR_NOMEM, "not enough device memory for two orderings");
    if (cap > L.cap_ord) {
      // capacity 0 while the buffers are being replaced: if one allocation fails (LSSPA_ERR_NOMEM) the caller may
      // retry with a smaller batch and must then find "no workspace", not the old capacity over freed buffers
      const int samples_kept = L.cap_samples;
      DevBuf<double> lifts_kept = L.lifts;
      L.lifts = DevBuf<double>();
      free_lane_workspace(L);
      L.lifts = lifts_kept;
      L.cap_samples = samples_kept;
      const size_t pp = ctx->p_pad, nblk = pp / NB;
      const size_t nm = ctx->tri ? 2 : 1;
      const size_t es = ctx->esz();
      TRY(dev_alloc(ctx, L.A, nm * cap * pp * pp * es));
      TRY(dev_alloc(ctx, L.V, (size_t)cap * v_elems_per_ordering(ctx) * es));
      TRY(dev_alloc(ctx, L.Dinv, nm * cap * nblk * 4096 * es));
      TRY(dev_alloc(ctx, L.diag0, nm * cap * pp));
      TRY(dev_alloc(ctx, L.Ppart, (size_t)cap * (ctx->m_pad / 64) * pp));
      if (ctx->tri) {
        TRY(dev_alloc(ctx, L.run, (size_t)cap * pp));
        TRY(dev_alloc(ctx, L.row_flags, (size_t)2 * cap));
      }
      TRY(dev_alloc(ctx, L.perms_d, (size_t)2 * cap * ctx->p));
      L.cap_ord = cap;
    }
  }
  if (want_samples > L.cap_samples) {
    L.cap_samples = 0;   // dev_alloc releases the old buffer before it asks for the new one
    TRY(dev_alloc(ctx, L.lifts, (size_t)want_samples * ctx->p));
    L.cap_samples = want_samples;
  }
  return LSSPA_OK;
}

int ensure_pinned(lsspa_ctx* ctx, Lane& L, size_t count) {
  if (L.perms_h_count >= count) return LSSPA_OK;
  TRY(sync_all(ctx));  // pending copies still read the old buffers
  L.perms_h_count = 0;
  for (int b = 0; b < 2; ++b) {
    if (L.perms_h[b]) (void)hipHostFree(L.perms_h[b]);
    L.perms_h[b] = nullptr;
    L.perms_busy[b] = false;
    hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&L.perms_h[b]), count * sizeof(int32_t), 0);
    if (e != hipSuccess) return ctx->fail(LSSPA_ERR_NOMEM, "hipHostMalloc", e);
    if (!L.perms_ev[b]) HIPCHK(hipEventCreateWithFlags(&L.perms_ev[b], hipEventDisableTiming));
  }
  L.perms_h_count = count;
  return LSSPA_OK;
}

// streams and events of a lane, made on first use
int ensure_lane(lsspa_ctx* ctx, Lane& L) {
  if (!L.ev_done) {
    HIPCHK(hipEventCreateWithFlags(&L.ev_mid, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&L.ev_done, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&L.ev_consumed, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&L.ev_main, hipEventDisableTiming));
  }
  if (ctx->n_lanes == 2 && !L.st) HIPCHK(hipStreamCreateWithFlags(&L.st, hipStreamNonBlocking));
  return LSSPA_OK;
}

// small problems take the fused kernel (developer flag 1024 forces the general path, which full_fit, get_factors and
// debug_factor also need: they read the factors back from the work matrices)
static double sum_check_tol(const lsspa_ctx* ctx) {
  return ((ctx->f32 || ctx->r2_f32) ? 1e-4 : 1e-9) * std::max(1.0, std::fabs(ctx->r2));
}

static bool small_path(const lsspa_ctx* ctx) {
  return ctx->tri && !ctx->f32 && small_p_eligible(ctx->p) && !(ctx->flags & 1024) && !ctx->general_path_once;
}

// Run gather -> factorisation -> strip -> lift for n_ord orderings already resident in perms_d.
// lifts for sample s land in lifts_d[(s_off + s)][p].
// Orderings [ord_off, ord_off + n_ord) of the staged batch on stream st.  Each slice owns its part of every
// workspace buffer (a slice's matrices are laid out [source][ordering] inside its own region), so two
// slices can run on two streams at once.
int run_slice(lsspa_ctx* ctx, Lane& L, int ord_off, int n_ord, int per_sample, int s_off, hipStream_t st) {
  const int p = ctx->p, p_pad = ctx->p_pad, m_pad = ctx->m_pad, nblk = p_pad / NB;
  const int n_src = ctx->tri ? 2 : 1;
  const int n_mats = n_ord * n_src;
  const size_t es = ctx->esz(), pp2 = (size_t)p_pad * p_pad;
  char* const A_s = L.A.ptr + (size_t)ord_off * n_src * pp2 * es;
  char* const Dinv_s = L.Dinv.ptr + (size_t)ord_off * n_src * nblk * 4096 * es;
  double* const diag0_s = L.diag0.ptr + (size_t)ord_off * n_src * p_pad;
  char* const V_s = L.V.ptr + (size_t)ord_off * v_elems_per_ordering(ctx) * es;
  const bool vt = vt_path(ctx);   // V^T by the panel launches' X tiles instead of the strip kernel
  double* const Ppart_s = L.Ppart.ptr + (size_t)ord_off * (m_pad / 64) * p_pad;
  const int32_t* const perms_s = L.perms_cur + (size_t)ord_off * p;
  const bool timed = (st == ctx->lane_stream(L));   // the profiling events live on the lane's main stream
  // a1: small problems take the fused kernel (developer flag 1024 forces the general path, which full_fit,
  // get_factors and debug_factor also need: they read the factors back from the work matrices)
  if (small_path(ctx)) {
    ProfScope ps(timed ? ctx : nullptr, LSSPA_K_SMALL, st);
    SmallArgs sa;
    sa.S[0] = ctx->G.ptr;
    sa.S[1] = ctx->H.ptr;
    sa.s[0] = ctx->g.ptr;
    sa.s[1] = ctx->h.ptr;
    sa.aug[0] = 2.0 * ctx->aug_train + 1.0;
    sa.aug[1] = 2.0 * ctx->y_norm_sq + 1.0;
    sa.ld_src = p_pad;
    sa.fwd_only = (L.perms_fwd_only && per_sample == 2) ? 1 : 0;
    sa.perms = sa.fwd_only ? L.perms_cur + (size_t)(ord_off / 2) * p : perms_s;
    sa.p = p;
    sa.nb = (p + 1 + 15) / 16;
    sa.n_ord = n_ord;
    sa.per_sample = per_sample;
    sa.lifts = L.lifts.ptr + (size_t)s_off * p;
    sa.y_norm_sq = ctx->y_norm_sq;
    sa.piv_tol = 16.0 * (double)p * 2.220446049250313e-16;
    sa.info = ctx->info_d.ptr;
    sa.variant = (ctx->flags & 16384) ? 1 : 0;
    // the batch's own end-to-end check (run_orderings) inside the kernel where the kernel can make it
    sa.r2 = ctx->r2;
    sa.sum_tol = -1.0;
    sa.sum_quiet = 0.0;
    L.sum_checked = false;
    if (ctx->r2_valid && small_p_checks_sum(sa)) {
      sa.sum_tol = sum_check_tol(ctx);
      L.sum_checked = true;
    }
    if (per_sample == 2)
      HIPCHK(hipMemsetAsync(sa.lifts, 0, sizeof(double) * (size_t)(n_ord / 2) * p, st));
    HIPCHK(launch_small_p(sa, st));
    return LSSPA_OK;
  }
  {
    ProfScope ps(timed ? ctx : nullptr, LSSPA_K_GATHER, st);
    GatherArgs ga;
    ga.S[0] = ctx->G.ptr;
    ga.s[0] = ctx->g.ptr;
    ga.aug[0] = 2.0 * ctx->aug_train + 1.0;
    ga.S[1] = ctx->tri ? ctx->H.ptr : nullptr;
    ga.s[1] = ctx->tri ? ctx->h.ptr : nullptr;
    ga.aug[1] = 2.0 * ctx->y_norm_sq + 1.0;
    ga.Sf[0] = ga.Sf[1] = nullptr;
    if (ctx->f32) {   // made by ensure_f32_sources() before the first slice is launched
      ga.Sf[0] = ctx->Gf.ptr;
      ga.Sf[1] = ctx->tri ? ctx->Hf.ptr : nullptr;
    }
    ga.ld_src = p_pad;
    ga.perms = perms_s;
    ga.p = p;
    ga.p_pad = p_pad;
    ga.n_ord = n_ord;
    ga.n_src = n_src;
    ga.A = A_s;
    ga.diag0 = diag0_s;
    ga.f32 = ctx->f32;
    ga.paired = (per_sample == 2) ? 1 : 0;   // stage_and_run lays pairs out back to back
    HIPCHK(launch_gather(ga, st));
  }
  // a pivot below ~p ulps of its feature's own variance is numerically zero (collinear feature)
  const double piv_tol = 16.0 * (double)p * (ctx->f32 ? 1.1920929e-07 : 2.220446049250313e-16);
  {
    ProfScope ps(timed ? ctx : nullptr, LSSPA_K_CHOL_DIAG, st);
    // (the diagonal launch also zeroes the fused lift scan's row flags of its matrices)
    HIPCHK(launch_chol2_diag(A_s, Dinv_s, diag0_s, piv_tol, ctx->info_d.ptr, p_pad, n_mats, ctx->f32, st,
                             fused_scan(ctx) ? L.row_flags.ptr + (size_t)ord_off * n_src : nullptr));
  }
  // panel steps; with vt one more (X tiles only): step Jo also computes block column Jo of V^T
  const int n_panel = p_pad / 128 - 1;
  PanelLift pl;
  pl.mode = 0;
  if (fused_scan(ctx)) {
    pl.flags = L.row_flags.ptr + (size_t)ord_off * n_src;
    pl.run = L.run.ptr + (size_t)ord_off * p_pad;
    pl.Ppart = Ppart_s;
    pl.pstride = (int64_t)(m_pad / 64) * p_pad;
    pl.p = p;
    // the factors themselves are wanted (debug_factor reads V^T back): keep the last panel's stores
    pl.mode = ctx->general_path_once ? 1 : 2;
  }
  for (int Jo = 0; Jo < n_panel + (vt ? 1 : 0); ++Jo) {
    {
      ProfScope ps(timed ? ctx : nullptr, LSSPA_K_CHOL_PANEL, st);
      HIPCHK(launch_chol2_panel(A_s, Dinv_s, diag0_s, piv_tol, ctx->info_d.ptr, p_pad, Jo, n_mats, ctx->f32, st,
                                ctx->flags, round_up(p + 1, 16), vt ? V_s : nullptr, n_ord, pl.mode ? &pl : nullptr));
    }
    // two lanes: the other lane's next batch may start once this one is about half done
    // (the hand-over point scanned at the C3 shape, round 4, eight launches: after launch 0 / 1 / 2 / 3 / 4 / 5 / 6 / 7 ->
    // 6.32 / 6.33 / 6.29 / 6.25 / 6.32 / 6.39 / 6.40 / 6.48 ms per step, one lane 6.44-6.52)
    // (with two half-batches per step on the two lanes, 20 steps: after launch 1 / 2 / 3 / 4 / 5 -> 6.24 / 6.22 / 6.19 /
    // 6.28 / 6.38 ms)
    static const int handover_env = [] { const char* e = getenv("LSSPA_HANDOVER"); return e ? atoi(e) : -1; }();
    const int handover = handover_env >= 0 ? std::min(handover_env, n_panel + (vt ? 1 : 0) - 1)
                                           : std::max(0, (n_panel + (vt ? 1 : 0)) / 2 - 1);
    if (L.mid_armed && timed && Jo == handover) {
      HIPCHK(hipEventRecord(L.ev_mid, st));
      L.mid_valid = true;
      L.mid_armed = false;
    }
  }
  if (!vt) {
    ProfScope ps(timed ? ctx : nullptr, LSSPA_K_STRIP, st);
    StripArgs sa;
    sa.A = A_s;
    sa.Dinv = Dinv_s;
    sa.rhs = ctx->tri ? A_s + (size_t)n_ord * pp2 * es : nullptr;
    sa.Ft = ctx->tri ? nullptr : ctx->Ft.ptr;
    sa.f32 = ctx->f32;
    sa.perms = perms_s;
    sa.V = V_s;
    sa.p = p;
    sa.p_pad = p_pad;
    sa.m_pad = m_pad;
    sa.n_ord = n_ord;
    sa.tri = ctx->tri;
    sa.flags = ctx->flags;
    sa.row_live = round_up(p + 1, 16);
    sa.col_live = ctx->tri ? round_up(p + 1, 16) : round_up(ctx->m, 16);
    HIPCHK(launch_strip(sa, st));
  }
  {
    ProfScope ps(timed ? ctx : nullptr, LSSPA_K_LIFT, st);
    LiftArgs la;
    la.A = A_s;
    la.At = ctx->tri ? A_s + (size_t)n_ord * pp2 * es : nullptr;
    la.f32 = ctx->f32;
    la.ytil = ctx->tri ? nullptr : ctx->ytil.ptr;
    la.V = V_s;
    la.vt = vt ? 1 : 0;
    la.fused = pl.mode ? 1 : 0;
    la.perms = perms_s;
    la.Ppart = Ppart_s;
    la.lifts = L.lifts.ptr + (size_t)s_off * p;
    la.y_norm_sq = ctx->y_norm_sq;
    la.p = p;
    la.p_pad = p_pad;
    la.m_pad = m_pad;
    la.n_ord = n_ord;
    la.per_sample = per_sample;
    la.tri = ctx->tri;
    la.paired = (per_sample == 2) ? 1 : 0;   // stage_and_run builds the second ordering as the reverse of the first
    HIPCHK(launch_lift(la, st));
  }
  return LSSPA_OK;
}

// the problem (or a derived copy of it) changed on the context's stream: lanes order themselves behind this point
int mark_problem(lsspa_ctx* ctx) {
  if (!ctx->ev_problem) HIPCHK(hipEventCreateWithFlags(&ctx->ev_problem, hipEventDisableTiming));
  HIPCHK(hipEventRecord(ctx->ev_problem, ctx->stream));
  ++ctx->problem_epoch;
  return LSSPA_OK;
}

// fp32 mode: the gather reads fp32 copies of the Gram matrices.  Converted once per problem on the context's
// stream, BEFORE a batch forks onto a second stream (a slice on the side stream must not race the conversion).
int ensure_f32_sources(lsspa_ctx* ctx) {
  if (!ctx->f32 || ctx->src_f32_valid) return LSSPA_OK;
  const size_t cnt = (size_t)ctx->p * ctx->p_pad;
  TRY(dev_alloc(ctx, ctx->Gf, cnt));
  HIPCHK(launch_to_f32(ctx->G.ptr, ctx->Gf.ptr, (int64_t)cnt, ctx->stream));
  if (ctx->tri) {
    TRY(dev_alloc(ctx, ctx->Hf, cnt));
    HIPCHK(launch_to_f32(ctx->H.ptr, ctx->Hf.ptr, (int64_t)cnt, ctx->stream));
  }
  ctx->src_f32_valid = true;
  return mark_problem(ctx);
}

// Run gather -> factorisation (with V^T) -> lift for n_ord orderings already resident in perms_d.
// lifts for sample s land in lifts_d[(s_off + s)][p].  (Rounds 1-3 could cut a batch into two slices on two streams of
// ONE lane, developer flag 32, +0.8 %; the two lanes of round 4 do that across batches and better -- removed.)
int run_orderings(lsspa_ctx* ctx, Lane& L, int n_ord, int per_sample, int s_off) {
  const hipStream_t st = ctx->lane_stream(L);
  L.sum_checked = false;
  TRY(run_slice(ctx, L, 0, n_ord, per_sample, s_off, st));
  // the batch's own end-to-end check, once the full model's R^2 is known (lsspa_full_fit): every lift vector sums to it
  // (the register-resident small-problem kernel has made it per ordering already)
  if (ctx->r2_valid && !ctx->general_path_once && !L.sum_checked)
    HIPCHK(launch_sum_check(L.lifts.ptr + (size_t)s_off * ctx->p, n_ord / per_sample, ctx->p, ctx->r2, sum_check_tol(ctx),
                            ctx->info_d.ptr, st));
  return LSSPA_OK;
}

// Stage `count` orderings (with their reverses when per_sample == 2) on lane L and run them.
int stage_and_run(lsspa_ctx* ctx, Lane& L, const int32_t* perms, int n_samples, int per_sample, int s_off) {
  const int p = ctx->p;
  const int n_ord = n_samples * per_sample;
  const hipStream_t st = ctx->lane_stream(L);
  const int turn = L.perms_turn;
  L.perms_turn ^= 1;
  if (L.perms_busy[turn]) {  // the copy that last read this buffer must have run
    HIPCHK(hipEventSynchronize(L.perms_ev[turn]));
    L.perms_busy[turn] = false;
  }
  // ... and, where uploads go by the copy stream, the kernels that read the device slot two batches ago must have
  // finished: the host's lead over the GPU stays at two batches a lane.  (A copy on its own stream is done long before
  // its batch starts, so the wait above no longer holds the host back -- and a host five groups ahead of a one-lane
  // C2 run met a 7 ms stall inside the runtime at its fifth launch: 128 steps 0.075 ms each against 0.029 for 64.)
  if (L.perms_used_valid[turn]) HIPCHK(hipEventSynchronize(L.perms_used[turn]));
  int32_t* hp = L.perms_h[turn];
  // the small-problem kernels read a sample's reverse ordering out of the forward one themselves: half the staging, half
  // the upload (the host's share of a 2048-ordering group at p = 100 was longer than the GPU's)
  L.perms_fwd_only = per_sample == 2 && small_path(ctx);
  if (L.perms_fwd_only || per_sample == 1) {
    std::memcpy(hp, perms, sizeof(int32_t) * (size_t)n_samples * p);   // validated by the caller
  } else {
    for (int s = 0; s < n_samples; ++s) {
      const int32_t* src = perms + (size_t)s * p;
      int32_t* d0 = hp + (size_t)s * 2 * p;
      std::memcpy(d0, src, sizeof(int32_t) * (size_t)p);
      int32_t* d1 = d0 + p;
      for (int j = 0; j < p; ++j) d1[j] = src[p - 1 - j];
    }
  }
  int32_t* dp = L.perms_d.ptr + (size_t)turn * L.cap_ord * p;
  const size_t bytes = sizeof(int32_t) * (size_t)(L.perms_fwd_only ? n_samples : n_ord) * p;
  if (bytes <= ((size_t)256 << 10) || (ctx->n_lanes == 2 && bytes <= ((size_t)1 << 20))) {
    // a small upload rides on the lane's own stream: one copy and one event (the pinned buffer's reuse guard)
    // instead of a hand-off to the copy stream and back -- four stream operations that cost the host more than
    // the copy costs the GPU.  With two lanes that holds up to 1 MB (the other lane's kernels run meanwhile, and copy
    // streams beside two lanes and the context's stream are more streams than hardware queues: C3 6.70 against 5.92 ms
    // a step, measured).  With one lane only up to 256 KB: a group of 4096 orderings at p = 100 waited 36 us for its
    // own 0.8 MB between two 400 us kernels -- on the copy stream it arrives while the previous group runs.
    HIPCHK(hipMemcpyAsync(dp, hp, bytes, hipMemcpyHostToDevice, st));
    HIPCHK(hipEventRecord(L.perms_ev[turn], st));
    L.perms_busy[turn] = true;
    L.perms_cur = dp;
    L.perms_used_valid[turn] = false;    // stream order protects the device slot
    return run_orderings(ctx, L, n_ord, per_sample, s_off);
  }
  if (!L.copy_stream) {
    HIPCHK(hipStreamCreateWithFlags(&L.copy_stream, hipStreamNonBlocking));
    for (int b = 0; b < 2; ++b) HIPCHK(hipEventCreateWithFlags(&L.perms_used[b], hipEventDisableTiming));
  }
  if (L.perms_used_valid[turn]) HIPCHK(hipStreamWaitEvent(L.copy_stream, L.perms_used[turn], 0));
  HIPCHK(hipMemcpyAsync(dp, hp, bytes, hipMemcpyHostToDevice, L.copy_stream));
  HIPCHK(hipEventRecord(L.perms_ev[turn], L.copy_stream));
  L.perms_busy[turn] = true;
  HIPCHK(hipStreamWaitEvent(st, L.perms_ev[turn], 0));
  L.perms_cur = dp;
  const int rc = run_orderings(ctx, L, n_ord, per_sample, s_off);
  if (rc != LSSPA_OK) return rc;
  HIPCHK(hipEventRecord(L.perms_used[turn], st));
  L.perms_used_valid[turn] = true;
  return LSSPA_OK;
}

// Launch a batch on the next lane: every kernel up to the lift vectors.  Nothing here touches the running statistics,
// so a batch can be launched before the previous one has been accumulated (or is ever accumulated: lift_discard).
int lift_launch(lsspa_ctx* ctx, const int32_t* perms, int B, int per, Lane** out) {
  const int p = ctx->p;
  Lane& L = ctx->lanes[ctx->n_lanes == 2 ? ctx->lane_turn : 0];
  if (L.in_flight) return ctx->fail(LSSPA_ERR_STATE, "both lanes hold a launched batch: collect or discard one first");
  TRY(ensure_lane(ctx, L));
  TRY(ensure_f32_sources(ctx));   // on the context's stream ...
  // orderings per launch sequence: everything at once up to ~8 GB of workspace or 4096 orderings
  const size_t per_ord = bytes_per_ordering(ctx);
  const int want = (int)std::max<size_t>(2, std::min<size_t>(4096, ((size_t)8 << 30) / per_ord));
  TRY(ensure_workspace(ctx, L, std::min(B * per, std::max(want, 512)), B));
  const int sub = std::max(1, L.cap_ord / per);
  TRY(ensure_pinned(ctx, L, (size_t)std::min(sub, (int)B) * per * p));
  const hipStream_t st = ctx->lane_stream(L);
  if (ctx->n_lanes == 2) {
    Lane& other = ctx->lanes[1 - ctx->lane_turn];
    // ... which the lane's stream has to see (fp32 sources, a freshly loaded problem, a reset)
    if (L.problem_seen != ctx->problem_epoch) {
      // NOT an event recorded now: the context's stream already holds the statistics of the previous batch, which
      // wait for that batch's kernels -- waiting on "now" would chain this batch behind the whole previous one
      HIPCHK(hipStreamWaitEvent(st, ctx->ev_problem, 0));
      L.problem_seen = ctx->problem_epoch;
    }
    // this lane's previous lift vectors have been read by the statistics
    if (L.consumed_valid) HIPCHK(hipStreamWaitEvent(st, L.ev_consumed, 0));
    // stagger: start when the other lane's batch is half way
    if (other.mid_valid) HIPCHK(hipStreamWaitEvent(st, other.ev_mid, 0));
    L.mid_armed = true;
  }
  for (int s0 = 0; s0 < B; s0 += sub) {
    const int ns = std::min(sub, B - s0);
    TRY(stage_and_run(ctx, L, perms + (size_t)s0 * p, ns, per, s0));
  }
  if (ctx->n_lanes == 2) {
    if (L.mid_armed) {   // one-level path or a single panel: no mid point was met
      HIPCHK(hipEventRecord(L.ev_mid, st));
      L.mid_valid = true;
      L.mid_armed = false;
    }
    HIPCHK(hipEventRecord(L.ev_done, st));
    L.done_valid = true;
    ctx->lane_turn ^= 1;
  }
  L.B = B;
  L.per = per;
  L.taken = 0;
  L.in_flight = true;
  ctx->lane_last = (int)(&L - ctx->lanes);
  if (out) *out = &L;
  return LSSPA_OK;
}

// accumulate: 0 = lift vectors only, 1 = fold into the pending buffer, 2 = fold and merge at once.  2 is a one-GPU
// short cut: refused while a batch is pending (its moments are about another mean) and on a context whose
// communicator spans several ranks (the merge would use per-rank statistics).  Checked BEFORE anything is enqueued.
int check_accumulate(lsspa_ctx* ctx, int accumulate) {
  if (accumulate < 0 || accumulate > 2) return ctx->fail(LSSPA_ERR_ARG, "accumulate must be 0, 1 or 2");
  if (accumulate == 2 && ctx->pend_dirty)
    return ctx->fail(LSSPA_ERR_STATE, "accumulate = 2 (fold and merge at once) with a batch pending: merge it first");
  if (accumulate == 2 && ctx->comm && comm_world(ctx->comm) > 1)
    return ctx->fail(LSSPA_ERR_STATE, "accumulate = 2 (fold and merge at once) on a context with a multi-rank "
                                      "communicator: the moments must be all-reduced before the merge");
  return LSSPA_OK;
}

// Fold `count` samples of a launched batch, from sample `first` on, into the pending statistics (accumulate) and /
// or copy their lift vectors out -- on the context's stream, in order.  The parts of a batch are taken front to
// back; the lane is given back with the last one.
// What becomes of a collected chunk's lift vectors beyond the statistics: appended to the history (the thin-form
// estimator, or the running form's staging until lsspa_error_advance names their sample ids) or, when the caller knows
// the ids already (lsspa_group_collect), folded into the running estimator's sums straight from the lane's buffer.
int keep_lifts(lsspa_ctx* ctx, const double* src, int count, const int64_t* ids /* first_id, stride or null */) {
  if (ctx->run_on && ids) {
    const int n_pad = ((count + KCH - 1) / KCH) * KCH;
    TRY(dev_alloc(ctx, ctx->xi_d, (size_t)ERR_DRAWS * n_pad));
    ProfScope ps(ctx, LSSPA_K_ERROR);
    HIPCHK(launch_error_accumulate(ctx->run_seed, ids[0], ids[1], count, n_pad, ctx->xi_d.ptr, src, ctx->p, 1,
                                   ctx->ldh(), ctx->p, ctx->Dacc.ptr, ctx->sacc.ptr, ctx->stream));
    return LSSPA_OK;
  }
  if (ctx->hist_cap > 0) TRY(hist_append(ctx, src, count, hipMemcpyDeviceToDevice));
  return LSSPA_OK;
}

int lane_taken(lsspa_ctx* ctx, Lane& L, int upto);

int lift_collect(lsspa_ctx* ctx, Lane& L, int first, int count, double* lifts_out, int accumulate,
                 const int64_t* est_ids = nullptr) {
  if (!L.in_flight) return ctx->fail(LSSPA_ERR_STATE, "no launched batch on this lane");
  const int p = ctx->p;
  if (count <= 0) count = L.B - first;
  if (first != L.taken || count < 1 || first + count > L.B)
    return ctx->fail(LSSPA_ERR_ARG, "parts of a launched batch are collected front to back, without gaps");
  TRY(check_accumulate(ctx, accumulate));   // before the stream is touched: a refused call leaves the lane as it was
  const double* src = L.lifts.ptr + (size_t)first * p;
  if (ctx->n_lanes == 2 && first == 0) HIPCHK(hipStreamWaitEvent(ctx->stream, L.ev_done, 0));
  if (accumulate == 2 && stats_small_fusable(count, p)) {
    // single GPU, small p: moments and merge in ONE launch; the advanced mean and n land in the other halves of the
    // pairs, which then become the current ones
    ProfScope ps(ctx, LSSPA_K_STATS);
    HIPCHK(launch_stats_small_fused(src, ctx->mean.ptr, ctx->state_n.ptr, ctx->mean_alt.ptr, ctx->state_alt.ptr,
                                    ctx->M2.ptr, count, p, ctx->stream));
    std::swap(ctx->mean, ctx->mean_alt);
    std::swap(ctx->state_n, ctx->state_alt);
    TRY(keep_lifts(ctx, src, count, est_ids));
  } else if (accumulate) {
    {
      ProfScope ps(ctx, LSSPA_K_STATS);
      const int nz = stats_batch_slices(count, p);
      if (nz > 1) TRY(dev_alloc(ctx, ctx->stat_parts, (size_t)nz * ((size_t)1 + p + (size_t)p * p)));
      HIPCHK(launch_stats_batch(src, ctx->mean.ptr, ctx->pend.ptr, count, p, ctx->pend_dirty ? 1 : 0,
                                nz > 1 ? ctx->stat_parts.ptr : nullptr, ctx->stream));
      ctx->pend_dirty = true;
    }
    TRY(keep_lifts(ctx, src, count, est_ids));
    if (accumulate == 2) TRY(lsspa_stats_merge(ctx));   // the general path: the same effect in its usual launches
  }
  if (lifts_out) {
    HIPCHK(hipMemcpyAsync(lifts_out, src, sizeof(double) * (size_t)count * p, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  return lane_taken(ctx, L, first + count);
}

// Small problems, one rank: the statistics of SEVERAL chunks of a launched batch in one launch (stats_small_multi_kernel),
// chunk by chunk in order -- bit for bit what lift_collect(accumulate = 2) does chunk after chunk.  The chunks follow
// each other in the batch (first[c + 1] = first[c] + count[c]).  snap: keep the running mean and n after every chunk
// (ctx->mean_snap / n_snap) for the checks that belong to them.  The caller has checked fusability (chunks_fusable).
bool chunks_fusable(const lsspa_ctx* ctx, const Lane& L, int n_chunks, const int32_t* first, const int32_t* count) {
  if (ctx->comm || ctx->pend_dirty || n_chunks < 2 || n_chunks > StatsChunks::MAX || !L.in_flight) return false;
  if (first[0] != L.taken) return false;
  int next = first[0];
  for (int c = 0; c < n_chunks; ++c) {
    if (first[c] != next || !stats_small_fusable(count[c], ctx->p)) return false;
    next += count[c];
  }
  return next <= L.B;
}

int collect_chunks_small(lsspa_ctx* ctx, Lane& L, int n_chunks, const int32_t* first, const int32_t* count,
                                bool snap) {
  const int p = ctx->p;
  if (ctx->n_lanes == 2 && first[0] == 0) HIPCHK(hipStreamWaitEvent(ctx->stream, L.ev_done, 0));
  StatsChunks ch;
  ch.n = n_chunks;
  for (int c = 0; c < n_chunks; ++c) {
    ch.first[c] = first[c];
    ch.count[c] = count[c];
  }
  if (snap) {
    TRY(dev_alloc(ctx, ctx->mean_snap, (size_t)StatsChunks::MAX * p));
    TRY(dev_alloc(ctx, ctx->n_snap, (size_t)StatsChunks::MAX));
  }
  {
    ProfScope ps(ctx, LSSPA_K_STATS);
    HIPCHK(launch_stats_small_multi(L.lifts.ptr, ctx->mean.ptr, ctx->state_n.ptr, ctx->mean_alt.ptr, ctx->state_alt.ptr,
                                    ctx->M2.ptr, ch, p, snap ? ctx->mean_snap.ptr : nullptr,
                                    snap ? ctx->n_snap.ptr : nullptr, ctx->stream));
  }
  std::swap(ctx->mean, ctx->mean_alt);
  std::swap(ctx->state_n, ctx->state_alt);
  return LSSPA_OK;
}

// what lift_collect does to the lane when the last sample of its batch has been taken
int lane_taken(lsspa_ctx* ctx, Lane& L, int upto) {
  L.taken = upto;
  if (L.taken == L.B) {
    if (ctx->n_lanes == 2) {
      HIPCHK(hipEventRecord(L.ev_consumed, ctx->stream));
      L.consumed_valid = true;
    }
    L.in_flight = false;
  }
  return LSSPA_OK;
}

bool is_permutation(const int32_t* perm, int p, std::vector<char>& seen) {
  seen.assign(p, 0);
  for (int j = 0; j < p; ++j) {
    const int32_t f = perm[j];
    if (f < 0 || f >= p || seen[f]) return false;
    seen[f] = 1;
  }
  return true;
}

// No exception crosses the C ABI: std::vector / std::string / new inside an entry point turn into a status code
// (an escaping std::bad_alloc would be std::terminate, i.e. an abort of the host process).
int abi_caught(lsspa_ctx* ctx) noexcept {
  const char* what = "C++ exception inside the library";
  int code = LSSPA_ERR_HIP;
  try {
    throw;
  } catch (const std::bad_alloc&) {
    what = "out of host memory";
    code = LSSPA_ERR_NOMEM;
  } catch (const std::exception& e) {
    what = e.what();
  } catch (...) {
  }
  try {
    if (ctx) ctx->err = what;
    else g_create_error = what;
  } catch (...) {
  }
  return code;
}

}  // namespace

// =============================================================================================
extern "C" {

int lsspa_abi_version(void) { return LSSPA_ABI_VERSION; }

const char* lsspa_last_error(const lsspa_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int lsspa_create(int32_t device, lsspa_ctx** out) {
  if (!out) {
    g_create_error = "out pointer is NULL";
    return LSSPA_ERR_ARG;
  }
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n < 1) {
    g_create_error = std::string("no HIP device available: ") + hipGetErrorString(e);
    return LSSPA_ERR_HIP;
  }
  if (device < 0 || device >= n) {
    g_create_error = "device index out of range";
    return LSSPA_ERR_ARG;
  }
  hipDeviceProp_t prop;
  if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) {
    g_create_error = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e);
    return LSSPA_ERR_HIP;
  }
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    g_create_error = std::string("this library is built for gfx950 only, device is ") + prop.gcnArchName;
    return LSSPA_ERR_HIP;
  }
  lsspa_ctx* ctx = new (std::nothrow) lsspa_ctx();
  if (!ctx) {
    g_create_error = "out of host memory";
    return LSSPA_ERR_NOMEM;
  }
  ctx->device = device;
  if (prop.multiProcessorCount > 0) ctx->n_cu = prop.multiProcessorCount;
  if ((e = hipSetDevice(device)) != hipSuccess ||
      (e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) {
    g_create_error = std::string("stream creation: ") + hipGetErrorString(e);
    delete ctx;
    return LSSPA_ERR_HIP;
  }
  ctx->own_stream = true;
  // The two lanes' streams NOW, second and third of the context: the runtime deals its few hardware queues to streams
  // in the order they are made, and a lane that lands on the queue of the context's stream waits behind the statistics
  // that wait for the other lane -- the lanes then run one after the other.  Made on first use they came after the
  // streamed reduction's copy stream: the public call at C3, alone in a process, 38.0 k orderings/s against 41.7 k
  // with this order (round 5, tools/full_run_probe.py; bench.py's own engine never streams a reduction and never saw it).
  for (int k = 0; k < 2; ++k)
    if ((e = hipStreamCreateWithFlags(&ctx->lanes[k].st, hipStreamNonBlocking)) != hipSuccess) {
      g_create_error = std::string("stream creation: ") + hipGetErrorString(e);
      for (int j = 0; j < k; ++j) (void)hipStreamDestroy(ctx->lanes[j].st);
      (void)hipStreamDestroy(ctx->stream);
      delete ctx;
      return LSSPA_ERR_HIP;
    }
  *out = ctx;
  return LSSPA_OK;
}

int lsspa_destroy(lsspa_ctx* ctx) try {
  if (!ctx) return LSSPA_OK;
  (void)hipSetDevice(ctx->device);
  (void)sync_all(ctx);
  for (auto& r : ctx->prof_recs) {
    (void)hipEventDestroy(r.beg);
    (void)hipEventDestroy(r.end);
  }
  dev_free(ctx->G); dev_free(ctx->g); dev_free(ctx->H); dev_free(ctx->h); dev_free(ctx->Ft);
  dev_free(ctx->ytil); dev_free(ctx->scal); dev_free(ctx->info_d);
  dev_free(ctx->mean); dev_free(ctx->M2); dev_free(ctx->pend); dev_free(ctx->state_n);
  dev_free(ctx->mean_alt); dev_free(ctx->state_alt);
  dev_free(ctx->Cred);
  dev_free(ctx->Gf); dev_free(ctx->Hf);
  free_workspace(ctx);
  for (Lane& L : ctx->lanes) {
    if (L.copy_stream) {
      (void)hipStreamDestroy(L.copy_stream);
      for (int b = 0; b < 2; ++b) (void)hipEventDestroy(L.perms_used[b]);
    }
    if (L.ev_done) {
      (void)hipEventDestroy(L.ev_mid);
      (void)hipEventDestroy(L.ev_done);
      (void)hipEventDestroy(L.ev_consumed);
      (void)hipEventDestroy(L.ev_main);
    }
    if (L.st) (void)hipStreamDestroy(L.st);
    for (int b = 0; b < 2; ++b) {
      if (L.perms_h[b]) (void)hipHostFree(L.perms_h[b]);
      if (L.perms_ev[b]) (void)hipEventDestroy(L.perms_ev[b]);
    }
  }
  dev_free(ctx->hist); dev_free(ctx->xi_d); dev_free(ctx->draws); dev_free(ctx->err_out);
  dev_free(ctx->Dacc); dev_free(ctx->sacc);
  if (ctx->res_h) (void)hipHostFree(ctx->res_h);
  for (hipEvent_t& e : ctx->res_ev)
    if (e) (void)hipEventDestroy(e);
  if (ctx->ev_problem) (void)hipEventDestroy(ctx->ev_problem);
  comm_destroy(ctx->comm);
  ctx->comm = nullptr;
  dev_free(ctx->pack); dev_free(ctx->xfer); dev_free(ctx->ibuf); dev_free(ctx->theta_d);
  dev_free(ctx->mean_snap); dev_free(ctx->n_snap);
  dev_free(ctx->grp_P); dev_free(ctx->grp_S); dev_free(ctx->grp_D); dev_free(ctx->grp_s); dev_free(ctx->grp_norms);
  for (int b = 0; b < 2; ++b) {
    dev_free(ctx->red_x[b]);
    dev_free(ctx->red_y[b]);
    if (ctx->red_copied[b]) (void)hipEventDestroy(ctx->red_copied[b]);
    if (ctx->red_consumed[b]) (void)hipEventDestroy(ctx->red_consumed[b]);
  }
  if (ctx->red_cs) (void)hipStreamDestroy(ctx->red_cs);
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_set_stream(lsspa_ctx* ctx, void* hip_stream) try {
  if (!ctx) return LSSPA_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  TRY(sync_all(ctx));
  if (hip_stream) {
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
    ctx->own_stream = false;
  } else if (!ctx->own_stream) {
    HIPCHK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    ctx->own_stream = true;
  }
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_synchronize(lsspa_ctx* ctx) try {
  if (!ctx) return LSSPA_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  return sync_all(ctx);
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_set_lanes(lsspa_ctx* ctx, int32_t n) try {
  if (!ctx || (n != 1 && n != 2)) return LSSPA_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  for (const Lane& L : ctx->lanes)
    if (L.in_flight) return ctx->fail(LSSPA_ERR_STATE, "a launched batch is still to be collected");
  TRY(sync_all(ctx));
  if (n == 1) free_lane_workspace(ctx->lanes[1]);   // the second workspace goes back to the pool
  ctx->n_lanes = n;
  ctx->lane_turn = 0;
  for (Lane& L : ctx->lanes) {
    L.mid_valid = L.done_valid = L.consumed_valid = false;
    L.problem_seen = -1;
  }
  if (n == 2 && !ctx->ev_problem) TRY(mark_problem(ctx));
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

// ---------------------------------------------------------------------------------------------
// C (device, [P1pad][P1pad], P1pad = round_up(p + 1, 128)) = [X | y]^T [X | y] over the n rows given
static int gram_side(lsspa_ctx* ctx, const void* X, const void* y, int64_t n, int64_t ld, int p, int is_f32,
                     double* C_out) {
  const int n_split = gram_default_split(n, p);
  DevBuf<double> slabs, C;
  C.ptr = C_out;
  int rc = dev_alloc(ctx, slabs, gram_workspace_bytes(p, n_split) / sizeof(double));
  if (rc == LSSPA_OK) {
    ProfScope ps(ctx, LSSPA_K_GRAM);
    GramArgs ga;
    ga.X = X;
    ga.y = y;
    ga.n = n;
    ga.ld = ld;
    ga.p = p;
    ga.is_f32 = is_f32;
    ga.slabs = slabs.ptr;
    ga.n_split = n_split;
    ga.C = C.ptr;
    ga.accumulate = 0;
    ga.variant = (ctx->flags >> 16) & 0xff;
    hipError_t e = launch_gram(ga, ctx->stream);
    if (e != hipSuccess) rc = ctx->fail(LSSPA_ERR_HIP, "gram launch", e);
  }
  if (rc == LSSPA_OK) {
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = ctx->fail(LSSPA_ERR_HIP, "gram sync", e);
  }
  dev_free(slabs);
  return rc;
}

// Gram of a HOST-resident [n][ld] matrix, streamed: the rows cross PCIe in chunks through two device
// buffers on a copy stream while the previous chunk's Gram runs on the compute stream; the chunk
// Grams accumulate in C in chunk order.  The copies read the caller's pageable memory through the runtime's ordinary
// path: the caller's pages are never page-locked (rounds 2-3 registered them in place; measured slower -- registering
// 2 x 800 MB cost 14-21 ms to save about 10 ms of copy time -- and it was the one code path that touched memory it did
// not own; removed in round 4).
static int gram_side_streamed(lsspa_ctx* ctx, const void* X, const void* y, int64_t n, int64_t ld, int p,
                              int is_f32, double* C_out) {
  const size_t es = is_f32 ? 4 : 8;
  int64_t rows = (((int64_t)96 << 20) / ((int64_t)p * (int64_t)es) / 16) * 16;   // ~96 MB chunks
  rows = std::max<int64_t>(1024, std::min<int64_t>(rows, ((n + 15) / 16) * 16));
  const int n_split = gram_default_split(rows, p);
  DevBuf<double> slabs, C;
  void* dX[2] = {nullptr, nullptr};
  void* dy[2] = {nullptr, nullptr};
  hipStream_t cs = nullptr;
  hipEvent_t copied[2] = {nullptr, nullptr}, consumed[2] = {nullptr, nullptr};
  C.ptr = C_out;
  int rc = dev_alloc(ctx, slabs, gram_workspace_bytes(p, n_split) / sizeof(double));
  hipError_t e = hipSuccess;
  if (rc == LSSPA_OK) {
    // the staging buffers, the copy stream and the events live with the context: made on first use, found in place by
    // the second side of this reduction and by the next call (four hipMalloc / hipFree pairs of ~100 MB per call before)
    for (int b = 0; b < 2 && rc == LSSPA_OK; ++b) {
      rc = dev_alloc(ctx, ctx->red_x[b], (size_t)rows * p * es);
      if (rc == LSSPA_OK) rc = dev_alloc(ctx, ctx->red_y[b], (size_t)rows * es);
      if (rc == LSSPA_OK && !ctx->red_copied[b]) e = hipEventCreateWithFlags(&ctx->red_copied[b], hipEventDisableTiming);
      if (rc == LSSPA_OK && e == hipSuccess && !ctx->red_consumed[b])
        e = hipEventCreateWithFlags(&ctx->red_consumed[b], hipEventDisableTiming);
      dX[b] = ctx->red_x[b].ptr;
      dy[b] = ctx->red_y[b].ptr;
      copied[b] = ctx->red_copied[b];
      consumed[b] = ctx->red_consumed[b];
    }
    if (rc == LSSPA_OK && e == hipSuccess && !ctx->red_cs) e = hipStreamCreateWithFlags(&ctx->red_cs, hipStreamNonBlocking);
    cs = ctx->red_cs;
    if (rc == LSSPA_OK && e != hipSuccess) rc = ctx->fail(LSSPA_ERR_NOMEM, "streamed reduction buffers", e);
  }
  const auto t_stream0 = std::chrono::steady_clock::now();
  auto seconds_since = [](std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  };
  if (rc == LSSPA_OK) {
    ProfScope ps(ctx, LSSPA_K_GRAM);
    int k = 0;
    for (int64_t r0 = 0; r0 < n && e == hipSuccess; r0 += rows, ++k) {
      const int b = k & 1;
      const int64_t nr = std::min<int64_t>(rows, n - r0);
      if (k >= 2) e = hipStreamWaitEvent(cs, consumed[b], 0);   // the Gram of chunk k-2 has read this buffer
      const char* src = static_cast<const char*>(X) + (size_t)r0 * ld * es;
      if (e == hipSuccess) {
        if (ld == p)
          e = hipMemcpyAsync(dX[b], src, (size_t)nr * p * es, hipMemcpyHostToDevice, cs);
        else
          e = hipMemcpy2DAsync(dX[b], (size_t)p * es, src, (size_t)ld * es, (size_t)p * es, (size_t)nr,
                               hipMemcpyHostToDevice, cs);
      }
      if (e == hipSuccess)
        e = hipMemcpyAsync(dy[b], static_cast<const char*>(y) + (size_t)r0 * es, (size_t)nr * es,
                           hipMemcpyHostToDevice, cs);
      if (e == hipSuccess) e = hipEventRecord(copied[b], cs);
      if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, copied[b], 0);
      if (e == hipSuccess) {
        GramArgs ga;
        ga.X = dX[b];
        ga.y = dy[b];
        ga.n = nr;
        ga.ld = p;
        ga.p = p;
        ga.is_f32 = is_f32;
        ga.slabs = slabs.ptr;
        ga.n_split = n_split;
        ga.C = C.ptr;
        ga.accumulate = k > 0;
        ga.variant = (ctx->flags >> 16) & 0xff;
        e = launch_gram(ga, ctx->stream);
      }
      if (e == hipSuccess) e = hipEventRecord(consumed[b], ctx->stream);
    }
    if (e != hipSuccess) rc = ctx->fail(LSSPA_ERR_HIP, "streamed gram", e);
  }
  // Both streams are idle before the buffers go -- on the error paths too: every copy
  // that reads the caller's pages was enqueued on cs, every kernel that reads dX / dy on the context's stream.
  hipError_t es1 = hipStreamSynchronize(ctx->stream);
  hipError_t es2 = cs ? hipStreamSynchronize(cs) : hipSuccess;
  if (rc == LSSPA_OK && es1 != hipSuccess) rc = ctx->fail(LSSPA_ERR_HIP, "streamed gram sync", es1);
  if (rc == LSSPA_OK && es2 != hipSuccess) rc = ctx->fail(LSSPA_ERR_HIP, "streamed gram copy sync", es2);
  ctx->red_stream_s += seconds_since(t_stream0);
  dev_free(slabs);
  return rc;
}

// The local part of the reduction: unscaled Gram sums of this context's rows into ctx->Cred
// ([2][P1pad][P1pad]: train, test), or -- rect mode, test side -- the transposed test factor.
static int reduce_rows(lsspa_ctx* ctx, const void* X_train, int64_t ld_train, const void* y_train, int64_t N,
                       const void* X_test, int64_t ld_test, const void* y_test, int64_t M, int32_t p,
                       int32_t dtype, int32_t location, int tri) {
  const size_t es = dtype == LSSPA_F32 ? 4 : 8;
  const int is_f32 = dtype == LSSPA_F32;
  const size_t c_elems = (size_t)round_up(p + 1, 128) * round_up(p + 1, 128);
  ctx->red_stream_s = ctx->red_finalize_s = 0.0;
  TRY(dev_alloc(ctx, ctx->Cred, 2 * c_elems));
  HIPCHK(hipMemsetAsync(ctx->Cred.ptr, 0, 2 * c_elems * 8, ctx->stream));
  auto side = [&](const void* X, const void* y, int64_t n, int64_t ld, bool train) -> int {
    if (n == 0) return LSSPA_OK;   // a rank without rows on this side contributes zeros
    double* C = ctx->Cred.ptr + (train ? 0 : c_elems);
    if (location == LSSPA_HOST && (train || tri)) return gram_side_streamed(ctx, X, y, n, ld, p, is_f32, C);
    if (train || tri) return gram_side(ctx, X, y, n, ld, p, is_f32, C);
    // rect mode: F = X_test as it stands (M < p rows), stored transposed; ||y_test||^2 on the side
    const void *dX = X, *dy = y;
    void *tX = nullptr, *ty = nullptr;
    int64_t dld = ld;
    int rc = LSSPA_OK;
    if (location == LSSPA_HOST) {
      hipError_t e = hipMalloc(&tX, (size_t)n * p * es);
      if (e == hipSuccess) e = hipMalloc(&ty, (size_t)n * es);
      if (e != hipSuccess) {
        if (tX) (void)hipFree(tX);
        return ctx->fail(LSSPA_ERR_NOMEM, "device copy of the data", e);
      }
      e = hipMemcpy2DAsync(tX, (size_t)p * es, X, (size_t)ld * es, (size_t)p * es, (size_t)n,
                           hipMemcpyHostToDevice, ctx->stream);
      if (e == hipSuccess) e = hipMemcpyAsync(ty, y, (size_t)n * es, hipMemcpyHostToDevice, ctx->stream);
      if (e != hipSuccess) rc = ctx->fail(LSSPA_ERR_HIP, "H2D copy of the data", e);
      dX = tX;
      dy = ty;
      dld = p;
    }
    if (rc == LSSPA_OK) {
      const int64_t total = (int64_t)p * ctx->m_pad;
      const int grid = (int)std::min<int64_t>((total + 255) / 256, 2048);
      if (is_f32)
        hipLaunchKernelGGL(transpose_test_kernel<float>, dim3(grid), dim3(256), 0, ctx->stream, (const float*)dX,
                           (const float*)dy, n, dld, p, ctx->m_pad, ctx->Ft.ptr, ctx->ytil.ptr);
      else
        hipLaunchKernelGGL(transpose_test_kernel<double>, dim3(grid), dim3(256), 0, ctx->stream,
                           (const double*)dX, (const double*)dy, n, dld, p, ctx->m_pad, ctx->Ft.ptr,
                           ctx->ytil.ptr);
      hipLaunchKernelGGL(sumsq_kernel, dim3(1), dim3(256), 0, ctx->stream, ctx->ytil.ptr, ctx->m_pad,
                         ctx->scal.ptr + 1);
      hipError_t e = hipGetLastError();
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
      if (e != hipSuccess) rc = ctx->fail(LSSPA_ERR_HIP, "test factor kernels", e);
    }
    if (tX) (void)hipFree(tX);
    if (ty) (void)hipFree(ty);
    return rc;
  };
  TRY(side(X_train, y_train, N, ld_train, true));
  TRY(side(X_test, y_test, M, ld_test, false));
  return LSSPA_OK;
}

// G = C_train / N + reg I, g, (tri) H = C_test, h, ||y_test||^2 from the summed Gram buffers
static int reduce_finalize(lsspa_ctx* ctx, int64_t N_total, double reg) {
  const int p = ctx->p;
  const auto t_fin0 = std::chrono::steady_clock::now();
  struct Stop {
    lsspa_ctx* c;
    std::chrono::steady_clock::time_point t0;
    ~Stop() { c->red_finalize_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
  } stop{ctx, t_fin0};
  const size_t c_elems = (size_t)round_up(p + 1, 128) * round_up(p + 1, 128);
  // (not part of the LSSPA_K_GRAM timing class: that class counts the Gram contractions, one launch per side)
  HIPCHK(launch_gram_finalize(ctx->Cred.ptr, p, 1.0 / (double)N_total, reg, ctx->G.ptr, ctx->p_pad, ctx->g.ptr,
                              ctx->scal.ptr + 0, ctx->stream));
  if (ctx->tri)
    HIPCHK(launch_gram_finalize(ctx->Cred.ptr + c_elems, p, 1.0, 0.0, ctx->H.ptr, ctx->p_pad, ctx->h.ptr,
                                ctx->scal.ptr + 1, ctx->stream));
  double sc[2];
  HIPCHK(hipMemcpyAsync(sc, ctx->scal.ptr, sizeof sc, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->aug_train = sc[0];
  ctx->y_norm_sq = sc[1];
  if (!(ctx->y_norm_sq > 0.0)) return ctx->fail(LSSPA_ERR_ARG, "y_test is identically zero (or NaN)");
  TRY(stats_reset(ctx));
  ctx->have_problem = true;
  TRY(mark_problem(ctx));
  ctx->reduce_open = false;
  dev_free(ctx->Cred);
  return LSSPA_OK;
}

int lsspa_reduce(lsspa_ctx* ctx, const void* X_train, int64_t ld_train, const void* y_train, int64_t N,
                 const void* X_test, int64_t ld_test, const void* y_test, int64_t M, int32_t p, double reg,
                 int32_t dtype, int32_t location) try {
  if (!ctx) return LSSPA_ERR_ARG;
  if (!X_train || !y_train || !X_test || !y_test) return ctx->fail(LSSPA_ERR_ARG, "NULL data pointer");
  if (p < 1 || N < p || M < 1 || ld_train < p || ld_test < p)
    return ctx->fail(LSSPA_ERR_ARG, "need 1 <= p <= N, M >= 1, ld >= p");
  if (dtype != LSSPA_F64 && dtype != LSSPA_F32) return ctx->fail(LSSPA_ERR_ARG, "dtype");
  if (location != LSSPA_HOST && location != LSSPA_DEVICE) return ctx->fail(LSSPA_ERR_ARG, "location");
  if (!(reg >= 0.0)) return ctx->fail(LSSPA_ERR_ARG, "reg must be >= 0");
  HIPCHK(hipSetDevice(ctx->device));
  const int tri = (M >= p) ? 1 : 0;
  if (!tri && M > (1 << 20)) return ctx->fail(LSSPA_ERR_ARG, "M too large for rect mode");
  TRY(set_dims(ctx, p, tri ? p : (int)M, tri));
  TRY(reduce_rows(ctx, X_train, ld_train, y_train, N, X_test, ld_test, y_test, M, p, dtype, location, tri));
  return reduce_finalize(ctx, N, reg);
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_reduce_timing(const lsspa_ctx* ctx, double* seconds4) {
  if (!ctx || !seconds4) return LSSPA_ERR_ARG;
  seconds4[0] = 0.0;
  seconds4[1] = ctx->red_stream_s;
  seconds4[2] = 0.0;
  seconds4[3] = ctx->red_finalize_s;
  return LSSPA_OK;
}

int lsspa_reduce_partial(lsspa_ctx* ctx, const void* X_train, int64_t ld_train, const void* y_train,
                         int64_t n_local, const void* X_test, int64_t ld_test, const void* y_test,
                         int64_t m_local, int64_t M_total, int32_t p, int32_t dtype, int32_t location) try {
  if (!ctx) return LSSPA_ERR_ARG;
  if (n_local < 0 || m_local < 0 || M_total < 1 || m_local > M_total || p < 1)
    return ctx->fail(LSSPA_ERR_ARG, "need n_local >= 0, 0 <= m_local <= M_total, p >= 1");
  if ((n_local > 0 && (!X_train || !y_train || ld_train < p)) || (m_local > 0 && (!X_test || !y_test || ld_test < p)))
    return ctx->fail(LSSPA_ERR_ARG, "NULL data pointer or ld < p");
  if (dtype != LSSPA_F64 && dtype != LSSPA_F32) return ctx->fail(LSSPA_ERR_ARG, "dtype");
  if (location != LSSPA_HOST && location != LSSPA_DEVICE) return ctx->fail(LSSPA_ERR_ARG, "location");
  HIPCHK(hipSetDevice(ctx->device));
  const int tri = (M_total >= p) ? 1 : 0;
  if (!tri && m_local != M_total)
    return ctx->fail(LSSPA_ERR_ARG, "M_total < p: the test rows are the factor itself, pass all of them on every rank");
  if (!tri && M_total > (1 << 20)) return ctx->fail(LSSPA_ERR_ARG, "M too large for rect mode");
  TRY(set_dims(ctx, p, tri ? p : (int)M_total, tri));
  TRY(reduce_rows(ctx, X_train, ld_train, y_train, n_local, X_test, ld_test, y_test, m_local, p, dtype, location,
                  tri));
  ctx->reduce_open = true;
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_reduce_buffer(lsspa_ctx* ctx, void** device_ptr, int64_t* count) try {
  if (!ctx || !device_ptr || !count) return LSSPA_ERR_ARG;
  if (!ctx->reduce_open) return ctx->fail(LSSPA_ERR_STATE, "no partial reduction in progress");
  *device_ptr = ctx->Cred.ptr;
  const int64_t P1pad = round_up(ctx->p + 1, 128);
  *count = 2 * P1pad * P1pad;
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_reduce_finish(lsspa_ctx* ctx, int64_t N_total, double reg) try {
  if (!ctx) return LSSPA_ERR_ARG;
  if (!ctx->reduce_open) return ctx->fail(LSSPA_ERR_STATE, "no partial reduction in progress");
  if (N_total < ctx->p) return ctx->fail(LSSPA_ERR_ARG, "need N_total >= p");
  if (!(reg >= 0.0)) return ctx->fail(LSSPA_ERR_ARG, "reg must be >= 0");
  HIPCHK(hipSetDevice(ctx->device));
  return reduce_finalize(ctx, N_total, reg);
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_set_reduced(lsspa_ctx* ctx, int32_t p, const double* G, const double* g, double aug_train,
                      int32_t tri, const double* H, const double* h, int32_t m, const double* Ft,
                      const double* ytil, double y_norm_sq) try {
  if (!ctx) return LSSPA_ERR_ARG;
  if (!G || !g) return ctx->fail(LSSPA_ERR_ARG, "G and g are required");
  if (tri ? (!H || !h) : (!Ft || !ytil)) return ctx->fail(LSSPA_ERR_ARG, "test side pointers missing");
  if (!(y_norm_sq > 0.0) || !(aug_train >= 0.0)) return ctx->fail(LSSPA_ERR_ARG, "norms must be positive");
  HIPCHK(hipSetDevice(ctx->device));
  TRY(set_dims(ctx, p, tri ? p : m, tri ? 1 : 0));
  const size_t pp = ctx->p_pad;
  HIPCHK(hipMemcpy2D(ctx->G.ptr, pp * 8, G, (size_t)p * 8, (size_t)p * 8, p, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(ctx->g.ptr, 0, pp * 8));
  HIPCHK(hipMemcpy(ctx->g.ptr, g, (size_t)p * 8, hipMemcpyHostToDevice));
  if (tri) {
    HIPCHK(hipMemcpy2D(ctx->H.ptr, pp * 8, H, (size_t)p * 8, (size_t)p * 8, p, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(ctx->h.ptr, 0, pp * 8));
    HIPCHK(hipMemcpy(ctx->h.ptr, h, (size_t)p * 8, hipMemcpyHostToDevice));
  } else {
    const size_t mp = ctx->m_pad;
    HIPCHK(hipMemset(ctx->Ft.ptr, 0, (size_t)p * mp * 8));
    HIPCHK(hipMemcpy2D(ctx->Ft.ptr, mp * 8, Ft, (size_t)m * 8, (size_t)m * 8, p, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(ctx->ytil.ptr, 0, mp * 8));
    HIPCHK(hipMemcpy(ctx->ytil.ptr, ytil, (size_t)m * 8, hipMemcpyHostToDevice));
  }
  ctx->aug_train = aug_train;
  ctx->y_norm_sq = y_norm_sq;
  TRY(stats_reset(ctx));
  ctx->have_problem = true;
  TRY(mark_problem(ctx));
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_get_problem(const lsspa_ctx* ctx, int32_t* p, int32_t* m, int32_t* tri, double* y_norm_sq) try {
  if (!ctx || !ctx->have_problem) return LSSPA_ERR_STATE;
  if (p) *p = ctx->p;
  if (m) *m = ctx->m;
  if (tri) *tri = ctx->tri;
  if (y_norm_sq) *y_norm_sq = ctx->y_norm_sq;
  return LSSPA_OK;
} catch (...) {
  return abi_caught(const_cast<lsspa_ctx*>(ctx));
}

int lsspa_get_gram(lsspa_ctx* ctx, double* G, double* g, double* H, double* h) try {
  if (!ctx) return LSSPA_ERR_ARG;
  if (!ctx->have_problem) return ctx->fail(LSSPA_ERR_STATE, "no problem loaded");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  const size_t p = ctx->p, pp = ctx->p_pad;
  if (G) HIPCHK(hipMemcpy2D(G, p * 8, ctx->G.ptr, pp * 8, p * 8, p, hipMemcpyDeviceToHost));
  if (g) HIPCHK(hipMemcpy(g, ctx->g.ptr, p * 8, hipMemcpyDeviceToHost));
  if ((H || h) && !ctx->tri) return ctx->fail(LSSPA_ERR_STATE, "no test Gram in rect mode");
  if (H) HIPCHK(hipMemcpy2D(H, p * 8, ctx->H.ptr, pp * 8, p * 8, p, hipMemcpyDeviceToHost));
  if (h) HIPCHK(hipMemcpy(h, ctx->h.ptr, p * 8, hipMemcpyDeviceToHost));
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

// device -> host copy of `count` work-matrix elements starting at element offset `off`, as doubles
static int fetch_elems(lsspa_ctx* ctx, const char* base, size_t off, size_t count, double* dst) {
  if (!ctx->f32) {
    HIPCHK(hipMemcpy(dst, base + off * 8, count * 8, hipMemcpyDeviceToHost));
    return LSSPA_OK;
  }
  std::vector<float> tmp(count);
  HIPCHK(hipMemcpy(tmp.data(), base + off * 4, count * 4, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < count; ++i) dst[i] = (double)tmp[i];
  return LSSPA_OK;
}

// factor the identity ordering into workspace slot 0 (lifts into lifts_d row 0)
static int factor_identity(lsspa_ctx* ctx, const int32_t* perm_or_null) {
  const int p = ctx->p;
  for (const Lane& L : ctx->lanes)
    if (L.in_flight) return ctx->fail(LSSPA_ERR_STATE, "a launched batch is still to be collected");
  TRY(sync_all(ctx));
  Lane& L = ctx->lanes[0];
  TRY(ensure_lane(ctx, L));
  TRY(ensure_f32_sources(ctx));
  TRY(ensure_workspace(ctx, L, 2, 1));
  TRY(ensure_pinned(ctx, L, (size_t)2 * p));
  std::vector<int32_t> id(p);
  for (int j = 0; j < p; ++j) id[j] = perm_or_null ? perm_or_null[j] : j;
  TRY(sync_all(ctx));                       // the conversion above ran on the context's stream
  ctx->general_path_once = true;            // the callers read L back from the work matrices
  const int rc = stage_and_run(ctx, L, id.data(), 1, 1, 0);
  ctx->general_path_once = false;
  if (rc != LSSPA_OK) return rc;
  TRY(sync_all(ctx));                       // callers read results with blocking copies
  return LSSPA_OK;
}

int lsspa_full_fit(lsspa_ctx* ctx, double* theta, double* r_squared, int32_t* info) try {
  if (!ctx) return LSSPA_ERR_ARG;
  if (!ctx->have_problem) return ctx->fail(LSSPA_ERR_STATE, "no problem loaded");
  HIPCHK(hipSetDevice(ctx->device));
  const int p = ctx->p;
  int32_t saved = 0, now = 0;
  TRY(sync_all(ctx));
  HIPCHK(hipMemcpy(&saved, ctx->info_d.ptr, 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemset(ctx->info_d.ptr, 0, 4));
  TRY(factor_identity(ctx, nullptr));
  HIPCHK(hipMemcpy(&now, ctx->info_d.ptr, 4, hipMemcpyDeviceToHost));
  saved |= now;
  HIPCHK(hipMemcpy(ctx->info_d.ptr, &saved, 4, hipMemcpyHostToDevice));
  if (info) *info = now;
  if (theta) {
    DevBuf<double>& th = ctx->theta_d;          // kept with the context (a hipMalloc / hipFree pair per call otherwise)
    TRY(dev_alloc(ctx, th, (size_t)2 * p));     // theta, and the running right-hand side when p exceeds the LDS
    hipError_t e = launch_backsolve(ctx->lanes[0].A.ptr, th.ptr, p, ctx->p_pad, ctx->f32, ctx->stream, th.ptr + p);
    if (e == hipSuccess) e = hipMemcpyAsync(theta, th.ptr, sizeof(double) * p, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return ctx->fail(LSSPA_ERR_HIP, "backsolve", e);
  }
  if (r_squared) {
    std::vector<double> l(p);
    HIPCHK(hipMemcpy(l.data(), ctx->lanes[0].lifts.ptr, sizeof(double) * p, hipMemcpyDeviceToHost));
    double s = 0.0;
    for (int j = 0; j < p; ++j) s += l[j];
    *r_squared = s;
    // from now on every batch is checked against it (run_orderings); not after a factorisation that broke down
    ctx->r2 = s;
    ctx->r2_valid = (now == 0) && std::isfinite(s);
    ctx->r2_f32 = ctx->f32 != 0;
  }
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_get_factors(lsspa_ctx* ctx, double* R_tr, double* q_tr, double* F_te, double* q_te) try {
  if (!ctx) return LSSPA_ERR_ARG;
  if (!ctx->have_problem) return ctx->fail(LSSPA_ERR_STATE, "no problem loaded");
  HIPCHK(hipSetDevice(ctx->device));
  const int p = ctx->p, m = ctx->m, ppad = ctx->p_pad;
  const size_t mat = (size_t)ppad * ppad;       // chunk-major factor matrices (tiles.h: cm_off)
  TRY(factor_identity(ctx, nullptr));
  std::vector<double> L(mat);
  TRY(fetch_elems(ctx, ctx->lanes[0].A.ptr, 0, mat, L.data()));
  if (R_tr)
    for (int a = 0; a < p; ++a)
      for (int b = 0; b < p; ++b) R_tr[(size_t)a * p + b] = (b >= a) ? L[cm_off(ppad, b, a)] : 0.0;
  if (q_tr)
    for (int a = 0; a < p; ++a) q_tr[a] = L[cm_off(ppad, p, a)];
  if (ctx->tri) {
    if (F_te || q_te) {
      // slot layout of run_orderings: the test matrices follow the n_ord = 1 train matrices
      TRY(fetch_elems(ctx, ctx->lanes[0].A.ptr, mat, mat, L.data()));
      if (F_te)
        for (int a = 0; a < p; ++a)
          for (int b = 0; b < p; ++b) F_te[(size_t)a * p + b] = (b >= a) ? L[cm_off(ppad, b, a)] : 0.0;
      if (q_te)
        for (int a = 0; a < p; ++a) q_te[a] = L[cm_off(ppad, p, a)];
    }
  } else {
    if (F_te) {
      std::vector<double> Ft((size_t)p * ctx->m_pad);
      HIPCHK(hipMemcpy(Ft.data(), ctx->Ft.ptr, Ft.size() * 8, hipMemcpyDeviceToHost));
      for (int r = 0; r < m; ++r)
        for (int f = 0; f < p; ++f) F_te[(size_t)r * p + f] = Ft[(size_t)f * ctx->m_pad + r];
    }
    if (q_te) HIPCHK(hipMemcpy(q_te, ctx->ytil.ptr, sizeof(double) * m, hipMemcpyDeviceToHost));
  }
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

// ---------------------------------------------------------------------------------------------
int lsspa_lift_batch(lsspa_ctx* ctx, const int32_t* perms, int32_t B, int32_t antithetical,
                     double* lifts_out, int32_t accumulate) try {
  if (!ctx) return LSSPA_ERR_ARG;
  if (!ctx->have_problem) return ctx->fail(LSSPA_ERR_STATE, "no problem loaded");
  if (!perms || B < 1) return ctx->fail(LSSPA_ERR_ARG, "perms / B");
  HIPCHK(hipSetDevice(ctx->device));
  const int p = ctx->p;
  const int per = antithetical ? 2 : 1;
  // validate before anything is launched: a repeated index would make a permuted Gram singular
  if (!all_permutations(perms, B, p, ctx->perm_mark))
    return ctx->fail(LSSPA_ERR_ARG, "perms: a row is not a permutation of 0..p-1");
  TRY(check_accumulate(ctx, accumulate));   // nothing is launched for a call that cannot be collected
  Lane* L = nullptr;
  TRY(lift_launch(ctx, perms, B, per, &L));
  const int rc = lift_collect(ctx, *L, 0, B, lifts_out, accumulate);
  if (rc != LSSPA_OK && L->in_flight) {
    // the caller has no ticket for this batch: give the lane back as lsspa_lift_discard would (the message of the
    // failure stays in place)
    if (ctx->n_lanes == 2 && L->taken > 0 && hipEventRecord(L->ev_consumed, ctx->stream) == hipSuccess)
      L->consumed_valid = true;
    L->in_flight = false;
  }
  return rc;
} catch (...) {
  return abi_caught(ctx);
}

// The two halves of lsspa_lift_batch, for callers that want a batch in flight while they decide what to do with
// the previous one (the driver launches batch k+1 before it evaluates the stop rule on batch k).
int lsspa_lift_launch(lsspa_ctx* ctx, const int32_t* perms, int32_t B, int32_t antithetical, int32_t* ticket) try {
  if (!ctx) return LSSPA_ERR_ARG;
  if (!ctx->have_problem) return ctx->fail(LSSPA_ERR_STATE, "no problem loaded");
  if (!perms || B < 1 || !ticket) return ctx->fail(LSSPA_ERR_ARG, "perms / B / ticket");
  HIPCHK(hipSetDevice(ctx->device));
  if (!all_permutations(perms, B, ctx->p, ctx->perm_mark))
    return ctx->fail(LSSPA_ERR_ARG, "perms: a row is not a permutation of 0..p-1");
  Lane* L = nullptr;
  TRY(lift_launch(ctx, perms, B, antithetical ? 2 : 1, &L));
  *ticket = (int32_t)(L - ctx->lanes);
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_lift_collect(lsspa_ctx* ctx, int32_t ticket, int32_t first, int32_t count, double* lifts_out,
                       int32_t accumulate) try {
  if (!ctx || ticket < 0 || ticket > 1 || first < 0) return LSSPA_ERR_ARG;
  if (!ctx->have_problem) return ctx->fail(LSSPA_ERR_STATE, "no problem loaded");
  HIPCHK(hipSetDevice(ctx->device));
  return lift_collect(ctx, ctx->lanes[ticket], first, count, lifts_out, accumulate);
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_lift_collect_chunks(lsspa_ctx* ctx, int32_t ticket, int32_t first, int32_t chunk, int32_t n_chunks,
                              int32_t accumulate) try {
  if (!ctx || ticket < 0 || ticket > 1 || first < 0 || chunk < 1 || n_chunks < 1) return LSSPA_ERR_ARG;
  if (!ctx->have_problem) return ctx->fail(LSSPA_ERR_STATE, "no problem loaded");
  HIPCHK(hipSetDevice(ctx->device));
  Lane& L = ctx->lanes[ticket];
  // everything a part could be refused for, before the first part is taken: a refused call leaves the lane as it was
  if (!L.in_flight) return ctx->fail(LSSPA_ERR_STATE, "no launched batch on this lane");
  if (first != L.taken || (int64_t)first + (int64_t)n_chunks * chunk > L.B)
    return ctx->fail(LSSPA_ERR_ARG, "parts of a launched batch are collected front to back, without gaps, inside the batch");
  TRY(check_accumulate(ctx, accumulate));
  if (accumulate == 2 && n_chunks <= StatsChunks::MAX) {
    int32_t f[StatsChunks::MAX], k[StatsChunks::MAX];
    for (int c = 0; c < n_chunks; ++c) {
      f[c] = first + c * chunk;
      k[c] = chunk;
    }
    if (chunks_fusable(ctx, L, n_chunks, f, k)) {
      TRY(check_accumulate(ctx, 2));
      TRY(collect_chunks_small(ctx, L, n_chunks, f, k, false));
      TRY(keep_lifts(ctx, L.lifts.ptr + (size_t)first * ctx->p, n_chunks * chunk, nullptr));
      return lane_taken(ctx, L, first + n_chunks * chunk);
    }
  }
  for (int c = 0; c < n_chunks; ++c) TRY(lift_collect(ctx, L, first + c * chunk, chunk, nullptr, accumulate));
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_lift_discard(lsspa_ctx* ctx, int32_t ticket) try {
  if (!ctx || ticket < 0 || ticket > 1) return LSSPA_ERR_ARG;
  Lane& L = ctx->lanes[ticket];
  if (!L.in_flight) return ctx->fail(LSSPA_ERR_STATE, "no launched batch on this lane");
  if (ctx->n_lanes == 2 && L.taken > 0) {
    // part of the batch was collected: those statistics kernels, on the context's stream, still read this lane's
    // lift vectors -- the lane's next launch (on its own stream) has to wait for them as after a full collect
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipEventRecord(L.ev_consumed, ctx->stream));
    L.consumed_valid = true;
  }
  L.in_flight = false;    // its kernels run to completion in stream order; nothing reads their output
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_get_info(lsspa_ctx* ctx, int32_t* info) try {
  if (!ctx || !info) return LSSPA_ERR_ARG;
  if (!ctx->have_problem) return ctx->fail(LSSPA_ERR_STATE, "no problem loaded");
  HIPCHK(hipSetDevice(ctx->device));
  TRY(sync_all(ctx));   // the flag is raised by kernels on the lanes' streams
  HIPCHK(hipMemcpyAsync(info, ctx->info_d.ptr, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_get_sum_deviation(lsspa_ctx* ctx, double* max_deviation) try {
  if (!ctx || !max_deviation) return LSSPA_ERR_ARG;
  if (!ctx->have_problem) return ctx->fail(LSSPA_ERR_STATE, "no problem loaded");
  HIPCHK(hipSetDevice(ctx->device));
  TRY(sync_all(ctx));
  HIPCHK(hipMemcpyAsync(max_deviation, ctx->info_d.ptr + 2, 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_get_info_collected(lsspa_ctx* ctx, int32_t* info) try {
  if (!ctx || !info) return LSSPA_ERR_ARG;
  if (!ctx->have_problem) return ctx->fail(LSSPA_ERR_STATE, "no problem loaded");
  HIPCHK(hipSetDevice(ctx->device));
  // the context's stream is ordered behind every batch that was collected (lift_collect waits for the lane's last
  // kernel before the statistics read its lift vectors): the bits of those batches are in place when it has drained
  HIPCHK(hipMemcpyAsync(info, ctx->info_d.ptr, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_stats_reset(lsspa_ctx* ctx) try {
  if (!ctx) return LSSPA_ERR_ARG;
  if (!ctx->have_problem) return ctx->fail(LSSPA_ERR_STATE, "no problem loaded");
  HIPCHK(hipSetDevice(ctx->device));
  return stats_reset(ctx);
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_stats_pending(lsspa_ctx* ctx, void** device_ptr, int64_t* count) try {
  if (!ctx || !device_ptr || !count) return LSSPA_ERR_ARG;
  if (!ctx->have_problem) return ctx->fail(LSSPA_ERR_STATE, "no problem loaded");
  *device_ptr = ctx->pend.ptr;
  *count = (int64_t)1 + ctx->p + (int64_t)ctx->p * ctx->p;
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_stats_merge(lsspa_ctx* ctx) try {
  if (!ctx) return LSSPA_ERR_ARG;
  if (!ctx->have_problem) return ctx->fail(LSSPA_ERR_STATE, "no problem loaded");
  HIPCHK(hipSetDevice(ctx->device));
  bool cleared = false;
  {
    ProfScope ps(ctx, LSSPA_K_STATS);
    HIPCHK(launch_stats_merge(ctx->pend.ptr, ctx->state_n.ptr, ctx->mean.ptr, ctx->M2.ptr, ctx->p, ctx->stream,
                              &cleared));
  }
  // an empty pending buffer (n_b = 0) is what a rank with no samples contributes
  if (!cleared)
    HIPCHK(hipMemsetAsync(ctx->pend.ptr, 0, sizeof(double) * ((size_t)1 + ctx->p + (size_t)ctx->p * ctx->p),
                          ctx->stream));
  ctx->pend_dirty = false;
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_stats_get(lsspa_ctx* ctx, int64_t* n, double* mean, double* cov_biased) try {
  if (!ctx) return LSSPA_ERR_ARG;
  if (!ctx->have_problem) return ctx->fail(LSSPA_ERR_STATE, "no problem loaded");
  HIPCHK(hipSetDevice(ctx->device));
  const size_t p = ctx->p;
  double nd = 0.0;
  HIPCHK(hipMemcpyAsync(&nd, ctx->state_n.ptr, 8, hipMemcpyDeviceToHost, ctx->stream));
  if (mean) HIPCHK(hipMemcpyAsync(mean, ctx->mean.ptr, p * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (cov_biased) HIPCHK(hipMemcpyAsync(cov_biased, ctx->M2.ptr, p * p * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (n) *n = (int64_t)llround(nd);
  if (cov_biased && nd > 0.0) {
    const double inv = 1.0 / nd;
    for (size_t i = 0; i < p * p; ++i) cov_biased[i] *= inv;
  }
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_stats_set(lsspa_ctx* ctx, int64_t n, const double* mean, const double* cov_biased) try {
  if (!ctx || !mean || !cov_biased || n < 0) return LSSPA_ERR_ARG;
  if (!ctx->have_problem) return ctx->fail(LSSPA_ERR_STATE, "no problem loaded");
  HIPCHK(hipSetDevice(ctx->device));
  const size_t p = ctx->p;
  std::vector<double> m2(p * p);
  for (size_t i = 0; i < p * p; ++i) m2[i] = cov_biased[i] * (double)n;
  double st[8] = {(double)n, 0, 0, 0, 0, 0, 0, 0};
  HIPCHK(hipMemcpyAsync(ctx->state_n.ptr, st, sizeof st, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->mean.ptr, mean, p * 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->M2.ptr, m2.data(), p * p * 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->pend.ptr, 0, sizeof(double) * ((size_t)1 + p + p * p), ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));   // m2 / st are stack and heap temporaries
  ctx->pend_dirty = false;
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

// ---------------------------------------------------------------------------------------------
int lsspa_history_enable(lsspa_ctx* ctx, int64_t capacity) try {
  if (!ctx || capacity < 0) return LSSPA_ERR_ARG;
  if (!ctx->have_problem) return ctx->fail(LSSPA_ERR_STATE, "no problem loaded");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->hist_n = 0;
  ctx->hist_cap = 0;
  ctx->run_on = false;
  if (capacity == 0) {
    // switched off; small buffers stay (a kept engine's next call of the same shape finds them in place -- three hipFree
    // calls, each a device synchronisation, were 1.2 ms of a 46 ms call); a long history is given back
    if (ctx->hist.count * sizeof(double) > ((size_t)64 << 20)) dev_free(ctx->hist);
    return LSSPA_OK;
  }
  const int64_t rows = ((capacity + KCH - 1) / KCH) * KCH;   // the draws kernel reads whole 16-row chunks
  TRY(dev_alloc(ctx, ctx->hist, (size_t)rows * ctx->ldh()));
  // zero once: padding columns and the rows of a partial last chunk are read (times a zero of Xi)
  HIPCHK(hipMemsetAsync(ctx->hist.ptr, 0, ctx->hist.count * 8, ctx->stream));
  TRY(dev_alloc(ctx, ctx->draws, (size_t)ERR_DRAWS * ctx->ldh()));
  TRY(dev_alloc(ctx, ctx->err_out, 2 * (size_t)ctx->p + 2 + ERR_DRAWS));   // [quantiles, mean, n], row norms
  HIPCHK(hipMemsetAsync(ctx->draws.ptr, 0, ctx->draws.count * 8, ctx->stream));
  ctx->hist_cap = capacity;
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_history_get(lsspa_ctx* ctx, int64_t* count, double* lifts) try {
  if (!ctx || !count) return LSSPA_ERR_ARG;
  if (!ctx->have_problem) return ctx->fail(LSSPA_ERR_STATE, "no problem loaded");
  *count = ctx->hist_n;
  if (lifts && ctx->hist_n > 0) {
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMemcpy2DAsync(lifts, (size_t)ctx->p * 8, ctx->hist.ptr, (size_t)ctx->ldh() * 8, (size_t)ctx->p * 8,
                            (size_t)ctx->hist_n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_history_append(lsspa_ctx* ctx, const double* lifts, int64_t rows) try {
  if (!ctx || rows < 0 || (rows > 0 && !lifts)) return LSSPA_ERR_ARG;
  if (!ctx->have_problem) return ctx->fail(LSSPA_ERR_STATE, "no problem loaded");
  if (ctx->hist_cap == 0) return ctx->fail(LSSPA_ERR_STATE, "history is not enabled");
  if (rows == 0) return LSSPA_OK;
  HIPCHK(hipSetDevice(ctx->device));
  TRY(hist_append(ctx, lifts, rows, hipMemcpyHostToDevice));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_error_draws(lsspa_ctx* ctx, const double* xi, int64_t ld_xi, int64_t n_local, int64_t n_total) try {
  if (!ctx || n_local < 0 || n_total < n_local || (n_local > 0 && (!xi || ld_xi < n_local)))
    return LSSPA_ERR_ARG;
  if (!ctx->have_problem) return ctx->fail(LSSPA_ERR_STATE, "no problem loaded");
  if (ctx->hist_cap == 0) return ctx->fail(LSSPA_ERR_STATE, "history is not enabled");
  if (n_local != ctx->hist_n)
    return ctx->fail(LSSPA_ERR_ARG, "n_local must equal the number of lift vectors in this context's history");
  if (ctx->pend_dirty) return ctx->fail(LSSPA_ERR_STATE, "merge the pending batch first: the mean is stale");
  HIPCHK(hipSetDevice(ctx->device));
  const int ldh = ctx->ldh();
  if (n_local == 0) {   // a rank without samples contributes zeros to the all-reduce
    HIPCHK(hipMemsetAsync(ctx->draws.ptr, 0, ctx->draws.count * 8, ctx->stream));
    return LSSPA_OK;
  }
  const int64_t n_pad = ((n_local + KCH - 1) / KCH) * KCH;
  TRY(dev_alloc(ctx, ctx->xi_d, (size_t)ERR_DRAWS * n_pad));
  if (n_pad != n_local) HIPCHK(hipMemsetAsync(ctx->xi_d.ptr, 0, (size_t)ERR_DRAWS * n_pad * 8, ctx->stream));
  HIPCHK(hipMemcpy2DAsync(ctx->xi_d.ptr, (size_t)n_pad * 8, xi, (size_t)ld_xi * 8, (size_t)n_local * 8, ERR_DRAWS,
                          hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));   // xi is the caller's (pageable, possibly temporary) buffer: it may
                                               // be released as soon as this call returns (lsspa.h)
  const double nt = (double)n_total;
  const double scale = 1.0 / sqrt(nt * (nt - 1.0));   // inf for n_total = 1, as numpy's division gives
  ProfScope ps(ctx, LSSPA_K_ERROR);
  HIPCHK(launch_error_draws(ctx->xi_d.ptr, (int)n_pad, ctx->hist.ptr, ldh, (int)n_pad, ctx->mean.ptr, scale,
                            ctx->p, ctx->draws.ptr, ldh, ctx->stream));
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_error_buffer(lsspa_ctx* ctx, void** device_ptr, int64_t* count) try {
  if (!ctx || !device_ptr || !count) return LSSPA_ERR_ARG;
  if (ctx->hist_cap == 0) return ctx->fail(LSSPA_ERR_STATE, "history is not enabled");
  *device_ptr = ctx->draws.ptr;
  *count = (int64_t)ERR_DRAWS * ctx->ldh();
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_error_quantiles(lsspa_ctx* ctx, double* feature_errors, double* overall_error) try {
  if (!ctx || !feature_errors || !overall_error) return LSSPA_ERR_ARG;
  if (!ctx->have_problem) return ctx->fail(LSSPA_ERR_STATE, "no problem loaded");
  if (ctx->hist_cap == 0) return ctx->fail(LSSPA_ERR_STATE, "history is not enabled");
  HIPCHK(hipSetDevice(ctx->device));
  const int p = ctx->p;
  {
    ProfScope ps(ctx, LSSPA_K_ERROR);
    HIPCHK(launch_error_quantiles(ctx->draws.ptr, ctx->ldh(), p, ctx->err_out.ptr + 2 * p + 2, ctx->err_out.ptr,
                                  ctx->stream));
  }
  std::vector<double> out((size_t)p + 1);
  HIPCHK(hipMemcpyAsync(out.data(), ctx->err_out.ptr, ((size_t)p + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  std::copy(out.begin(), out.begin() + p, feature_errors);
  *overall_error = out[p];
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

// ---- running form of the device-side estimator -------------------------------------------------------------
int lsspa_error_running_enable(lsspa_ctx* ctx, uint64_t seed) try {
  if (!ctx) return LSSPA_ERR_ARG;
  if (!ctx->have_problem) return ctx->fail(LSSPA_ERR_STATE, "no problem loaded");
  // the staging of the lift vectors is the history buffer (one chunk at a time; it grows on demand)
  TRY(lsspa_history_enable(ctx, 256));
  const size_t ldh = ctx->ldh();
  TRY(dev_alloc(ctx, ctx->Dacc, (size_t)ERR_DRAWS * ldh));
  TRY(dev_alloc(ctx, ctx->sacc, (size_t)ERR_DRAWS));
  HIPCHK(hipMemsetAsync(ctx->Dacc.ptr, 0, ctx->Dacc.count * 8, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->sacc.ptr, 0, ctx->sacc.count * 8, ctx->stream));
  const size_t need = (size_t)lsspa_ctx::RES_SLOTS * (2 * (size_t)ctx->p + 2);
  if (ctx->res_h_count < need) {
    if (ctx->res_h) (void)hipHostFree(ctx->res_h);
    ctx->res_h = nullptr;
    ctx->res_h_count = 0;
    if (hipHostMalloc(reinterpret_cast<void**>(&ctx->res_h), need * 8, hipHostMallocMapped | hipHostMallocCoherent) !=
            hipSuccess ||
        hipHostGetDevicePointer(reinterpret_cast<void**>(&ctx->res_hd), ctx->res_h, 0) != hipSuccess) {
      if (ctx->res_h) (void)hipHostFree(ctx->res_h);
      ctx->res_h = ctx->res_hd = nullptr;
      (void)hipGetLastError();
      return ctx->fail(LSSPA_ERR_NOMEM, "hipHostMalloc (result slots of the estimator)");
    }
    ctx->res_h_count = need;
  }
  for (int k = 0; k < lsspa_ctx::RES_SLOTS; ++k) {
    // (release to system scope: the kernel's stores to the pinned slot are the host's to read once the event is done)
    if (!ctx->res_ev[k])
      HIPCHK(hipEventCreateWithFlags(&ctx->res_ev[k], hipEventDisableTiming | hipEventReleaseToSystem));
    ctx->res_valid[k] = false;
  }
  ctx->run_seed = seed;
  ctx->run_on = true;
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_error_advance(lsspa_ctx* ctx, int64_t first_id, int64_t stride) try {
  if (!ctx || first_id < 0 || stride < 1) return LSSPA_ERR_ARG;
  if (!ctx->have_problem) return ctx->fail(LSSPA_ERR_STATE, "no problem loaded");
  if (!ctx->run_on) return ctx->fail(LSSPA_ERR_STATE, "the running estimator is not enabled");
  const int64_t cnt = ctx->hist_n;
  if (cnt == 0) return LSSPA_OK;
  HIPCHK(hipSetDevice(ctx->device));
  const int64_t n_pad = ((cnt + KCH - 1) / KCH) * KCH;
  TRY(dev_alloc(ctx, ctx->xi_d, (size_t)ERR_DRAWS * n_pad));
  ProfScope ps(ctx, LSSPA_K_ERROR);
  // rows cnt .. n_pad of the staging hold older chunks' (finite) lift vectors: they meet the zero columns of Xi
  HIPCHK(launch_error_accumulate(ctx->run_seed, first_id, stride, (int)cnt, (int)n_pad, ctx->xi_d.ptr, ctx->hist.ptr,
                                 ctx->ldh(), 0, ctx->ldh(), ctx->p, ctx->Dacc.ptr, ctx->sacc.ptr, ctx->stream));
  ctx->hist_n = 0;
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_error_running_draws(lsspa_ctx* ctx, int64_t n_total) try {
  if (!ctx || n_total < 0) return LSSPA_ERR_ARG;
  if (!ctx->have_problem) return ctx->fail(LSSPA_ERR_STATE, "no problem loaded");
  if (!ctx->run_on) return ctx->fail(LSSPA_ERR_STATE, "the running estimator is not enabled");
  if (ctx->hist_n != 0) return ctx->fail(LSSPA_ERR_STATE, "collected samples not yet folded in: call lsspa_error_advance");
  if (ctx->pend_dirty) return ctx->fail(LSSPA_ERR_STATE, "merge the pending batch first: the mean is stale");
  HIPCHK(hipSetDevice(ctx->device));
  const double nt = (double)n_total;
  const double scale = 1.0 / sqrt(nt * (nt - 1.0));   // inf for n_total = 1, as numpy's division gives
  ProfScope ps(ctx, LSSPA_K_ERROR);
  HIPCHK(launch_error_running_draws(ctx->Dacc.ptr, ctx->sacc.ptr, ctx->mean.ptr, scale, ctx->p, ctx->ldh(),
                                    ctx->draws.ptr, ctx->stream));
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

// the shared tail of the two enqueue forms: the quantile kernel has written [quantiles, mean, n] into the pinned slot
// itself; its event tells the host when
// record == false (lsspa_group_collect, all but the group's last check): the slot shares the event of a LATER check of
// the same call -- a host that reads its checks group by group need not pay an event per check
static int finish_check(lsspa_ctx* ctx, int slot, bool record = true) {
  if (record) HIPCHK(hipEventRecord(ctx->res_ev[slot], ctx->stream));
  ctx->res_valid[slot] = true;
  ctx->res_via[slot] = slot;
  return LSSPA_OK;
}

static int quantiles_enqueue(lsspa_ctx* ctx, int32_t slot, bool record);
int lsspa_error_quantiles_enqueue(lsspa_ctx* ctx, int32_t slot) try {
  return quantiles_enqueue(ctx, slot, true);
} catch (...) {
  return abi_caught(ctx);
}

static int quantiles_enqueue(lsspa_ctx* ctx, int32_t slot, bool record) {
  if (!ctx || slot < 0 || slot >= lsspa_ctx::RES_SLOTS) return LSSPA_ERR_ARG;
  if (!ctx->have_problem) return ctx->fail(LSSPA_ERR_STATE, "no problem loaded");
  if (!ctx->run_on) return ctx->fail(LSSPA_ERR_STATE, "the running estimator is not enabled");
  HIPCHK(hipSetDevice(ctx->device));
  const size_t p = ctx->p;
  {
    ProfScope ps(ctx, LSSPA_K_ERROR);
    HIPCHK(launch_error_quantiles(ctx->draws.ptr, ctx->ldh(), (int)p, ctx->err_out.ptr + 2 * p + 2,
                                  ctx->res_hd + (size_t)slot * (2 * p + 2), ctx->stream, ctx->mean.ptr,
                                  ctx->state_n.ptr));
  }
  return finish_check(ctx, slot, record);
}

static int check_enqueue(lsspa_ctx* ctx, int64_t n_total, int32_t slot, bool record);
int lsspa_error_check_enqueue(lsspa_ctx* ctx, int64_t n_total, int32_t slot) try {
  return check_enqueue(ctx, n_total, slot, true);
} catch (...) {
  return abi_caught(ctx);
}

static int check_enqueue(lsspa_ctx* ctx, int64_t n_total, int32_t slot, bool record) {
  if (!ctx || n_total < 0 || slot < 0 || slot >= lsspa_ctx::RES_SLOTS) return LSSPA_ERR_ARG;
  if (!ctx->have_problem) return ctx->fail(LSSPA_ERR_STATE, "no problem loaded");
  if (!ctx->run_on) return ctx->fail(LSSPA_ERR_STATE, "the running estimator is not enabled");
  if (ctx->hist_n != 0) return ctx->fail(LSSPA_ERR_STATE, "collected samples not yet folded in: call lsspa_error_advance");
  if (ctx->pend_dirty) return ctx->fail(LSSPA_ERR_STATE, "merge the pending batch first: the mean is stale");
  HIPCHK(hipSetDevice(ctx->device));
  const size_t p = ctx->p;
  const double nt = (double)n_total;
  const double scale = 1.0 / sqrt(nt * (nt - 1.0));
  {
    ProfScope ps(ctx, LSSPA_K_ERROR);
    HIPCHK(launch_error_quantiles_running(ctx->Dacc.ptr, ctx->sacc.ptr, ctx->mean.ptr, scale, ctx->ldh(), (int)p,
                                          ctx->err_out.ptr + 2 * p + 2, ctx->res_hd + (size_t)slot * (2 * p + 2),
                                          ctx->state_n.ptr, ctx->stream));
  }
  return finish_check(ctx, slot, record);
}

int lsspa_error_result(lsspa_ctx* ctx, int32_t slot, int32_t wait, int32_t* ready, double* feature_errors,
                       double* overall_error, double* mean, int64_t* n) try {
  if (!ctx || slot < 0 || slot >= lsspa_ctx::RES_SLOTS || !ready) return LSSPA_ERR_ARG;
  if (!ctx->run_on || !ctx->res_valid[slot]) return ctx->fail(LSSPA_ERR_STATE, "nothing was enqueued into this slot");
  HIPCHK(hipSetDevice(ctx->device));
  const int via = ctx->res_via[slot];
  if (wait) {
    HIPCHK(hipEventSynchronize(ctx->res_ev[via]));
  } else {
    const hipError_t e = hipEventQuery(ctx->res_ev[via]);
    if (e == hipErrorNotReady) {
      (void)hipGetLastError();
      *ready = 0;
      return LSSPA_OK;
    }
    HIPCHK(e);
  }
  const size_t p = ctx->p;
  const double* src = ctx->res_h + (size_t)slot * (2 * p + 2);
  if (feature_errors) std::copy(src, src + p, feature_errors);
  if (overall_error) *overall_error = src[p];
  if (mean) std::copy(src + p + 1, src + 2 * p + 1, mean);
  if (n) *n = (int64_t)llround(src[2 * p + 1]);
  *ready = 1;
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_error_state_get(lsspa_ctx* ctx, double* D, double* s) try {
  if (!ctx || !D || !s) return LSSPA_ERR_ARG;
  if (!ctx->run_on) return ctx->fail(LSSPA_ERR_STATE, "the running estimator is not enabled");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipMemcpy2DAsync(D, (size_t)ctx->p * 8, ctx->Dacc.ptr, (size_t)ctx->ldh() * 8, (size_t)ctx->p * 8, ERR_DRAWS,
                          hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(s, ctx->sacc.ptr, (size_t)ERR_DRAWS * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_error_state_set(lsspa_ctx* ctx, const double* D, const double* s) try {
  if (!ctx || !D || !s) return LSSPA_ERR_ARG;
  if (!ctx->run_on) return ctx->fail(LSSPA_ERR_STATE, "the running estimator is not enabled");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipMemcpy2DAsync(ctx->Dacc.ptr, (size_t)ctx->ldh() * 8, D, (size_t)ctx->p * 8, (size_t)ctx->p * 8, ERR_DRAWS,
                          hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->sacc.ptr, s, (size_t)ERR_DRAWS * 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_error_xi(lsspa_ctx* ctx, uint64_t seed, int64_t first_id, int64_t stride, int64_t count, double* xi) try {
  if (!ctx || !xi || count < 1 || count > (1 << 20) || first_id < 0 || stride < 1) return LSSPA_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  const int64_t n_pad = ((count + KCH - 1) / KCH) * KCH;
  TRY(dev_alloc(ctx, ctx->xi_d, (size_t)ERR_DRAWS * n_pad));
  HIPCHK(launch_error_xi(seed, first_id, stride, (int)count, (int)n_pad, ctx->xi_d.ptr, ctx->stream));
  HIPCHK(hipMemcpy2DAsync(xi, (size_t)count * 8, ctx->xi_d.ptr, (size_t)n_pad * 8, (size_t)count * 8, ERR_DRAWS,
                          hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

// The per-chunk tail of a launched group in ONE call (the host loop of a small problem is bound by its own call
// overhead otherwise: a chunk of 256 orderings takes the GPU 37 us at p = 100).  For every chunk c, in order:
// collect (fold into the statistics), all-reduce + merge when the context's communicator spans several ranks, fold the
// chunk into the running estimator, and -- where n_after[c] > 0 -- enqueue the check of that sample count into slot[c].
int lsspa_group_collect(lsspa_ctx* ctx, int32_t ticket, int32_t n_chunks, const int32_t* first, const int32_t* count,
                        const int64_t* first_id, int64_t stride, const int64_t* n_after, const int32_t* slot) try {
  if (!ctx || ticket < 0 || ticket > 1 || n_chunks < 1 || !first || !count || !first_id || !n_after || !slot)
    return LSSPA_ERR_ARG;
  if (!ctx->have_problem) return ctx->fail(LSSPA_ERR_STATE, "no problem loaded");
  if (!ctx->run_on) return ctx->fail(LSSPA_ERR_STATE, "the running estimator is not enabled");
  if (ctx->hist_n != 0) return ctx->fail(LSSPA_ERR_STATE, "collected samples not yet folded in: call lsspa_error_advance");
  if (stride < 1) return ctx->fail(LSSPA_ERR_ARG, "stride must be positive");
  for (int c = 0; c < n_chunks; ++c)
    if (first_id[c] < 0 || n_after[c] < 0 || slot[c] < 0 || slot[c] >= lsspa_ctx::RES_SLOTS)
      return ctx->fail(LSSPA_ERR_ARG, "sample ids and counts must be non-negative, slots within range");
  HIPCHK(hipSetDevice(ctx->device));
  {   // the chunks follow each other from the lane's next sample on, inside the batch: checked before the first is taken
    const Lane& Lc = ctx->lanes[ticket];
    if (!Lc.in_flight) return ctx->fail(LSSPA_ERR_STATE, "no launched batch on this lane");
    int64_t next = Lc.taken;
    for (int c = 0; c < n_chunks; ++c) {
      if (count[c] < 0 || (count[c] > 0 && first[c] != next))
        return ctx->fail(LSSPA_ERR_ARG, "parts of a launched batch are collected front to back, without gaps");
      next += count[c];
    }
    if (next > Lc.B) return ctx->fail(LSSPA_ERR_ARG, "the chunks reach beyond the launched batch");
  }
  // a communicator on the context (of one rank or of many): the moments and the draws go through it, as in the
  // separate calls; none: the chunk is folded and merged at once
  const bool several = ctx->comm != nullptr;
  int last_check = -1;
  for (int c = 0; c < n_chunks; ++c)
    if (n_after[c] > 0) last_check = c;
  Lane& Lg = ctx->lanes[ticket];
  if (chunks_fusable(ctx, Lg, n_chunks, first, count)) {
    // small problem, one rank: the group's statistics in one launch; every check reads the mean and n after ITS chunk
    // ... and the estimator likewise: the chunks' products side by side, summed in order with a snapshot per chunk, all
    // the checks' quantiles in one launch (launch_error_group) -- five launches a group instead of five a chunk
    EstChunks ch;
    EstChecks ck;
    ch.n = n_chunks;
    ck.n = 0;
    long long off = 0;
    for (int c = 0; c < n_chunks; ++c) {
      ch.first[c] = first[c];
      ch.count[c] = count[c];
      ch.n_pad[c] = ((count[c] + KCH - 1) / KCH) * KCH;
      ch.xi_off[c] = off;
      off += (long long)ERR_DRAWS * ch.n_pad[c];
      ch.first_id[c] = first_id[c];
      ch.scale[c] = 0.0;
      if (n_after[c] > 0) {
        const double nt = (double)n_after[c];
        ch.scale[c] = 1.0 / sqrt(nt * (nt - 1.0));
        ck.chunk[ck.n] = c;
        ck.slot[ck.n] = slot[c];
        ck.scale[ck.n] = ch.scale[c];
        ++ck.n;
      }
    }
    const size_t ld = ctx->ldh();
    TRY(dev_alloc(ctx, ctx->xi_d, (size_t)off));
    TRY(dev_alloc(ctx, ctx->grp_P, (size_t)n_chunks * ERR_DRAWS * ld));
    TRY(dev_alloc(ctx, ctx->grp_D, (size_t)n_chunks * ERR_DRAWS * ld));
    TRY(dev_alloc(ctx, ctx->grp_S, (size_t)n_chunks * ERR_DRAWS));
    TRY(dev_alloc(ctx, ctx->grp_s, (size_t)n_chunks * ERR_DRAWS));
    TRY(dev_alloc(ctx, ctx->grp_norms, (size_t)n_chunks * ERR_DRAWS));
    // (every buffer is there before the first launch: a refused allocation leaves statistics and estimator in step)
    TRY(collect_chunks_small(ctx, Lg, n_chunks, first, count, true));
    {
      ProfScope ps(ctx, LSSPA_K_ERROR);
      HIPCHK(launch_error_group(ctx->run_seed, stride, ch, ck, ctx->xi_d.ptr, Lg.lifts.ptr, ctx->p, (int)ld,
                                ctx->grp_P.ptr, ctx->grp_S.ptr, ctx->Dacc.ptr, ctx->sacc.ptr, ctx->grp_D.ptr,
                                ctx->grp_s.ptr, ctx->mean_snap.ptr, ctx->n_snap.ptr, ctx->grp_norms.ptr, ctx->res_hd,
                                ctx->stream));
    }
    for (int c = 0; c < n_chunks; ++c)
      if (n_after[c] > 0) TRY(finish_check(ctx, slot[c], c == last_check));
    TRY(lane_taken(ctx, Lg, first[n_chunks - 1] + count[n_chunks - 1]));
    for (int c = 0; c < n_chunks; ++c)
      if (n_after[c] > 0) ctx->res_via[slot[c]] = slot[last_check];
    return LSSPA_OK;
  }
  for (int c = 0; c < n_chunks; ++c) {
    if (count[c] > 0) {
      const int64_t ids[2] = {first_id[c], stride};
      TRY(lift_collect(ctx, ctx->lanes[ticket], first[c], count[c], nullptr, several ? 1 : 2, ids));
    }
    if (several) {
      TRY(lsspa_stats_allreduce(ctx));
      TRY(lsspa_stats_merge(ctx));
    }
    if (n_after[c] > 0) {
      const bool record = (c == last_check);
      if (several) {
        TRY(lsspa_error_running_draws(ctx, n_after[c]));
        TRY(lsspa_error_allreduce(ctx));
        TRY(quantiles_enqueue(ctx, slot[c], record));
      } else {
        TRY(check_enqueue(ctx, n_after[c], slot[c], record));
      }
    }
  }
  // the earlier checks of the call are done when its last one is: they are read through that slot's event
  for (int c = 0; c < n_chunks; ++c)
    if (n_after[c] > 0) ctx->res_via[slot[c]] = slot[last_check];
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

// ---------------------------------------------------------------------------------------------
// Collectives: RCCL on the context's stream, so kernels -> all-reduce -> merge need no host synchronisation.
int lsspa_comm_unique_id(uint8_t* id128) try {
  std::string err;
  if (comm_unique_id(id128, err) != 0) {
    g_create_error = err;
    return LSSPA_ERR_HIP;
  }
  return LSSPA_OK;
} catch (...) {
  return abi_caught(nullptr);
}

int lsspa_comm_init(lsspa_ctx* ctx, const uint8_t* id128, int32_t rank, int32_t world) try {
  if (!ctx) return LSSPA_ERR_ARG;
  if (!id128 || world < 1 || rank < 0 || rank >= world) return ctx->fail(LSSPA_ERR_ARG, "need an id and 0 <= rank < world");
  if (ctx->comm) return ctx->fail(LSSPA_ERR_STATE, "this context already has a communicator");
  HIPCHK(hipSetDevice(ctx->device));
  std::string err;
  if (comm_create(id128, rank, world, ctx->device, &ctx->comm, err) != 0) return ctx->fail(LSSPA_ERR_HIP, err.c_str());
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_comm_destroy(lsspa_ctx* ctx) try {
  if (!ctx) return LSSPA_ERR_ARG;
  if (!ctx->comm) return LSSPA_OK;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  comm_destroy(ctx->comm);
  ctx->comm = nullptr;
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_comm_info(const lsspa_ctx* ctx, int32_t* rank, int32_t* world) try {
  if (!ctx) return LSSPA_ERR_ARG;
  if (rank) *rank = comm_rank(ctx->comm);
  if (world) *world = comm_world(ctx->comm);
  return ctx->comm ? LSSPA_OK : LSSPA_ERR_STATE;
} catch (...) {
  return abi_caught(const_cast<lsspa_ctx*>(ctx));
}

static int allreduce_buffer(lsspa_ctx* ctx, double* buf, size_t count) {
  std::string err;
  if (comm_allreduce_f64(ctx->comm, buf, count, ctx->stream, err) != 0) return ctx->fail(LSSPA_ERR_HIP, err.c_str());
  return LSSPA_OK;
}

int lsspa_stats_allreduce(lsspa_ctx* ctx) try {
  if (!ctx) return LSSPA_ERR_ARG;
  if (!ctx->have_problem) return ctx->fail(LSSPA_ERR_STATE, "no problem loaded");
  if (!ctx->comm) return ctx->fail(LSSPA_ERR_STATE, "no communicator: call lsspa_comm_init first");
  HIPCHK(hipSetDevice(ctx->device));
  const int p = ctx->p;
  ProfScope ps(ctx, LSSPA_K_COMM);
  if (p < ctx->pack_from_p) return allreduce_buffer(ctx, ctx->pend.ptr, (size_t)1 + p + (size_t)p * p);
  const size_t cnt = (size_t)stats_packed_count(p);
  TRY(dev_alloc(ctx, ctx->pack, cnt));
  HIPCHK(launch_stats_pack(ctx->pend.ptr, ctx->pack.ptr, p, ctx->stream));
  TRY(allreduce_buffer(ctx, ctx->pack.ptr, cnt));
  HIPCHK(launch_stats_unpack(ctx->pack.ptr, ctx->pend.ptr, p, ctx->stream));
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_reduce_allreduce(lsspa_ctx* ctx) try {
  if (!ctx) return LSSPA_ERR_ARG;
  if (!ctx->reduce_open) return ctx->fail(LSSPA_ERR_STATE, "no partial reduction in progress");
  if (!ctx->comm) return ctx->fail(LSSPA_ERR_STATE, "no communicator: call lsspa_comm_init first");
  HIPCHK(hipSetDevice(ctx->device));
  const size_t P1pad = (size_t)round_up(ctx->p + 1, 128);
  ProfScope ps(ctx, LSSPA_K_COMM);
  return allreduce_buffer(ctx, ctx->Cred.ptr, 2 * P1pad * P1pad);
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_error_allreduce(lsspa_ctx* ctx) try {
  if (!ctx) return LSSPA_ERR_ARG;
  if (ctx->hist_cap == 0) return ctx->fail(LSSPA_ERR_STATE, "history is not enabled");
  if (!ctx->comm) return ctx->fail(LSSPA_ERR_STATE, "no communicator: call lsspa_comm_init first");
  HIPCHK(hipSetDevice(ctx->device));
  ProfScope ps(ctx, LSSPA_K_COMM);
  return allreduce_buffer(ctx, ctx->draws.ptr, (size_t)ERR_DRAWS * ctx->ldh());
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_comm_sum_i64(lsspa_ctx* ctx, int64_t* values, int32_t count) try {
  if (!ctx || !values || count < 1) return LSSPA_ERR_ARG;
  if (!ctx->comm) return ctx->fail(LSSPA_ERR_STATE, "no communicator: call lsspa_comm_init first");
  HIPCHK(hipSetDevice(ctx->device));
  TRY(dev_alloc(ctx, ctx->ibuf, (size_t)count));
  HIPCHK(hipMemcpyAsync(ctx->ibuf.ptr, values, (size_t)count * 8, hipMemcpyHostToDevice, ctx->stream));
  std::string err;
  if (comm_allreduce_i64(ctx->comm, ctx->ibuf.ptr, (size_t)count, ctx->stream, err) != 0)
    return ctx->fail(LSSPA_ERR_HIP, err.c_str());
  HIPCHK(hipMemcpyAsync(values, ctx->ibuf.ptr, (size_t)count * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_comm_allgather(lsspa_ctx* ctx, const double* send, int64_t count, double* recv) try {
  if (!ctx || count < 0 || (count > 0 && (!send || !recv))) return LSSPA_ERR_ARG;
  if (!ctx->comm) return ctx->fail(LSSPA_ERR_STATE, "no communicator: call lsspa_comm_init first");
  if (count == 0) return LSSPA_OK;
  HIPCHK(hipSetDevice(ctx->device));
  const size_t world = (size_t)comm_world(ctx->comm), cnt = (size_t)count;
  TRY(dev_alloc(ctx, ctx->xfer, (world + 1) * cnt));
  double* d_send = ctx->xfer.ptr;
  double* d_recv = ctx->xfer.ptr + cnt;
  HIPCHK(hipMemcpyAsync(d_send, send, cnt * 8, hipMemcpyHostToDevice, ctx->stream));
  std::string err;
  if (comm_allgather_f64(ctx->comm, d_send, d_recv, cnt, ctx->stream, err) != 0)
    return ctx->fail(LSSPA_ERR_HIP, err.c_str());
  HIPCHK(hipMemcpyAsync(recv, d_recv, world * cnt * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

// ---------------------------------------------------------------------------------------------
int lsspa_profile_enable(lsspa_ctx* ctx, int32_t on) try {
  if (!ctx) return LSSPA_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  TRY(prof_collect(ctx));
  ctx->prof_on = on != 0;
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_profile_get(lsspa_ctx* ctx, int32_t k, double* total_ms, int64_t* launches) try {
  if (!ctx || k < 0 || k >= LSSPA_K_COUNT) return LSSPA_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  TRY(prof_collect(ctx));
  if (total_ms) *total_ms = ctx->prof_ms[k];
  if (launches) *launches = ctx->prof_n[k];
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_profile_reset(lsspa_ctx* ctx) try {
  if (!ctx) return LSSPA_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  TRY(prof_collect(ctx));
  for (int k = 0; k < LSSPA_K_COUNT; ++k) {
    ctx->prof_ms[k] = 0.0;
    ctx->prof_n[k] = 0;
  }
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

// ---------------------------------------------------------------------------------------------
int lsspa_set_flags(lsspa_ctx* ctx, int32_t flags) try {
  if (!ctx) return LSSPA_ERR_ARG;
  ctx->flags = flags;
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_debug_set_r2(lsspa_ctx* ctx, double r2) try {
  if (!ctx) return LSSPA_ERR_ARG;
  if (!ctx->r2_valid) return ctx->fail(LSSPA_ERR_STATE, "no R^2 yet: call lsspa_full_fit first");
  ctx->r2 = r2;
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_host_argsort_rows(const double* keys, int64_t B, int32_t p, int32_t* out, uint8_t* redo, int32_t threads,
                            int64_t* n_redo) try {
  if (!keys || !out || !redo || B < 1 || p < 1 || !n_redo) return LSSPA_ERR_ARG;
  *n_redo = argsort_rows_host(keys, B, p, out, redo, threads);
  return LSSPA_OK;
} catch (...) {
  return abi_caught(nullptr);
}

int lsspa_sampler_create(int32_t p, int32_t bits, const uint64_t* sv, const uint64_t* q0, double scale, int64_t limit,
                         int32_t block, int64_t ahead, int64_t ahead_unasked, int32_t threads, int32_t rank,
                         int32_t world, void** out) try {
  if (!out) return LSSPA_ERR_ARG;
  *out = nullptr;
  if (p < 1 || bits < 1 || bits > 64 || !sv || !q0 || !(scale > 0.0) || limit < 0 || block < 1 || ahead < 1 ||
      ahead_unasked < 1 || threads < 1 || world < 1 || rank < 0 || rank >= world) {
    g_create_error = "lsspa_sampler_create: bad argument";
    return LSSPA_ERR_ARG;
  }
  *out = sobol_sampler_create(p, bits, sv, q0, scale, limit, block, ahead, ahead_unasked, threads, rank, world);
  return LSSPA_OK;
} catch (...) {
  return abi_caught(nullptr);
}

int lsspa_sampler_take(void* sampler, int64_t count, int32_t* out, int64_t cap, int64_t* n_taken, int64_t* n_own,
                       int64_t* redo_pos, int64_t* redo_id, int64_t* n_redo) try {
  if (!sampler || count < 0 || !out || cap < 0 || !n_taken || !n_own || !redo_pos || !redo_id || !n_redo)
    return LSSPA_ERR_ARG;
  const char* err = nullptr;
  const int rc = sobol_sampler_take(static_cast<SobolSampler*>(sampler), count, out, cap, n_taken, n_own, redo_pos,
                                    redo_id, n_redo, &err);
  if (rc == 1) {
    g_create_error = "lsspa_sampler_take: the output buffer is too small for this rank's share";
    return LSSPA_ERR_ARG;
  }
  if (rc == 2) {
    g_create_error = std::string("lsspa_sampler: ") + (err ? err : "the producer failed");
    return LSSPA_ERR_STATE;
  }
  return LSSPA_OK;
} catch (...) {
  return abi_caught(nullptr);
}

int lsspa_sampler_destroy(void* sampler) try {
  sobol_sampler_destroy(static_cast<SobolSampler*>(sampler));
  return LSSPA_OK;
} catch (...) {
  return abi_caught(nullptr);
}

int lsspa_debug_check_perms(const int32_t* perms, int32_t B, int32_t p, int32_t plain) try {
  if (!perms || B < 1 || p < 1) return 0;
  std::vector<int32_t> mark;
  return (plain ? all_permutations_plain(perms, B, p, mark) : all_permutations(perms, B, p, mark)) ? 1 : 0;
} catch (...) {
  return 0;
}

int lsspa_debug_pack_from(lsspa_ctx* ctx, int32_t p_min) try {
  if (!ctx || p_min < 1) return LSSPA_ERR_ARG;
  ctx->pack_from_p = p_min;
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_debug_fail_alloc(lsspa_ctx* ctx, int32_t nth) try {
  if (!ctx || nth < 0) return LSSPA_ERR_ARG;
  ctx->fail_alloc_in = nth;
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_set_precision(lsspa_ctx* ctx, int32_t dtype) try {
  if (!ctx) return LSSPA_ERR_ARG;
  if (dtype != LSSPA_F64 && dtype != LSSPA_F32) return ctx->fail(LSSPA_ERR_ARG, "dtype");
  HIPCHK(hipSetDevice(ctx->device));
  if ((dtype == LSSPA_F32) != (ctx->f32 != 0)) {
    for (const Lane& L : ctx->lanes)
      if (L.in_flight) return ctx->fail(LSSPA_ERR_STATE, "a launched batch is still to be collected");
    TRY(sync_all(ctx));     // kernels on the lanes' streams still use the workspace that is about to go
    ctx->f32 = dtype == LSSPA_F32;
    free_workspace(ctx);   // re-created for the new element size on demand
  }
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_mfma_probe(lsspa_ctx* ctx, const double* A16x4, const double* B4x16, double* D16x16, int32_t dtype) try {
  if (!ctx || !A16x4 || !B4x16 || !D16x16) return LSSPA_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  DevBuf<double> buf;
  TRY(dev_alloc(ctx, buf, 64 + 64 + 256));
  hipError_t e = hipMemcpy(buf.ptr, A16x4, 64 * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(buf.ptr + 64, B4x16, 64 * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = launch_mfma_probe(buf.ptr, buf.ptr + 64, buf.ptr + 128, dtype == LSSPA_F32, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess) e = hipMemcpy(D16x16, buf.ptr + 128, 256 * 8, hipMemcpyDeviceToHost);
  dev_free(buf);
  if (e != hipSuccess) return ctx->fail(LSSPA_ERR_HIP, "mfma probe", e);
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

int lsspa_debug_factor(lsspa_ctx* ctx, const int32_t* perm, double* L, double* Lt, double* V, int32_t* p_pad,
                       int32_t* m_pad, int32_t* v_rows) try {
  if (!ctx) return LSSPA_ERR_ARG;
  if (!ctx->have_problem) return ctx->fail(LSSPA_ERR_STATE, "no problem loaded");
  HIPCHK(hipSetDevice(ctx->device));
  const size_t pp = ctx->p_pad, n_iblk = (ctx->p + NB - 1) / NB;
  if (p_pad) *p_pad = ctx->p_pad;
  if (m_pad) *m_pad = ctx->m_pad;
  if (v_rows) *v_rows = (int32_t)(n_iblk * NB);
  if (!perm) return LSSPA_OK;  // size query
  std::vector<char> seen;
  if (!is_permutation(perm, ctx->p, seen)) return ctx->fail(LSSPA_ERR_ARG, "perm is not a permutation");
  TRY(factor_identity(ctx, perm));
  const size_t ldv = (size_t)ldv_of(ctx->m_pad), mp = ctx->m_pad;
  // the device matrices are chunk-major; hand them out dense row-major
  auto unpack = [&](size_t mat_index, double* dst) -> int {
    std::vector<double> tmp(pp * pp);
    TRY(fetch_elems(ctx, ctx->lanes[0].A.ptr, mat_index * pp * pp, pp * pp, tmp.data()));
    for (size_t r = 0; r < pp; ++r)
      for (size_t c = 0; c < pp; ++c) dst[r * pp + c] = tmp[cm_off((int)pp, (int)r, (int)c)];
    return LSSPA_OK;
  };
  if (L) TRY(unpack(0, L));
  if (Lt && ctx->tri) TRY(unpack(1, Lt));
  if (V && vt_path(ctx)) {
    // the panel launches leave V^T (chunk-major, upper block triangle written; rows at or beyond p belong to no feature)
    const size_t rows = n_iblk * NB;
    std::vector<double> tmp(pp * pp);
    TRY(fetch_elems(ctx, ctx->lanes[0].V.ptr, 0, pp * pp, tmp.data()));
    for (size_t r = 0; r < rows; ++r)
      for (size_t c = 0; c < mp; ++c)
        V[r * mp + c] = (c / 128 > r / 128 || c >= (size_t)ctx->p) ? 0.0 : tmp[cm_off((int)pp, (int)c, (int)r)];
  } else if (V) {
    const size_t rows = n_iblk * NB;
    std::vector<double> tmp(rows * ldv);
    TRY(fetch_elems(ctx, ctx->lanes[0].V.ptr, 0, rows * ldv, tmp.data()));
    // tri: entries above a 128-column strip's first diagonal block are structurally zero and never written on the
    // device (strip2_kernel)
    for (size_t r = 0; r < rows; ++r)
      for (size_t c = 0; c < mp; ++c)
        V[r * mp + c] = (ctx->tri && r < (c / 128) * 128) ? 0.0 : tmp[r * ldv + c];
  }
  return LSSPA_OK;
} catch (...) {
  return abi_caught(ctx);
}

}  // extern "C"
