/* lsspa.h -- C ABI of the MI355X-native LS-SPA engine (liblsspa_hip.so).
 *
 * The reference (cvxgrp/ls-spa @ v2) has no FFI layer: its boundary is the Python
 * package surface of ls_spa/ls_spa.py.  Each entry point below names the reference
 * function(s) it stands in for; the Python host package ls-spa_amd/ls_spa binds them
 * with ctypes and re-creates the reference's own signatures on top.
 *
 * Conventions: extern "C"; plain pointers and sizes only; int status return
 * (0 = LSSPA_OK); no C++ exception crosses the boundary; the caller owns every host
 * buffer, the library owns every device buffer behind the opaque context; one context
 * per GPU; a context is not thread-safe.  All matrices are row-major.  After a
 * non-zero status lsspa_last_error() returns a description.
 */
#ifndef LSSPA_H
#define LSSPA_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSSPA_ABI_VERSION 1

#define LSSPA_OK 0
#define LSSPA_ERR_ARG 1     /* bad argument / shape */
#define LSSPA_ERR_HIP 2     /* a HIP runtime call failed */
#define LSSPA_ERR_STATE 3   /* call sequence error (no problem loaded, ...) */
#define LSSPA_ERR_NOMEM 4   /* device or host allocation failed */

#define LSSPA_F64 0
#define LSSPA_F32 1
#define LSSPA_HOST 0
#define LSSPA_DEVICE 1

/* info bit flags (lsspa_get_info) */
#define LSSPA_INFO_NOT_PD 1 /* a non-positive pivot was met in a Cholesky step */
#define LSSPA_INFO_SCAN_WAIT 4 /* an X tile gave up waiting for row p of its panel (fused lift scan): results invalid */
#define LSSPA_INFO_SUM 8 /* a sample's lifts did not sum to the R^2 of the full model (to 1e-9, fp32 work: 1e-4): results invalid */

typedef struct lsspa_ctx lsspa_ctx;

int lsspa_abi_version(void);
/* NULL context: error of the last failed lsspa_create on this thread */
const char* lsspa_last_error(const lsspa_ctx* ctx);

int lsspa_create(int32_t device, lsspa_ctx** out);
int lsspa_destroy(lsspa_ctx* ctx);
/* run on a caller-provided hipStream_t (e.g. torch's current stream); NULL = library-owned */
int lsspa_set_stream(lsspa_ctx* ctx, void* hip_stream);
int lsspa_synchronize(lsspa_ctx* ctx);

/* a1 -- replaces reduce_data (ls_spa/ls_spa.py:290-318) and the |y_test|^2 of :180.
 * Forms G = X_tr^T X_tr / N + reg I, g = X_tr^T y_tr / N and, when M >= p, H = X_te^T X_te,
 * h = X_te^T y_te by one MFMA Gram pass each; when M < p the test rows themselves are
 * kept (transposed) as the test factor.  X pointers: row-major [rows][ld]; dtype applies to
 * X and y alike; location says whether the four pointers are host or device memory.
 * p <= 32767 (32-bit element counts of one p x p work matrix); a larger p is refused here with LSSPA_ERR_ARG and a
 * message naming it (the reference has no limit, ls_spa/ls_spa.py:163).  Up to p = 13567 the gather stages a whole
 * source row and the ordering in the 160 KB of LDS of a CU; beyond that a segmented (slower) gather takes over, so that
 * memory, not LDS, bounds the feature count.
 * Host arrays are never written and stay the caller's: they are read during the call only, through ordinary copies
 * (nothing of the caller's is ever page-locked: measured, the runtime's own pageable path is faster than registering
 * the arrays first).
 * No C++ exception leaves any entry point of this header: host allocation failure is LSSPA_ERR_NOMEM. */
int lsspa_reduce(lsspa_ctx* ctx, const void* X_train, int64_t ld_train, const void* y_train, int64_t N,
                 const void* X_test, int64_t ld_test, const void* y_test, int64_t M, int32_t p, double reg,
                 int32_t dtype, int32_t location);

/* Host seconds the last reduction from HOST arrays spent in its parts: [1] the streamed copies and Gram kernels (to
 * the last one's completion), [3] finalize (scaling, statistics reset, sync); [0] and [2] (page-locking and
 * un-locking of the caller's arrays, rounds 2-3) are always 0 now.  bench.py's e2e_breakdown. */
int lsspa_reduce_timing(const lsspa_ctx* ctx, double* seconds4);

/* a1, rows spread over several GPUs: every rank reduces the rows it holds,
 *   lsspa_reduce_partial : unscaled Gram sums of n_local training rows and m_local test rows.  M_total is the
 *                          test-row count over all ranks; if M_total < p the test rows ARE the test factor and
 *                          every rank has to pass all of them (m_local == M_total).  n_local / m_local may be 0.
 *   lsspa_reduce_buffer  : device pointer / element count (fp64) of the sums -- the all-reduce(SUM) target
 *   lsspa_reduce_finish  : G, g, H, h from the summed buffers with the global N; the context is then in the
 *                          same state as after lsspa_reduce on the stacked rows (up to summation order). */
int lsspa_reduce_partial(lsspa_ctx* ctx, const void* X_train, int64_t ld_train, const void* y_train,
                         int64_t n_local, const void* X_test, int64_t ld_test, const void* y_test,
                         int64_t m_local, int64_t M_total, int32_t p, int32_t dtype, int32_t location);
int lsspa_reduce_buffer(lsspa_ctx* ctx, void** device_ptr, int64_t* count);
int lsspa_reduce_finish(lsspa_ctx* ctx, int64_t N_total, double reg);

/* Load an already reduced problem (host pointers) -- the inputs square_shapley takes
 * (ls_spa/ls_spa.py:256-258) in Gram form.  G [p][p], g [p]; aug_train >= g^T G^-1 g.
 * tri != 0: H [p][p], h [p] (test Gram);  tri == 0: Ft [p][m] (transposed test factor), ytil [m]. */
int lsspa_set_reduced(lsspa_ctx* ctx, int32_t p, const double* G, const double* g, double aug_train,
                      int32_t tri, const double* H, const double* h, int32_t m, const double* Ft,
                      const double* ytil, double y_norm_sq);

/* Element type of the per-ordering work (Cholesky factors, solves): LSSPA_F64 (default; matches the
 * reference to ~1e-15) or LSSPA_F32 (half the HBM traffic, fp32 MFMA; the Gram reduction, the lift
 * accumulation and the running statistics stay fp64).  The reference has no counterpart: it computes
 * in float64 throughout (ls_spa/ls_spa.py:309-317). */
int lsspa_set_precision(lsspa_ctx* ctx, int32_t dtype);

int lsspa_get_problem(const lsspa_ctx* ctx, int32_t* p, int32_t* m, int32_t* tri, double* y_norm_sq);
/* device -> host copies of the reduced problem; any pointer may be NULL */
int lsspa_get_gram(lsspa_ctx* ctx, double* G, double* g, double* H, double* h);

/* a7 -- replaces theta = lstsq(...) and r_squared (ls_spa/ls_spa.py:240-243): factor the
 * identity ordering, back-substitute, sum its lift vector.  info: LSSPA_INFO_* flags. */
int lsspa_full_fit(lsspa_ctx* ctx, double* theta, double* r_squared, int32_t* info);
/* the reduce_data outputs in the reference's layout: R_tr [p][p] upper triangular,
 * q_tr [p], F_te [m][p], q_te [m]  (any pointer may be NULL) */
int lsspa_get_factors(lsspa_ctx* ctx, double* R_tr, double* q_tr, double* F_te, double* q_te);

/* a2 + a3 -- replaces square_shapley over a batch of orderings and the antithetical
 * pairing (ls_spa/ls_spa.py:203-208, :256-287).  perms: host int32 [B][p].  With
 * antithetical != 0 every ordering is also evaluated reversed and the two lift vectors are
 * averaged (one sample).  lifts_out: host [B][p] or NULL.  accumulate = 1: add the batch's
 * moments about the running mean to the pending-batch buffer (a4: all-reduce it over the ranks if there are several,
 * then lsspa_stats_merge).  accumulate = 2, one GPU: fold the batch into the running statistics at once -- the effect
 * of accumulate = 1 followed by lsspa_stats_merge (ls_spa/ls_spa.py:212-216 for the whole batch), in one launch for
 * p <= 128; refused (LSSPA_ERR_STATE) while a batch is pending. */
int lsspa_lift_batch(lsspa_ctx* ctx, const int32_t* perms, int32_t B, int32_t antithetical,
                     double* lifts_out, int32_t accumulate);
/* The two halves of lsspa_lift_batch.  lsspa_lift_launch enqueues every kernel of a batch up to its lift vectors and
 * returns a ticket; nothing it does touches the running statistics.  lsspa_lift_collect folds `count` samples of the
 * ticket's batch, from sample `first` on, into the pending buffer (accumulate) and / or copies their lift vectors out
 * (count <= 0: all the rest); the parts of a batch are taken front to back.  lsspa_lift_discard drops what is left.
 * Why: the reference evaluates its stop rule after every batch_size samples (ls_spa/ls_spa.py:222-230); a batch
 * that small may fill a fraction of the GPU (16 samples per rank when 128 are dealt over 8 GPUs).  The driver
 * therefore launches several chunks of a QMC sampler's orderings as ONE batch, accumulates and checks them chunk by
 * chunk in the reference's order, and discards the chunks beyond a stop -- same results, a fuller GPU.  With two
 * lanes (lsspa_set_lanes) a second batch may be launched before the first is fully collected. */
int lsspa_lift_launch(lsspa_ctx* ctx, const int32_t* perms, int32_t B, int32_t antithetical, int32_t* ticket);
int lsspa_lift_collect(lsspa_ctx* ctx, int32_t ticket, int32_t first, int32_t count, double* lifts_out,
                       int32_t accumulate);
/* n_chunks consecutive parts of `chunk` samples each, from sample `first` on, folded one after the other: the effect of
 * n_chunks calls of lsspa_lift_collect(ticket, first + c chunk, chunk, NULL, accumulate).  With accumulate = 2 on a small
 * problem (p <= 128, chunks of up to 512 samples, at most 32 of them, one rank) the parts' statistics are ONE launch
 * (every part still merged by itself, in order: the same numbers to the last bit) -- at p = 100 a part's own launch is
 * a fifth of its step and cannot hide behind the next batch's kernel. */
int lsspa_lift_collect_chunks(lsspa_ctx* ctx, int32_t ticket, int32_t first, int32_t chunk, int32_t n_chunks,
                              int32_t accumulate);
int lsspa_lift_discard(lsspa_ctx* ctx, int32_t ticket);
/* 1 (default): every batch runs on the context's stream, one after the other.  2: successive batches alternate
 * between two workspaces on two streams, staggered by half a batch, so that the memory-bound stages (gather, lifts)
 * and the launch tails of one batch run beside the matrix-pipe-bound stages of the other; statistics, collectives
 * and merges stay in batch order on the context's stream.  Results do not depend on the setting. */
int lsspa_set_lanes(lsspa_ctx* ctx, int32_t n);
int lsspa_get_info(lsspa_ctx* ctx, int32_t* info);
/* The same word without waiting for batches that were launched and never collected (they may still be running on a
 * lane): the bits of every batch whose samples were collected are in it.  lsspa_get_info waits for everything. */
int lsspa_get_info_collected(lsspa_ctx* ctx, int32_t* info);
/* Every ordering's lifts telescope to the R^2 of the full model (ls_spa/ls_spa.py:284-285), so every sample's lift
 * vector must sum to it.  Once lsspa_full_fit has computed that R^2, every batch is checked on the device right after
 * its lifts: a deviation beyond 1e-9 max(1, |R^2|) (fp32 per-ordering work: 1e-4) raises LSSPA_INFO_SUM -- the data a
 * kernel took from another workgroup, a tile left out, an ordering read wrong all show here.  This returns the largest
 * deviation of all batches since the last lsspa_stats_reset (waits for everything in flight). */
int lsspa_get_sum_deviation(lsspa_ctx* ctx, double* max_deviation);

/* a4 -- replaces merge_sample_mean / merge_sample_cov (ls_spa/ls_spa.py:103-119, :212-216).
 * The pending-batch buffer is a device fp64 array [1 + p + p*p] = [n_b, sum(l - mu), sum (l - mu)(l - mu)^T];
 * with several GPUs it is the (only) all-reduce target; lsspa_stats_merge folds it into the
 * running (n, mean, M2) by Chan's pairwise update and clears it. */
int lsspa_stats_reset(lsspa_ctx* ctx);
int lsspa_stats_pending(lsspa_ctx* ctx, void** device_ptr, int64_t* count);
int lsspa_stats_merge(lsspa_ctx* ctx);
/* n samples, mean [p], biased covariance [p][p] (may be NULL) */
int lsspa_stats_get(lsspa_ctx* ctx, int64_t* n, double* mean, double* cov_biased);
/* checkpoint / resume: overwrite the running statistics with (n, mean [p], biased covariance [p][p]) as
 * lsspa_stats_get returned them; the pending buffer is cleared.  (The reference keeps these in three Python
 * locals, ls_spa/ls_spa.py:190-192, and cannot resume.) */
int lsspa_stats_set(lsspa_ctx* ctx, int64_t n, const double* mean, const double* cov_biased);

/* a5 -- device-side replacement of error_estimates (ls_spa/ls_spa.py:321-341) in its thin form:
 * 1024 draws x = Xi (L - 1 mean^T) / sqrt(n (n - 1)) with L the [n][p] lift vectors of all samples so far, which
 * has the covariance C_unbiased / n the reference samples from, without any p x p factorisation.
 *   lsspa_history_enable : keep every accumulated sample's lift vector in HBM (capacity = rows allocated up front,
 *                          grown geometrically when exceeded; 0 switches the history -- and the running form below -- off
 *                          and gives a history of more than 64 MB back; smaller buffers stay for the next call).
 *                          lsspa_stats_reset / lsspa_reduce / lsspa_set_reduced empty the history.
 *   lsspa_history_get    : copy it out ([count][p], host); lifts may be NULL to query the count
 *   lsspa_history_append : push rows back in (resume)
 *   lsspa_error_draws    : xi is host fp64 [1024][ld_xi], standard normals; its first n_local columns go with this
 *                          context's n_local history rows, in order.  n_total is the sample count over all ranks
 *                          (== n_local on one GPU).  The centring uses the merged running mean.
 *   lsspa_error_buffer   : device pointer / element count of the draws, the all-reduce(SUM) target between
 *                          lsspa_error_draws and lsspa_error_quantiles when the samples are spread over ranks
 *   lsspa_error_quantiles: feature_errors [p] = 0.95-quantile of |x_a| over the draws, overall_error = the same
 *                          quantile of ||x||_2 (numpy's default linear interpolation)
 * Lifetime of host buffers: every entry point that takes a host pointer has finished reading it when it returns
 * (lsspa_error_draws and lsspa_stats_set synchronise their upload; lsspa_lift_batch copies the orderings into
 * pinned staging before it returns), so the caller may pass temporaries. */
int lsspa_history_enable(lsspa_ctx* ctx, int64_t capacity);
int lsspa_history_get(lsspa_ctx* ctx, int64_t* count, double* lifts);
int lsspa_history_append(lsspa_ctx* ctx, const double* lifts, int64_t rows);
int lsspa_error_draws(lsspa_ctx* ctx, const double* xi, int64_t ld_xi, int64_t n_local, int64_t n_total);
int lsspa_error_buffer(lsspa_ctx* ctx, void** device_ptr, int64_t* count);
int lsspa_error_quantiles(lsspa_ctx* ctx, double* feature_errors, double* overall_error);

/* a5, running form (what ls_spa(error_estimator='device') uses since round 5) -- the same estimator with a cost per
 * check that does not depend on the number of samples, and no host random numbers.  Xi[d][k], draw d of sample k, is a
 * pure function of (seed, k, d): Philox4x32-10 with key = seed and counter = (k, d / 2), Box-Muller on its four
 * output words (k_error.hip; the generator's published known-answer vectors and the normals themselves are pinned by
 * tests/philox_ref.py through lsspa_error_xi).  The context keeps D = Xi L [1024][p] and s = Xi 1 [1024] over the
 * samples folded in so far; at a check x = (D - s mean^T) / sqrt(n (n - 1)), which given the lift vectors is
 * N(0, C_unbiased / n) exactly as the reference's draws are (ls_spa/ls_spa.py:334-336), and the quantiles follow as
 * above (:337-340).  Successive checks share the columns of Xi of the samples they share (the reference redraws).
 *   lsspa_error_running_enable  : allocate and zero D, s (and a small staging of lift vectors: every sample that
 *                                 lsspa_lift_batch / _collect accumulates is staged until the next advance);
 *                                 lsspa_stats_reset zeroes D and s again; lsspa_history_enable(ctx, 0) switches it off
 *   lsspa_error_advance         : fold the staged lift vectors in; they are samples first_id, first_id + stride, ...
 *                                 of the run (the driver deals sample i of a chunk to rank i mod world, so the ids
 *                                 -- hence Xi and every result -- do not depend on the number of ranks)
 *   lsspa_error_running_draws   : x of this context's samples into the draws buffer (lsspa_error_buffer): with
 *                                 several ranks all-reduce it (lsspa_error_allreduce) -- x is linear in (D, s)
 *   lsspa_error_quantiles_enqueue / lsspa_error_result : the quantile kernels, then feature errors, overall error, the
 *                                 running mean and n copied into pinned slot `slot` (0 .. 63) behind an event -- nothing
 *                                 waits.  lsspa_error_result reads a slot: wait != 0 blocks on its event, wait == 0
 *                                 polls (*ready = 0: not yet).  This is what lets the driver evaluate the stop rule of
 *                                 check k while the samples of check k + 1 are already running (they are dropped on a
 *                                 stop), so the estimator is off the critical path.
 *   lsspa_error_state_get / _set: D [1024][p] and s [1024] to / from host memory (checkpoint / resume)
 *   lsspa_error_xi              : test hook -- the normals Xi [1024][count] of `count` sample ids, to host memory */
int lsspa_error_running_enable(lsspa_ctx* ctx, uint64_t seed);
int lsspa_error_advance(lsspa_ctx* ctx, int64_t first_id, int64_t stride);
int lsspa_error_running_draws(lsspa_ctx* ctx, int64_t n_total);
int lsspa_error_quantiles_enqueue(lsspa_ctx* ctx, int32_t slot);
/* one rank: lsspa_error_running_draws + lsspa_error_quantiles_enqueue in one call and two launches -- the quantile
 * kernels evaluate x = (D - s mean^T) / sqrt(n (n - 1)) as they read it, the draws buffer is not written */
int lsspa_error_check_enqueue(lsspa_ctx* ctx, int64_t n_total, int32_t slot);
int lsspa_error_result(lsspa_ctx* ctx, int32_t slot, int32_t wait, int32_t* ready, double* feature_errors,
                       double* overall_error, double* mean, int64_t* n);
/* The per-chunk tail of a launched batch in one call, for hosts whose own call overhead would bound a small problem
 * (a chunk of 256 orderings takes the GPU 37 us at p = 100).  For every chunk c = 0 .. n_chunks - 1, in order:
 * lsspa_lift_collect(ticket, first[c], count[c], NULL, accumulate) -- skipped for count[c] == 0 --; with a communicator
 * of several ranks lsspa_stats_allreduce + lsspa_stats_merge; lsspa_error_advance(first_id[c], stride); and, where
 * n_after[c] > 0, the check of that global sample count into slot[c] (one rank: lsspa_error_check_enqueue; several:
 * lsspa_error_running_draws + lsspa_error_allreduce + lsspa_error_quantiles_enqueue).  Nothing waits; the results are
 * read with lsspa_error_result.  Order and arithmetic are those of the separate calls (ls_spa/ls_spa.py:203-230).
 * One rank, p <= 128, 2 .. 32 chunks of up to 512 samples that follow each other in the batch: the statistics, the
 * estimator's sums and all the checks of the call are five launches (the state after every chunk is kept for the check
 * that belongs to it) -- the numbers are those of the chunk-by-chunk calls to the last bit. */
int lsspa_group_collect(lsspa_ctx* ctx, int32_t ticket, int32_t n_chunks, const int32_t* first, const int32_t* count,
                        const int64_t* first_id, int64_t stride, const int64_t* n_after, const int32_t* slot);
int lsspa_error_state_get(lsspa_ctx* ctx, double* D, double* s);
int lsspa_error_state_set(lsspa_ctx* ctx, const double* D, const double* s);
int lsspa_error_xi(lsspa_ctx* ctx, uint64_t seed, int64_t first_id, int64_t stride, int64_t count, double* xi);

/* (e) -- collectives.  The reference is single-process; these implement the multi-GPU form of its running-statistics
 * merge (ls_spa/ls_spa.py:103-119, :212-216): orderings are dealt over one process per GPU and the ONLY data-path
 * exchange is one SUM all-reduce of the pending-batch moments per chunk.  RCCL over xGMI, resolved
 * at run time (no link-time dependency; a single-GPU process never loads it); every collective is enqueued on the
 * context's stream, so kernels -> all-reduce -> merge run without host synchronisation.
 *   lsspa_comm_unique_id  : 128 opaque bytes made on rank 0 (ncclGetUniqueId) and handed to the other ranks by the
 *                           host (the Python package uses a TCP exchange on MASTER_ADDR); errors: lsspa_last_error(NULL)
 *   lsspa_comm_init       : collective over all ranks; binds the communicator to this context's GPU
 *   lsspa_stats_allreduce : the pending buffer [n_b, S, Q]; from p = 2048 on Q travels as its upper triangle
 *                           (1 + p + p (p + 1) / 2 elements: half the bytes; Q is exactly symmetric)
 *   lsspa_reduce_allreduce: the Gram sums of a row-sharded reduction (between lsspa_reduce_partial and _finish)
 *   lsspa_error_allreduce : the partial draws of the device-side estimator (between lsspa_error_draws and _quantiles)
 *   lsspa_comm_sum_i64    : host integers, summed over the ranks in place (row counts of a sharded reduction)
 *   lsspa_comm_allgather  : host fp64 [count] per rank -> [world][count] (lift vectors for attribution_history) */
#define LSSPA_COMM_ID_BYTES 128
int lsspa_comm_unique_id(uint8_t* id128);
int lsspa_comm_init(lsspa_ctx* ctx, const uint8_t* id128, int32_t rank, int32_t world);
int lsspa_comm_destroy(lsspa_ctx* ctx);
/* rank and size as RCCL reports them for this context's communicator (ncclCommUserRank / ncclCommCount) */
int lsspa_comm_info(const lsspa_ctx* ctx, int32_t* rank, int32_t* world);
int lsspa_stats_allreduce(lsspa_ctx* ctx);
int lsspa_reduce_allreduce(lsspa_ctx* ctx);
int lsspa_error_allreduce(lsspa_ctx* ctx);
int lsspa_comm_sum_i64(lsspa_ctx* ctx, int64_t* values, int32_t count);
int lsspa_comm_allgather(lsspa_ctx* ctx, const double* send, int64_t count, double* recv);

/* per-kernel-class HIP-event timing on the context's stream */
#define LSSPA_K_GATHER 0
#define LSSPA_K_CHOL_DIAG 1
#define LSSPA_K_CHOL_PANEL 2
#define LSSPA_K_STRIP 3
#define LSSPA_K_LIFT 4
#define LSSPA_K_STATS 5
#define LSSPA_K_GRAM 6
#define LSSPA_K_ERROR 7
#define LSSPA_K_COMM 8
#define LSSPA_K_SMALL 9     /* fused small-p kernel: gather .. lifts of one ordering in one workgroup */
#define LSSPA_K_COUNT 10
int lsspa_profile_enable(lsspa_ctx* ctx, int32_t on);
int lsspa_profile_get(lsspa_ctx* ctx, int32_t kernel_class, double* total_ms, int64_t* launches);
int lsspa_profile_reset(lsspa_ctx* ctx);

/* developer switches: cross-checks of kernel variants against each other (0 = shipped configuration):
 *   128  tri mode: V by the strip kernel (the shipped path of rect mode) instead of V^T by the panel launches' X tiles
 *   512  tri mode: the lift kernel reads V^T back and scans it, instead of the X tiles scanning their own blocks
 *  1024  general path also for small problems (p + 1 <= 128 normally takes the fused one-workgroup kernel)
 *  4096  fault injection: the L tiles never raise the flag the fused lift scan waits for -- every X tile runs into the
 *        scan's time-out (~0.1 s a launch), LSSPA_INFO_SCAN_WAIT and LSSPA_INFO_SUM are set, nothing hangs
 * 16384  small problems: the LDS-resident kernel also where the register-resident one applies (p + 1 <= 112)
 * 65536  Gram kernel: workgroup id = unit (the units of a row slice spread over the XCDs instead of sharing one L2)
 * Every combination computes the same lifts (tests/test_gpu_kernels.py).  Environment, read once per process:
 * LSSPA_HANDOVER=k moves the point at which the second lane's next launch sequence may start to "after panel launch k"
 * (default: the middle one).  Retired in round 5 (each switched between two code paths that both stay in use and
 * are pinned against the oracle by themselves): 64 plain dispatch order in the panel kernel (what a matrix count that
 * is not a multiple of eight takes), 256 unpaired gather (what antithetical = 0 takes), 2048 no skipping of the
 * all-padding tiles.  Earlier rounds carried more; DESIGN_HISTORY.md. */
int lsspa_set_flags(lsspa_ctx* ctx, int32_t flags);

/* test hooks */
/* the nth device allocation from now on fails with LSSPA_ERR_NOMEM (0 disarms): exercises the out-of-memory
 * paths, after which a context must still be usable (e.g. with a smaller batch) */
int lsspa_debug_fail_alloc(lsspa_ctx* ctx, int32_t nth);
/* use the packed (upper-triangle) form of lsspa_stats_allreduce from this p on (default 2048) */
int lsspa_debug_pack_from(lsspa_ctx* ctx, int32_t p_min);
/* Host helper of the QMC ordering sources (the reference: np.argsort of Sobol' points / projected normals,
 * experiments/ground_truth_medium.py:56-71): out [B][p] = the argsort of every row of keys [B][p], on up to `threads`
 * threads of this library (no interpreter lock between them and the caller's other threads).  A row with all keys
 * different has one argsort; redo[s] = 1 marks the rows with equal keys or a NaN, whose order is numpy's own business:
 * the caller sorts those with numpy (*n_redo of them).  No context, no GPU. */
int lsspa_host_argsort_rows(const double* keys, int64_t B, int32_t p, int32_t* out, uint8_t* redo, int32_t threads,
                            int64_t* n_redo);
/* The 'argsort' ordering source as a thread of this library (the reference: np.argsort(qmc.Sobol(p, seed).random(n), axis=1),
 * experiments/ground_truth_medium.py:56-60): Sobol' points by SciPy's own recurrence -- the caller reads direction numbers
 * sv [p][bits], the state before the first step q0 [p] and the scale off a SciPy engine it built, having checked them
 * against that engine's output -- and their row argsort, drawn `block` orderings at a time up to `ahead` ahead of the
 * consumer (`ahead_unasked` until the first lsspa_sampler_take) on `threads` sort threads; ordering number g of the run
 * belongs to rank g mod world, only those rows are sorted and handed out.  lsspa_sampler_take: the next `count`
 * orderings of the run -- *n_taken of them exist --, this rank's rows of them into out [cap][p] (*n_own rows); rows with
 * equal keys (numpy's order among them is its own) are listed by position in `out` and number in the run: the caller
 * sorts those with numpy.  Errors of these three calls are read with lsspa_last_error(NULL).  No context, no GPU. */
int lsspa_sampler_create(int32_t p, int32_t bits, const uint64_t* sv, const uint64_t* q0, double scale, int64_t limit,
                         int32_t block, int64_t ahead, int64_t ahead_unasked, int32_t threads, int32_t rank,
                         int32_t world, void** out);
int lsspa_sampler_take(void* sampler, int64_t count, int32_t* out, int64_t cap, int64_t* n_taken, int64_t* n_own,
                       int64_t* redo_pos, int64_t* redo_id, int64_t* n_redo);
int lsspa_sampler_destroy(void* sampler);
/* overwrite the R^2 the batches' sums are checked against (LSSPA_INFO_SUM; set by lsspa_full_fit): a test makes the
 * check fire on a healthy engine with it */
int lsspa_debug_set_r2(lsspa_ctx* ctx, double r2);
/* 1 if every row of perms [B][p] is a permutation of 0..p-1, else 0 -- the check every batch launch makes on the host
 * (csrc/host_perms.cpp: 128-bit sets by AVX2 for 8 <= p <= 128, a stamp array otherwise); plain != 0: the stamp loop
 * alone.  No context, no GPU: host code only. */
int lsspa_debug_check_perms(const int32_t* perms, int32_t B, int32_t p, int32_t plain);
int lsspa_mfma_probe(lsspa_ctx* ctx, const double* A16x4, const double* B4x16, double* D16x16, int32_t dtype);
/* factor one ordering and copy the padded factor(s) out: L [p_pad][p_pad] (train),
 * Lt [p_pad][p_pad] (test, tri mode only, else untouched), V [n_iblk*64][m_pad] */
int lsspa_debug_factor(lsspa_ctx* ctx, const int32_t* perm, double* L, double* Lt, double* V,
                       int32_t* p_pad, int32_t* m_pad, int32_t* v_rows);

#ifdef __cplusplus
}
#endif
#endif /* LSSPA_H */
