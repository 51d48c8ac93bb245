/*
 * fastmatch_hip.h -- C-ABI of libfastmatch_hip.so: the MI355X (gfx950) drop-in for the
 * arithmetic behind Fast-Match's descriptor-matching hot path.
 *
 * The reference (arnfred/Fast-Match) has no native FFI of its own: its Cython modules
 * call OpenCV's cv2.BFMatcher from Python.  Each entry point below therefore names the
 * reference call site whose arithmetic it replaces (file:line in the reference tree);
 * INTEGRATION.md shows the ctypes binding a maintainer would add at those sites.
 *
 * Conventions: plain C, no exceptions.  Every function returns 0 on success and a
 * negative FM_E* code on failure; fm_last_error() returns the message of the last
 * failure on that context (or the last context-less failure when ctx == NULL).
 * The caller owns every host buffer; the library owns device memory behind the opaque
 * handles.  One fm_ctx per device; calls on one ctx are serialised on its HIP stream
 * (not re-entrant per ctx; distinct ctxs are independent).  All calls are synchronous:
 * host outputs are valid on return.  There is no CPU fallback: without a usable
 * gfx950 device fm_ctx_create fails.
 */
#ifndef FASTMATCH_HIP_H
#define FASTMATCH_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FM_OK            0
#define FM_EINVAL       -1   /* bad argument (null handle, dim mismatch, k unsupported ...)  */
#define FM_EDEVICE      -2   /* HIP runtime error; message carries hipGetErrorString          */
#define FM_ENOMEM       -3   /* device or host allocation failed                              */
#define FM_EUNSUPPORTED -4   /* valid request this build cannot serve (e.g. dim > 128)        */

/* Bank kinds (fm_bank_info) */
#define FM_NO_STREAM ((void*)(intptr_t)-1)   /* "no consumer stream" (NULL is the null stream) */
#define FM_BANK_I8   1       /* uint8 / integer-valued float32 rows: exact int8-MFMA route    */
#define FM_BANK_F32  2       /* general float32 rows: fp32 fma-chain route                    */

typedef struct fm_ctx  fm_ctx;
typedef struct fm_bank fm_bank;

/* ABI revision of this header.  It changes whenever an existing signature or struct layout does (revision 3, r03:
 * fm_expand_fetch_many gained `slot`, fm_expand_desc gained `metric`; revision 4, r04: additions only; revision 5, r04:
 * fm_expand_desc gained the trailing `lazy`; revision 7, r05: additions -- fm_self_dist_plan, fm_bank_create_f32_cap,
 * fm_bank_append_f32, fm_expand_set_log / _log_counts / _fetch_log -- and fm_expand_run_lazy refuses to resume a run
 * that did not park; revision 8, r06: additions -- fm_knn, the option "f32_bound_every" -- and rounds[i][5] of the
 * per-round log may be -2).  A binding
 * compares fm_abi_version() with the FM_ABI_VERSION it was written against before its first call.            */
#define FM_ABI_VERSION 8
int  fm_abi_version(void);

typedef struct fm_stats {
    double   kernel_ms;      /* HIP-event time of the distance kernels since fm_reset_stats   */
    double   total_ms;       /* HIP-event time of whole calls (all kernels + copies)          */
    int64_t  kernel_launches;/* number of distance-kernel launches in kernel_ms               */
    int64_t  pairs;          /* descriptor pairs evaluated by those launches                  */
    int64_t  calls;          /* API calls accounted in total_ms                               */
} fm_stats;

/* fm_stats plus what later revisions add; struct_bytes = sizeof(fm_stats_ex) of the library that filled it.
 * bytes_moved = ALGORITHMIC bytes of the distance-kernel launches in kernel_ms: every bank row of a launch read
 * once (128 B per integer-route row, 512 B per float32 row) -- the numerator of an HBM-roofline fraction; the
 * bytes a launch really fetched come from the PMC counters (profiles/).                                       */
typedef struct fm_stats_ex {
    int64_t  struct_bytes;
    double   kernel_ms, total_ms;
    int64_t  kernel_launches, pairs, calls;
    int64_t  bytes_moved;
} fm_stats_ex;

/* ---- context ---------------------------------------------------------------------- */
int  fm_ctx_create(int device_id, fm_ctx** ctx);
int  fm_ctx_destroy(fm_ctx* ctx);
/* Per-context options: batch shape and launch tuning by name.  RESULTS NEVER DEPEND ON THEM (the parity
 * tests run the kernels under several settings); they replace the process-wide FM_* environment
 * variables, which now only seed a new context's defaults.  The reference has no counterpart: cv2 is
 * configured per matcher object (fastmatch.pyx:122, 161 build one per call).
 *   "batch_group"  1..16  fm_match_accepted_batch: most image pairs per distance-kernel launch (8)
 *   "batch_tail"   0..16  ... pairs in the short launch a run of pairs ends with (2; 0 = none: for callers
 *                         that enqueue the next batch before waiting for this one, fm_mark / fm_wait)
 *   "nsplit" "nb" "nw"    K1 grid shape: splits of the reduction range, 16-row blocks per wave (4 | 6 | 8),
 *                         waves per workgroup (4 | 8 | 16); 0 = built-in rule
 *   "nbuf"         0|2|3  K1 / K2 LDS stage buffers (0 = 3; top-2 shapes other than 4 blocks per wave: 2)
 *   "prio" "glds" "coop"  0|1  s_setprio around the MFMA burst / LDS-DMA staging / cross-workgroup bounds
 *   "f32_filter"   0..2   float32 route: 0 = all-pairs kernel only, 1 = fp16 filter for large calls, 2 = always
 *   "f32_nw" "f32_nsplit" "f32_fused" "f32_lpc"   K8 launch shape (0 / -1 = rule)
 *   "f32_bound_every"    K8: re-read the shared bounds every n-th stage once a sweep is 8 stages old (1 .. 64, a power of two; 4)
 *   "async_time_every"    every n-th async call carries kernel-timing events (4; 0 = none)
 *   "k1_order"     0..2   K1: how workgroups map to (output chunk, split of the reduction range): 0 split major,
 *                         1 the workgroups of one XCD own a set of output chunks for all splits, 2 they own a
 *                         contiguous share of the split-major order (the guide's XCD remap); see rowreduce.hip
 *   "bound_every"  1..1024  K1 / K2: workgroups re-read the shared K-th-best bounds at every stage of a sweep's first
 *                         eight and then at every n-th (a power of two; 16).  A stale bound is merely weaker
 *   "self_tri"     0..2   fm_self_dist / fm_self_dist_batch: 1 = the triangular sweep (every distance once) -- float32-route banks (r06)
 *                         from 65536 padded rows on; integer banks from 32768 padded rows
 *                         on and for every run of two or more integer banks in a batch call, whatever their sizes (1),
 *                         0 = always the masked full sweep, 2 = always the triangular one
 *   "tri_stages"   0..    ... 128-row stages per workgroup of its launch B (0 = chosen per bank size: fm_self_dist_plan)
 *   "refill_grid"  1..    fm_bank_refill_u8_async: workgroups of its preparation kernel (128: few, long-lived ones beside
 *                         the distance kernels)
 *   "expand_big"   0|1    K7: re-run pairs whose round exceeds 2048 query rows in the 4096-row variant (1)
 *   "expand_huge"  0|1    K7: ... and those that still do in the variant that takes a radius subset of any size in chunks (1)
 *   "expand_delegate" 0.. K7: a chunked round of at least this many descriptor pairs parks its run; the round's cross-check
 *                         is run by the dense kernels on the whole GPU and the run resumed (1 500 000; 0 = never)
 *   "delegated_rounds" 0  a COUNTER, not a setting: rounds whose cross-check the dense kernels ran (fm_expand_run,
 *                         fm_expand_run_lazy) since it was last set to 0 -- the only value it accepts
 *   "expand_grow"  0..4   K7: how often a run that fills its pending stack / result list / hash table is
 *                         repeated in a run state four times as large (2; r05: counted per array)
 *   "expand_prof"  0|1    K7: per-phase timers of the first pair of a launch on stderr
 * Unknown names and out-of-range values return FM_EINVAL.                                          */
int  fm_ctx_set_option(fm_ctx* ctx, const char* name, int64_t value);
int  fm_ctx_get_option(fm_ctx* ctx, const char* name, int64_t* value);
const char* fm_last_error(const fm_ctx* ctx);
int  fm_sync(fm_ctx* ctx);
int  fm_get_stats(fm_ctx* ctx, fm_stats* out);
/* Writes min(out_bytes, sizeof(fm_stats_ex)) bytes: a caller built against an older, shorter struct stays valid. */
int  fm_get_stats_ex(fm_ctx* ctx, fm_stats_ex* out, int64_t out_bytes);
int  fm_reset_stats(fm_ctx* ctx);
/* Name of the device the context runs on (e.g. "gfx950:..."), written NUL-terminated.  */
int  fm_device_name(fm_ctx* ctx, char* buf, int buflen);
/* Float32 route diagnostics: how many row-reduces went through the fp16 MFMA filter
 * (filter_f16.hip) and how many of those had to be redone by the all-pairs float32 kernel
 * because a candidate list could have been incomplete (results are identical either way). */
int  fm_f32_filter_stats(fm_ctx* ctx, int64_t* launches, int64_t* fallbacks);

/* Page-locked host memory for output buffers (optional): results copied into such a buffer
 * move by direct DMA instead of through the runtime's pageable staging path.             */
int  fm_host_alloc(fm_ctx* ctx, int64_t bytes, void** ptr);
int  fm_host_free(fm_ctx* ctx, void* ptr);

/* ---- descriptor banks --------------------------------------------------------------
 * A bank is a device-resident [n, dim] descriptor matrix plus what the kernels need
 * (bytes XOR 0x80 as int8, row norms).  It replaces the ndarray arguments the reference
 * hands to cv2 at every matcher call: Metric_Cache.original/thumb["descriptors"]
 * (cache.pyx:255-260, 278-284) and the Grid_Cache cell descriptors (cache.pyx:134-137).
 * dim must be <= 128 (SIFT = 128); shorter rows are zero-padded, which leaves every L2
 * distance unchanged.  n may be 0.
 *   fm_bank_create_u8  : rows are uint8 (CV_8U descriptors).
 *   fm_bank_create_f32 : rows are float32 (what cv2 SIFT emits).  If every value is an
 *       integer in [0,255] the bank takes the exact int8 route (kind FM_BANK_I8),
 *       otherwise it stays float32 (kind FM_BANK_F32).
 * A query/train pair must have the same kind and dim (cv2 raises on dtype mismatch).    */
int  fm_bank_create_u8 (fm_ctx* ctx, const uint8_t* rows, int64_t n, int dim, fm_bank** bank);
int  fm_bank_create_f32(fm_ctx* ctx, const float*   rows, int64_t n, int dim, fm_bank** bank);
/* Same, but the bank takes the float32 route even if every value happens to be an integer in
 * 0..255: for banks that must pair with a non-integer bank (e.g. one Grid_Cache cell of a
 * RootSIFT image whose values are all 0 or 1).  Distances are the same numbers either way.  */
int  fm_bank_create_f32_route(fm_ctx* ctx, const float* rows, int64_t n, int dim, fm_bank** bank);
int  fm_bank_destroy(fm_ctx* ctx, fm_bank* bank);
int  fm_bank_info(const fm_bank* bank, int64_t* n, int* dim, int* kind);
/* Attach per-row self distances (Metric_Cache.*["distances"], float64, cache.pyx:252,273)
 * to a query bank so that fm_match_ratio can run the ratio test on the device.          */
int  fm_bank_set_selfdist(fm_ctx* ctx, fm_bank* bank, const double* selfdist /*[n]*/);

/* A new image into an existing bank (created by fm_bank_create_u8 / an integer-valued fm_bank_create_f32 with at
 * least as many rows): no allocation and no host synchronisation -- the copy from `rows` (page-locked memory,
 * fm_host_alloc, for a copy that really is asynchronous; [n][dim] uint8) and the preparation kernel are enqueued
 * on the context's UPLOAD stream and run beside the kernels of the other streams.  The reference builds a new
 * Metric_Cache / cell array per image (cache.pyx:263-284, 124-138); a pipeline over a stream of images re-uses
 * the device arrays instead.  Rules: (1) work enqueued earlier that reads the bank must be COMPLETE (fm_wait on a
 * ticket taken after it, or fm_sync) -- the refill does not wait for it; (2) fm_upload_fence() before the first
 * call that uses a refilled bank; (3) self distances attached to the bank are stale: fm_self_dist_batch.       */
int  fm_bank_refill_u8_async(fm_ctx* ctx, fm_bank* bank, const uint8_t* rows /*page-locked*/, int64_t n);
/* Makes everything enqueued on the context AFTER this call wait (on the device) for the refills enqueued before. */
int  fm_upload_fence(fm_ctx* ctx);

/* ---- K2: brute-force 2-NN ------------------------------------------------------------
 * Replaces cv2.BFMatcher(cv2.NORM_L2, crossCheck=False).knnMatch(q, t, k=2)
 *   matchutil.py:39-43 (bf_match), called from cache.pyx:250; Classic Matching.ipynb:63.
 * idx[2*i+r], dist[2*i+r] = r-th nearest train row of query row i (ascending float32 distance,
 * lower train index first on ties -- ties of the float32 distance, as in cv::batchDistance: d^2 = n
 * and n + 1 can share a root from 4 197 200 on); idx -1 / dist +inf where t has fewer than 2 rows. */
int  fm_knn2(fm_ctx* ctx, const fm_bank* q, const fm_bank* t,
             int32_t* idx /*[nq*2]*/, float* dist /*[nq*2]*/);

/* knnMatch(q, t, k) for the other k the reference's operator signature admits (matchutil.py:39-43 bf_match(dt1, dt2, k),
 * 46-67 flann_match: `k` is any int; the reference itself calls k = 1 and 2 only).  idx[k*i+r], dist[k*i+r] as in fm_knn2;
 * k = 1, 2: the matrix-core path of fm_knn2; 3 <= k <= 8: an exact vector-ALU kernel (off the hot path: ~20 ms for
 * 100k x 100k uint8 rows); k > 8: FM_EUNSUPPORTED.                                                           */
int  fm_knn(fm_ctx* ctx, const fm_bank* q, const fm_bank* t, int32_t k,
            int32_t* idx /*[nq*k]*/, float* dist /*[nq*k]*/);

/* Classic Ratio-Match in one call: knnMatch(q, t, k=2) then ratio = m[0].distance /
 * m[1].distance (float64) and ratio < tau  -- Classic Matching.ipynb cell 3 (JSON 59-72), the
 * baseline Fast-Match is compared against (README.md:5-13).  Returns the accepted matches
 * compacted in ascending query index: qidx/tidx/dist (= d1)/ratio[0 .. min(*n_accepted, cap)).
 * A zero second distance (where the notebook's Python division raises) is rejected.       */
int  fm_knn2_ratio(fm_ctx* ctx, const fm_bank* q, const fm_bank* t, double tau, int64_t cap,
                   int32_t* qidx, int32_t* tidx, float* dist, double* ratio, int64_t* n_accepted);

/* Replaces bf_match(d, d, k=2) + [r[1].distance] (cache.pyx:250-252; exact substitute
 * for the approximate flann_match at cache.pyx:271-273).  selfdist[i] = distance from
 * row i to its 2nd entry of the self 2-NN list, as float64 of the float32 value (+inf for a bank of one row).
 * That entry's distance is min over j != i of d(i, j): row i itself is at 0, always the first entry unless a
 * duplicate with a lower index is, and then i (or another duplicate) is the second at 0 = that minimum; only the
 * VALUE is kept, so its order among ties does not matter.  Computed as such: a top-1 sweep of the bank over
 * itself with the diagonal masked (K1's top-1 kernel, not the top-2 one).  r05: integer banks of >= 32768 rows
 * take the TRIANGULAR sweep (option "self_tri"): d(i, j) = d(j, i), so every 16 x 16 tile above the diagonal is
 * computed once and used for both of its rows' minima (rowreduce.hip, "TRI"); same values, bit for bit.        */
int  fm_self_dist(fm_ctx* ctx, const fm_bank* bank, double* selfdist /*[n]*/);
/* HOST code, no context: the workgroups of that triangular sweep for a bank padded to n_pad rows (a multiple of 128)
 * (the kernel itself holds no table: it derives workgroup i by the same arithmetic, rowreduce.hip tri_entry)
 * -- table[4 i .. 4 i + 3] = (output chunk of 512 rows, first 128-row stage, end stage, 0) of workgroup i, the first
 * *n_diag of them the diagonal blocks (launch A), the rest launch B; stages == 0 lets the library choose the piece
 * length (returned in *stages_used).  Writes min(cap, *n_workgroups) entries.  Every (chunk k, stage >= 4 k) is
 * covered exactly once: tests/test_tri_plan.py.                                                                  */
int  fm_self_dist_plan(int64_t n_pad, int32_t stages, int32_t* table /*[cap][4]*/, int64_t cap, int32_t* n_workgroups,
                       int32_t* n_diag, int32_t* stages_used);
/* The Metric_Cache builds of n images in one call: self distances of every bank, ATTACHED to it on the device
 * (as fm_bank_set_selfdist would, without the trip through the host); consecutive integer-route banks -- of any
 * sizes (r05) -- share the triangular sweep's two launches, up to "batch_group" banks each: a dataset of small
 * images builds its caches 2 - 3.7 x faster this way than bank by bank.  out == NULL (or every out[i] NULL):
 * enqueue only, complete after fm_sync / fm_wait; otherwise out[i] (may be NULL per bank) also receives bank i's
 * values and the call is synchronous.  A bank must not be read by work still in flight (see fm_bank_refill).  */
int  fm_self_dist_batch(fm_ctx* ctx, int32_t n, fm_bank* const* banks, double* const* out /*[n] of [n_i], or NULL*/);

/* ---- X1: cross-checked 1-NN ----------------------------------------------------------
 * Replaces cv2.BFMatcher(cv2.NORM_L2, crossCheck=True).knnMatch(q, t, k=1)
 *   fastmatch.pyx:122-123 (match_thumbs) and fastmatch.pyx:161-162 (match_position).
 * tidx[i] = train index matched to query row i or -1 (empty inner list), dist[i] its
 * float32 distance (+inf when unmatched).  OpenCV semantics: reverse-NN + scatter-min
 * with lowest-index tie-breaks.                                */
int  fm_xcheck1(fm_ctx* ctx, const fm_bank* q, const fm_bank* t,
                int32_t* tidx /*[nq]*/, float* dist /*[nq]*/);

/* X1 up to the election, for a train set sharded over ranks (one process per GPU): keys[q] =
 * (distance key << 32) | (t_offset + train row in this bank) of the closest row of THIS bank
 * that elects q, ~0 if none; distance key = the float32 bits of the distance on both routes
 * (cv::batchDistance takes the square root BEFORE it compares, and two integer d^2 >= 4 197 200
 * can share one float32 root: the order of the roots, not of d^2, is OpenCV's).  The element-wise
 * minimum over the ranks
 * (one all-reduce(min) of nq words) equals the keys of the unsharded fm_xcheck1: same
 * matches, same tie-breaks (cv::BFMatcher cross-check, fastmatch.pyx:122-123, 161-162).   */
int  fm_xcheck1_keys(fm_ctx* ctx, const fm_bank* q, const fm_bank* t, int64_t t_offset, uint64_t* keys);
/* The same with the keys left in caller-supplied DEVICE memory (uint64[nq]): the operand of the
 * all-reduce(min) over the ranks, so the keys of a shard never visit the host.                 */
int  fm_xcheck1_keys_dev(fm_ctx* ctx, const fm_bank* q, const fm_bank* t, int64_t t_offset, uint64_t* d_keys);

/* ---- R1: ratio + threshold -----------------------------------------------------------
 * Replaces  ratios = m.distance / query_dis[m.queryIdx]   (fastmatch.pyx:124, 165)
 *      and  ratios < tau                                   (fastmatch.pyx:50, 75, 82)
 * in float64 on the device.  qrows == NULL means row i uses selfdist[i].
 * ratio and pass may each be NULL.                                                      */
int  fm_ratio_filter(fm_ctx* ctx, const float* dist, const double* selfdist,
                     const int32_t* qrows, int64_t n, double tau,
                     double* ratio /*[n] or NULL*/, uint8_t* pass /*[n] or NULL*/,
                     int64_t* n_pass /*or NULL*/);

/* X1 + R1 fused on the device for banks that stay resident (one match_position /
 * match_thumbs round: fastmatch.pyx:161-165, 122-124).  q must carry self distances
 * (fm_bank_set_selfdist).  Outputs as fm_xcheck1 plus ratio[i] (nan when unmatched)
 * and pass[i] = ratio[i] < tau; *n_pass = number of accepted matches.                   */
int  fm_match_ratio(fm_ctx* ctx, const fm_bank* q, const fm_bank* t, double tau,
                    int32_t* tidx /*[nq]*/, float* dist /*[nq]*/,
                    double* ratio /*[nq] or NULL*/, uint8_t* pass /*[nq] or NULL*/,
                    int64_t* n_pass /*or NULL*/);

/* As fm_match_ratio, but returns only the accepted matches (ratio < tau), compacted on the
 * device in ascending query index: qidx/tidx/dist/ratio[0 .. min(*n_accepted, cap)).
 * *n_accepted is the total number accepted (may exceed cap; the rest is dropped).        */
int  fm_match_accepted(fm_ctx* ctx, const fm_bank* q, const fm_bank* t, double tau, int64_t cap,
                       int32_t* qidx, int32_t* tidx, float* dist, double* ratio,
                       int64_t* n_accepted);

/* As fm_match_accepted, but the call only ENQUEUES the work on the context's stream and returns:
 * the results are valid after the next fm_sync(ctx) (or any synchronous call on the context).
 * Every output -- qidx, tidx, dist, ratio and *n_accepted -- must be page-locked host memory
 * (fm_host_alloc), which the compaction kernel writes directly; banks must be integer valued.
 * A stream of image pairs then runs back to back on the GPU with no host round
 * trip between pairs (the reference maps its matcher over pairs sequentially, turntable.py:59):
 * the K1 launches follow each other on the context's stream while each pair's small kernels
 * (election, ratio test, compaction) run on a second stream beside the next pair's K1.          */
int  fm_match_accepted_async(fm_ctx* ctx, const fm_bank* q, const fm_bank* t, double tau, int64_t cap,
                             int32_t* qidx, int32_t* tidx, float* dist, double* ratio,
                             int64_t* n_accepted /*page-locked*/);

/* Completion points for the enqueue-only calls.  fm_mark remembers everything enqueued on the context
 * so far and returns a ticket; fm_wait(ticket) blocks until that work is complete, while work enqueued
 * after the mark keeps running -- the results of batch i can be consumed while batch i + 1 computes
 * (double-buffered outputs).  fm_sync waits for everything.  Tickets are valid for the next 8 marks.  */
int  fm_mark(fm_ctx* ctx, int64_t* ticket);
int  fm_wait(fm_ctx* ctx, int64_t ticket);

/* n independent image pairs in one call, enqueued like n fm_match_accepted_async calls (same output
 * rules: every qidx[i] / tidx[i] / dist[i] / ratio[i] / n_accepted[i] page-locked, results valid after
 * fm_sync).  Consecutive pairs go through the distance kernel TOGETHER, up to eight pairs per launch (option
 * "batch_group": up to sixteen) -- r05: of ANY sizes (a dataset's images all differ; before, only pairs of equal padded sizes
 * shared a launch and the others cost 7 % more per descriptor pair), small ones included (a pair whose train bank has fewer than
 * 32768 rows is planned with 4-wave workgroups when it runs alone; inside a batch it is planned again in the batched kernel's
 * 8-wave shape, the other pairs fill the chip): inside one launch the
 * workgroups of the next pair fill the CUs the previous pair leaves, where separate launches drain the chip and pay a launch gap (~4 % of
 * a 100k x 100k pair).  The reference maps its matcher over the pairs of a dataset one after the
 * other (turntable.py:59); this is that loop as one call.  Pairs that cannot be grouped (an empty bank, a query
 * bank without self distances, an option that forces another kernel shape) are enqueued one by one; pairs on the float32 route, which has no enqueue-only form,
 * run synchronously in their place (their outputs are complete when the call returns, the pairs around
 * them stay asynchronous).  Every pair is validated before anything is enqueued.  A run of pairs ends with
 * a short launch (2 pairs) because only the LAST launch's small kernels are exposed to a caller that
 * synchronises after the call; a caller that enqueues the next batch first (fm_mark / fm_wait) sets the
 * options batch_tail = 0 and batch_group = 16 (fm_ctx_set_option) for one launch per run.              */
int  fm_match_accepted_batch(fm_ctx* ctx, int32_t n, const fm_bank* const* q, const fm_bank* const* t, double tau,
                             int64_t cap, int32_t* const* qidx, int32_t* const* tidx, float* const* dist,
                             double* const* ratio, int64_t* const* n_accepted /*page-locked words*/);

/* fm_match_accepted_batch with the results left on the device, as fm_match_accepted_dev_async leaves
 * them: pair i's rows at d_rows + i * cap * 3 (int32 [n][cap][3]), its count at d_counts[i] (and in the
 * page-locked word h_counts[i] if h_counts is not NULL).  One contiguous block per call, so the caller
 * ships a whole step with ONE all-gather of n * cap rows.  consumer_stream as in
 * fm_match_accepted_dev_async: ordered against the fills in both directions.                     */
int  fm_match_accepted_dev_batch(fm_ctx* ctx, int32_t n, const fm_bank* const* q, const fm_bank* const* t, double tau,
                                 int64_t cap, int32_t* d_rows /*device [n][cap][3]*/, int64_t* d_counts /*device [n]*/,
                                 int64_t* h_counts /*page-locked [n] or NULL*/, void* consumer_stream /*hipStream_t or FM_NO_STREAM*/);

/* As fm_match_accepted, but the accepted matches stay on the device: d_rows[i] = {query index,
 * train index, float32 distance bits} (12-byte rows, ascending query index, at most cap of
 * them) and *d_count = the number of rows written = min(accepted, cap), both in caller-supplied DEVICE
 * memory (n_accepted / h_count on the host receive the full number of accepted matches) -- the send
 * buffer of the multi-GPU result gather (fm_gather_matches / an RCCL all-gather), so nothing
 * bounces through the host.  The reference has no counterpart (single process); the rows are
 * the (queryIdx, trainIdx, distance) of the DMatch list fastmatch.pyx:161-165 consumes.
 * Synchronous: the buffers are complete on return.  n_accepted (host) may be NULL.          */
int  fm_match_accepted_dev(fm_ctx* ctx, const fm_bank* q, const fm_bank* t, double tau, int64_t cap,
                           int32_t* d_rows /*device [cap][3]*/, int64_t* d_count /*device*/,
                           int64_t* n_accepted /*host, or NULL*/);

/* fm_match_accepted_dev without the synchronisation: the call enqueues and returns.  The match
 * kernel (K1) goes to the context's stream, the small kernels behind it (election, ratio test,
 * compaction into d_rows / d_count) to a second stream, where they overlap the NEXT pair's K1.
 * consumer_stream is the stream that will read d_rows / d_count -- the stream the caller's all-gather
 * is enqueued on; NULL is the null stream (PyTorch's default stream), FM_NO_STREAM means none: the compaction first waits for the work that
 * stream has been given so far (the collective that last read these buffers), and the stream is
 * made to wait for the compaction, so the caller needs no host synchronisation between pairs.
 * h_count: page-locked host word (fm_host_alloc) that also receives the count, or NULL.
 * Integer-valued banks only.  fm_gather_matches called next follows the same ordering.         */
int  fm_match_accepted_dev_async(fm_ctx* ctx, const fm_bank* q, const fm_bank* t, double tau, int64_t cap,
                                 int32_t* d_rows /*device [cap][3]*/, int64_t* d_count /*device*/,
                                 int64_t* h_count /*page-locked or NULL*/, void* consumer_stream /*hipStream_t or FM_NO_STREAM*/);

/* ---- K4: many match_position rounds in one launch ------------------------------------
 * Round b matches the query rows  q_rows[q_off[b] .. q_off[b+1])  of bank q (the radius
 * subset Metric_Cache.get returns, cache.pyx:173-188, in its order) against the train
 * rows [t_off[b], t_off[b+1]) of bank t (cells of a Grid_Cache packed back to back).
 * Output is per query slot i in [0, q_off[B]):  tidx[i] = train row index LOCAL to the
 * round's cell (-1 = none), dist[i], and ratio[i] = dist / selfdist[q_rows[i]] when q
 * carries self distances (else nan).  Semantics per round are exactly fm_xcheck1 on the
 * gathered sub-matrices.  Both routes: integer-valued banks run the int8 round; float32 banks
 * the float32 round (fp16 MFMA filter + exact float32 chain, bit-identical to fm_xcheck1); a
 * float32 round that cannot be completed on the device reports tidx = -2 for all its slots
 * (redo it with fm_xcheck1 on gathered banks).  At most 4096 query rows per round.        */
int  fm_xcheck1_batched(fm_ctx* ctx, const fm_bank* q, const int32_t* q_rows,
                        const int64_t* q_off /*[B+1]*/, const fm_bank* t,
                        const int64_t* t_off /*[B+1]*/, int64_t n_rounds,
                        int32_t* tidx, float* dist, double* ratio /*or NULL*/);

/* ---- K7: device-resident expansion loop ------------------------------------------------
 * Replaces the whole of do_iter + get_neighbors + match_position (fastmatch.pyx:56-103,
 * 145-169) for an image pair whose features are all known up front: the query bank with its
 * keypoint positions and position index (Metric_Cache.original, cache.pyx:278-284) and
 * every Grid_Cache cell's descriptors packed back to back (cache.pyx:124-138).  One
 * persistent workgroup per pair replays the reference's depth-first order exactly; many
 * pairs run concurrently in one launch.  Results come back in discovery order.  The two banks
 * must be of one kind: integer valued (int8 round) or float32 (float32 round).            */
typedef struct fm_expand fm_expand;

typedef struct fm_expand_desc {
    const fm_bank* query;          /* query bank carrying self distances                    */
    const double*  query_pos;      /* [nq][2] keypoint positions (x, y)                     */
    double         index_bucket;   /* uniform-grid position index over query_pos:           */
    double         index_x0, index_y0;   /*   bucket size and origin                         */
    int32_t        index_nbx, index_nby; /*   buckets along x / y                            */
    const int32_t* index_order;    /* [nq] keypoints sorted by bucket (by*nbx + bx)         */
    const int32_t* index_start;    /* [nbx*nby + 1] first entry of each bucket in order[]   */
    const fm_bank* target;         /* all cells' descriptors, cell after cell               */
    const int64_t* cell_off;       /* [cols*rows + 1] row range of cell id = col*rows + row */
    const double*  target_pos;     /* [nt][2] full-image keypoint positions (offset applied)*/
    int32_t        width, height;  /* target image size                                     */
    int32_t        cell_w, cell_h; /* Grid_Cache cell size                                  */
    int32_t        rows, cols;     /* Grid_Cache.rows (cells along x), .cols (along y)      */
    int32_t        margin, radius;
    int64_t        match_cap;      /* first capacity of the result list (0 = 4 * nq) and    */
    int64_t        stack_cap;      /* of the pending stack (0 = default); fm_expand_run     */
                                   /* repeats a run that fills one in a state 4x as large   */
    int32_t        metric;         /* radius query metric: the reference builds its BallTree with
                                    * options["metric"] (cache.pyx:160, 276): FM_METRIC_*    */
    int32_t        lazy;           /* non-zero: LAZY TARGET -- the reference's own mode (cache.pyx:102-106, 124-138: a cell's
                                    * features are computed when the loop first reaches it).  `target` is a bank made with
                                    * fm_bank_create_u8_cap (possibly empty), cell_off / target_pos are ignored; cells are
                                    * added with fm_bank_append_u8 + fm_expand_set_cell and the pair is driven with
                                    * fm_expand_run_lazy.  Integer-route banks, or (r05) float32-route banks:
                                    * target from fm_bank_create_f32_cap, cells added with fm_bank_append_f32.          */
} fm_expand_desc;

#define FM_METRIC_EUCLIDEAN 0      /* "minkowski" (p = 2), "euclidean": dx^2 + dy^2 <= r^2     */
#define FM_METRIC_MANHATTAN 1      /* |dx| + |dy| <= r                                          */
#define FM_METRIC_CHEBYSHEV 2      /* max(|dx|, |dy|) <= r                                      */

#define FM_EXPAND_OK            0
#define FM_EXPAND_STACK_FULL    1
#define FM_EXPAND_SUBSET_FULL   2  /* a radius subset the device could not take.  Integer-route pairs that exceed the first
                                      * kernel's 2048 rows are re-run by fm_expand_run in a 4096-row kernel and then in one that
                                      * takes a subset of ANY size in chunks (options expand_big / expand_huge; float32 pairs go from
                                      * 2048 straight to the chunked kernel; r05: pairs under the float32-root guard and subsets with
                                      * thousands of keypoints at one distance too, and a round may ACCEPT any number); what is left:
                                      * a subset beyond 640 x 2048 rows, or expand_big / expand_huge switched off */
#define FM_EXPAND_OUT_OF_BOUNDS 3  /* a target position outside the image (cache.pyx:56-57) */
#define FM_EXPAND_MATCH_FULL    4  /* 1, 4, 5: reported only when the run state could not grow any further */
#define FM_EXPAND_TABLE_FULL    5
#define FM_EXPAND_LIST_FULL     6  /* float32 round: more candidates inside the fp16 margin than fit  */
#define FM_EXPAND_NEED_CELL     7  /* lazy target: the loop reached a cell that has not been added yet (fm_expand_run_lazy) */
#define FM_EXPAND_LOG_FULL      9  /* the per-round log's arrays filled and could not grow any further (fm_expand_set_log);
                                      * (8 is internal: a round parked for the dense kernels, settled inside the call) */

/* LIFETIME: the pair BORROWS desc->query and desc->target -- device arrays and, for delegated rounds and lazy
 * targets, the banks' host-side descriptors -- so both banks must outlive the fm_expand, and a bank that an fm_expand names
 * must not be refilled (fm_bank_refill_u8_async) while it exists: row counts, cell offsets and the float32-root guard were
 * fixed when the pair was created.  (The Python binding keeps references; a C client keeps the order destroy(pair), then banks.) */
int  fm_expand_create(fm_ctx* ctx, const fm_expand_desc* desc, fm_expand** out);
int  fm_expand_destroy(fm_ctx* ctx, fm_expand* ex);
/* Run n expansions in one launch, one workgroup each.  Run i = (pairs[i], seeds[i], tau[i]): seeds[i] =
 * [n_seeds[i]][2][2] float64 (query_pos, target_pos) in visiting order (fastmatch.pyx:50), tau[i] the
 * ratio threshold.  A pair may appear SEVERAL times -- the reference is driven as pairs x thresholds
 * (turntable.py:59-60: { tau : f(tau) for tau in thresholds } per pair) and the runs of one pair are
 * independent of each other: each gets a run state of its own (pending stack, seen / found tables,
 * result list; created on first use, kept), the k-th appearance of a pair in a launch uses its run
 * slot k.  Per run outputs: number of matches, rounds, descriptor pairs evaluated, and a FM_EXPAND_*
 * status (non-zero = the device gave up; the caller falls back to the host loop).                */
int  fm_expand_run(fm_ctx* ctx, int32_t n, fm_expand* const* pairs, const double* const* seeds,
                   const int64_t* n_seeds, const double* tau, int64_t* n_matches,
                   int64_t* n_rounds, int64_t* n_pairs, int32_t* status);
/* ---- lazy targets: the device loop with cells computed on demand (r04) ---------------------------------------------
 * fastmatch.pyx:154 -> cache.pyx:102-106, 124-138: a grid cell's features are computed (SIFT on the cell's crop) the first
 * time the expansion reaches it, so most cells of a large image are never computed.  With a lazy pair the loop still runs on
 * the device: a round that needs a missing cell parks the loop's state and ends the launch with FM_EXPAND_NEED_CELL; the host
 * computes that cell, appends its descriptors to the target bank, registers it and resumes.
 *   fm_bank_create_u8_cap : a bank with room for `capacity` rows (n of them now, possibly 0).
 *   fm_bank_append_u8     : n more rows at the next multiple of 32 rows (*first_row); FM_EINVAL when the capacity is used up.
 *                           The rows skipped in between are padding rows (never a nearest neighbour); a grown bank is meant for
 *                           fm_expand (cells name their own row ranges) -- as the TRAIN side of a dense call they would be
 *                           output rows like any other.
 *   fm_expand_set_cell    : cell (= col * rows + row) := rows [first_row, first_row + n_rows) with their full-image positions;
 *                           n_rows = 0 for a cell without features.
 *   fm_expand_run_lazy    : one run in slot 0, from the start (resume = 0) or from where the last launch parked (resume != 0:
 *                           FM_EINVAL unless that launch ended with FM_EXPAND_NEED_CELL; its seeds are re-used, `seeds` is ignored).
 *                           status 0: done (fm_expand_fetch); FM_EXPAND_NEED_CELL: *need_cell; else the device gave up.
 *                           (Rounds of >= "expand_delegate" descriptor pairs are cross-checked by the dense kernels inside the call.) */
int  fm_bank_create_u8_cap(fm_ctx* ctx, const uint8_t* rows, int64_t n, int dim, int64_t capacity, fm_bank** bank);
int  fm_bank_append_u8(fm_ctx* ctx, fm_bank* bank, const uint8_t* rows, int64_t n, int64_t* first_row);
/* r05, the same for descriptors that are NOT integer valued (RootSIFT-style float32 on a pixel target): an empty float32-route
 * bank with room for `capacity` rows whose fp16 planes use the power-of-two scale of `scale_like` (the query bank the target
 * will be matched against: a growing bank cannot derive a scale from rows it has not seen), and n more rows at the next
 * multiple of 32.  fm_bank_append_f32 returns FM_EUNSUPPORTED -- bank unchanged -- for a value that is not finite or that
 * leaves fp16's range under that scale (more than ~4 x the largest magnitude of scale_like).                              */
int  fm_bank_create_f32_cap(fm_ctx* ctx, int dim, int64_t capacity, const fm_bank* scale_like, fm_bank** bank);
int  fm_bank_append_f32(fm_ctx* ctx, fm_bank* bank, const float* rows, int64_t n, int64_t* first_row);
int  fm_expand_set_cell(fm_ctx* ctx, fm_expand* ex, int32_t cell, int64_t first_row, int64_t n_rows, const double* pos /*[n_rows][2]*/);
int  fm_expand_run_lazy(fm_ctx* ctx, fm_expand* ex, const double* seeds, int64_t n_seeds, double tau, int32_t resume,
                        int64_t* n_matches, int64_t* n_rounds, int64_t* n_pairs, int32_t* status, int32_t* need_cell);

/* ---- pre-extracted targets: all cells of a Grid_Cache at once (r04) -------------------------------------------------
 * The device loop's target bank holds "every cell's descriptors, cell after cell" (fm_expand_desc.target / cell_off /
 * target_pos).  For a target given as pre-extracted features (keypoint positions + descriptors instead of pixels) a
 * cell's features are the keypoints inside the crop Grid_Cache hands its caching function (cache.pyx:128-131: x from
 * row * cell_w - margin, clipped at 0, over cell_w + 2 * margin; same along y) -- a keypoint lands in up to four cells.
 *   fm_grid_pack_cells       : HOST code, no context: cell_off[rows * cols + 1] (cell id = col * rows + row) and *n_rows
 *                              always; when `capacity` >= *n_rows also src_row[n_rows] (the keypoint of every packed row,
 *                              ascending inside a cell) and target_pos[n_rows][2] (crop-local position + the offset
 *                              match_position adds, fastmatch.pyx:157-158).  A caller whose capacity was too small
 *                              reads *n_rows and calls again.
 *   fm_bank_create_u8_gather /
 *   fm_bank_create_f32_gather: a bank whose row i is rows[src_row[i]] -- the n_src descriptors cross PCIe once and are
 *                              gathered by the upload kernel.  FM_EINVAL for an entry outside [0, n_src).  float_route
 *                              as fm_bank_create_f32_route (non-zero: keep the float32 route even if integer valued).  */
int  fm_grid_pack_cells(const double* positions /*[n][2]*/, int64_t n, int32_t width, int32_t height, int32_t cell_w, int32_t cell_h,
                        int32_t rows, int32_t cols, int32_t margin, int64_t capacity, int64_t* cell_off, int64_t* n_rows,
                        int32_t* src_row, double* target_pos);
int  fm_bank_create_u8_gather(fm_ctx* ctx, const uint8_t* rows, int64_t n_src, int dim, const int32_t* src_row, int64_t n, fm_bank** bank);
int  fm_bank_create_f32_gather(fm_ctx* ctx, const float* rows, int64_t n_src, int dim, int float_route, const int32_t* src_row,
                               int64_t n, fm_bank** bank);

/* Memory of the run states: fm_expand_info reports the bytes ONE run state of the pair takes (pending stack, seen /
 * found tables, result arrays: ~210 MB for a 300k-keypoint pair) and how many exist; fm_expand_trim frees the states
 * from slot `keep` (>= 1) on; fm_mem_info is hipMemGetInfo of the context's device.  fm_expand_run creates a state for
 * every run of a launch and keeps it, so a caller that puts pairs x thresholds into one launch (turntable.py:59-60)
 * sizes its launches with these (fastmatch.run_device_loops does) instead of meeting FM_ENOMEM.                    */
int  fm_expand_info(const fm_expand* ex, int64_t* state_bytes, int32_t* n_slots);
int  fm_expand_trim(fm_ctx* ctx, fm_expand* ex, int32_t keep);
int  fm_mem_info(fm_ctx* ctx, int64_t* free_bytes, int64_t* total_bytes);
/* Copy the first n results of the last run in slot 0 of `ex`: query row index, positions [n][2][2]
 * (query x,y then target x,y) and ratio -- the tuples do_iter appends (fastmatch.pyx:86). */
int  fm_expand_fetch(fm_ctx* ctx, const fm_expand* ex, int64_t n, int32_t* index,
                     double* positions, double* ratio);
/* fm_expand_fetch for several runs of one fm_expand_run with a single synchronisation: n[i] results of
 * run slot slot[i] of ex[i] into index[i] / positions[i] / ratio[i] (any of the three arrays of
 * pointers, or single entries, may be NULL).  slot == NULL: slots by appearance, as fm_expand_run
 * assigns them (the k-th entry naming a pair reads its slot k).                                    */
int  fm_expand_fetch_many(fm_ctx* ctx, int32_t n_ex, const fm_expand* const* ex, const int32_t* slot, const int64_t* n,
                          int32_t* const* index, double* const* positions, double* const* ratio);

/* ---- the per-round log on the device (r05) ------------------------------------------------------------------------------
 * options["log"] of fastmatch.match (fastmatch.pyx:46, 79-80): do_iter appends log_round(...) (fastmatch.pyx:172-180) once per
 * processed round -- query_pos, target_pos, Grid_Cache.last, result_pos[ratios < tau], radius, ratios[ratios < tau], margin.
 * With fm_expand_set_log(enable != 0) the runs of the pair (fm_expand_run and fm_expand_run_lazy) record, per processed round
 * in order, rounds[i][0..3] = the popped (query_pos, target_pos) as float64 bit patterns, rounds[i][4] = the cell the round
 * fetched (col * rows + row: Grid_Cache.last is the crop of the most recently COMPUTED cell, cache.pyx:102-106, which the
 * caller derives from the order in which cells first appear), rounds[i][5] = its accepted matches (-1: the cell has no
 * features, -2: no pair passed the cross-check, an empty radius subset included -- in both cases match_position's arrays come
 * from an empty list and have shape (0,), not (0, 2, 2): fastmatch.pyx:155-156, 162-167); and, for every accepted
 * match in the round's order BEFORE the result dedup (fastmatch.pyx:82-86 dedups the matches, not the log), its query row, its
 * row of the (packed / growing) target bank and its float64 ratio.  fm_expand_log_counts gives the sizes of the last run in
 * `slot`, fm_expand_fetch_log copies them (any pointer may be NULL).  The arrays grow fourfold up to 12 times (when
 * there is no memory for larger ones the old ones stand); a run that still overflows ends with FM_EXPAND_LOG_FULL.  enable > 1: on, with `enable` records and
 * `enable` accepted matches as the arrays' FIRST capacity (the defaults: 4 x the grid's cells, 4 x the query's keypoints).  */
int  fm_expand_set_log(fm_ctx* ctx, fm_expand* ex, int32_t enable);
int  fm_expand_log_counts(fm_ctx* ctx, const fm_expand* ex, int32_t slot, int64_t* n_rounds, int64_t* n_entries);
int  fm_expand_fetch_log(fm_ctx* ctx, const fm_expand* ex, int32_t slot, int64_t n_rounds, int64_t n_entries,
                         int64_t* rounds /*[n_rounds][6]*/, int32_t* query_row /*[n_entries]*/, int32_t* target_row /*[n_entries]*/,
                         double* ratio /*[n_entries]*/);

/* ---- result gather across the GPUs of a node (RCCL over xGMI) --------------------------------
 * Independent image pairs are sharded over ranks (one process per GPU, pair i -> rank i mod N);
 * matching needs no communication and the per-pair match lists come back with ONE exchange: an
 * all-gather of every rank's accepted rows and their counts.  The reference is single process
 * (turntable.py:59 maps the matcher over its pairs sequentially); this is the data-parallel axis
 * it implies.  RCCL is bound at run time (dlopen), so a process that never gathers needs none.
 *   fm_comm_unique_id : rank 0 draws the 128-byte id; the launcher hands it to every rank.
 *   fm_comm_init      : collective; binds the context's device and stream to the communicator.
 *   fm_gather_matches : d_rows / d_count as fm_match_accepted_dev leaves them (device memory);
 *                       d_all_rows [nranks][cap][3] int32, d_all_counts [nranks] int64 (device).
 *                       Runs on the context's stream behind the matching kernels; wait != 0
 *                       synchronises before returning, wait == 0 returns at once (fm_sync later).  */
int  fm_comm_unique_id(void* id128);
int  fm_comm_init(fm_ctx* ctx, int nranks, int rank, const void* id128);
int  fm_comm_destroy(fm_ctx* ctx);
int  fm_gather_matches(fm_ctx* ctx, const int32_t* d_rows, const int64_t* d_count, int64_t cap,
                       int32_t* d_all_rows, int64_t* d_all_counts, int wait);
/* Counts first, then only *rows_per_rank = min(cap, max over the ranks of their counts) rows per rank:
 * d_all_rows is laid out [nranks][*rows_per_rank][3].  Ships what is there instead of the padded capacity
 * (xGMI traffic / 3 on the bench's workload) for one host synchronisation between the two collectives;
 * returns when both are done.                                                                          */
int  fm_gather_matches_counted(fm_ctx* ctx, const int32_t* d_rows, const int64_t* d_count, int64_t cap,
                               int32_t* d_all_rows, int64_t* d_all_counts, int64_t* rows_per_rank);

#ifdef __cplusplus
}
#endif
#endif /* FASTMATCH_HIP_H */
