/*
 * oracle/bfmatch_oracle.c -- CPU restatement of the arithmetic on Fast-Match's
 * descriptor-matching hot path.  TEST INFRASTRUCTURE ONLY: nothing in the product
 * path (fast-match_amd/) may call, link or import this file; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * PARITY UNPINNED: the reference has no tests / golden vectors for this path and
 * its arithmetic lives in an un-vendored, un-pinned third-party dependency
 * (OpenCV cv2.BFMatcher).  The reference's own .so files are
 * CPython-2.7 modules that cannot be loaded here.  This file restates OpenCV's
 * published BFMatcher::knnMatch / cv::batchDistance algorithm (SURVEY.md
 * Appendix A) as used at the reference's call sites:
 *
 *   fastmatch.pyx:122-123, 161-162   BFMatcher(NORM_L2, crossCheck=True).knnMatch(q, t, k=1)
 *   matchutil.py:39-43 (cache.pyx:250) BFMatcher(NORM_L2, False).knnMatch(d, d, k=2)
 *   fastmatch.pyx:124, 165           ratio = m.distance / query_dis[m.queryIdx]     (float64)
 *   fastmatch.pyx:50, 75, 82         accepted = ratio < tau                          (float64)
 *   Classic Matching.ipynb cell 3    knnMatch(q, t, k=2); m[0].distance / m[1].distance
 *
 * It is anchored by hand-derived known-answer tests (tests/test_oracle_kat.py)
 * and by the Grid_Cache golden vectors generated from the importable
 * bak/cache.py (tests/golden/).
 *
 * Build: see oracle/Makefile  (gcc -O3 -march=x86-64-v3 -fopenmp -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* ---- distance kernels ---------------------------------------------------- */

/* OpenCV normL2Sqr<float,float>: squared differences accumulated in float32.
 * Two accumulation orders are restated:
 *   order 0 ("unrolled4"): OpenCV's generic C++ loop, s += v0*v0+v1*v1+v2*v2+v3*v3
 *                           per group of four (core/stat.cpp, CV_ENABLE_UNROLLED);
 *   order 1 ("fma chain"): s = fmaf(v_k, v_k, s), k ascending.  This is the order
 *                           our device fp32 kernel executes (v_sub_f32 + v_fma_f32),
 *                           fixed so that non-integer descriptors are bit-comparable.
 *   order 2 ("sse2x4"):    OpenCV 2.4.x's SSE2 path of normL2Sqr_ (core/src/stat.cpp, the
 *                           build the reference's 2014 .so files were linked against):
 *                           two 4-lane accumulators over strides of 8, d0 += t0*t0,
 *                           d1 += t1*t1 (mul then add, no fma), lanes of d0+d1 summed
 *                           left to right.  Recalled from the OpenCV sources, like
 *                           Appendix A; not verified against a live cv2.
 *   order 3 ("simd4x4"):   OpenCV 3.4/4.x universal-intrinsics path on a 128-bit
 *                           baseline (core/src/norm.cpp): four 4-lane accumulators over
 *                           strides of 16, v_muladd without FMA3 = mul then add,
 *                           ((d0+d1)+d2)+d3 lane-wise, then (l0+l2)+(l1+l3).  Recalled.
 * For integer-valued inputs 0..255 (what OpenCV SIFT emits) every partial sum is an
 * exact integer <= 8 323 200 < 2^24, so all orders (and any SIMD order OpenCV may
 * use) give the same bits.                                       */
static inline float l2sqr_f32_unrolled4(const float* a, const float* b, int n)
{
    float s = 0.f;
    int i = 0;
    for (; i <= n - 4; i += 4) {
        float v0 = a[i] - b[i], v1 = a[i + 1] - b[i + 1];
        float v2 = a[i + 2] - b[i + 2], v3 = a[i + 3] - b[i + 3];
        s += v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3;
    }
    for (; i < n; i++) { float v = a[i] - b[i]; s += v * v; }
    return s;
}

static inline float l2sqr_f32_fmachain(const float* a, const float* b, int n)
{
    float s = 0.f;
    for (int i = 0; i < n; i++) { float v = a[i] - b[i]; s = fmaf(v, v, s); }
    return s;
}

static inline float l2sqr_f32_sse2x4(const float* a, const float* b, int n)
{
    float d0[4] = {0.f, 0.f, 0.f, 0.f}, d1[4] = {0.f, 0.f, 0.f, 0.f};
    int j = 0;
    for (; j <= n - 8; j += 8) {
        for (int l = 0; l < 4; l++) {
            float t0 = a[j + l] - b[j + l], t1 = a[j + 4 + l] - b[j + 4 + l];
            float p0 = t0 * t0, p1 = t1 * t1;
            d0[l] = d0[l] + p0;
            d1[l] = d1[l] + p1;
        }
    }
    float buf[4];
    for (int l = 0; l < 4; l++) buf[l] = d0[l] + d1[l];
    float d = buf[0] + buf[1];
    d = d + buf[2];
    d = d + buf[3];
    for (; j < n; j++) { float t = a[j] - b[j]; d += t * t; }
    return d;
}

static inline float l2sqr_f32_simd4x4(const float* a, const float* b, int n)
{
    float acc[4][4] = {{0.f}};
    int j = 0;
    for (; j <= n - 16; j += 16)
        for (int v = 0; v < 4; v++)
            for (int l = 0; l < 4; l++) {
                float t = a[j + 4 * v + l] - b[j + 4 * v + l];
                float p = t * t;
                acc[v][l] = p + acc[v][l];
            }
    float s[4];
    for (int l = 0; l < 4; l++) { float x = acc[0][l] + acc[1][l]; x = x + acc[2][l]; s[l] = x + acc[3][l]; }
    float d = (s[0] + s[2]) + (s[1] + s[3]);
    for (; j < n; j++) { float t = a[j] - b[j]; d += t * t; }
    return d;
}

/* OpenCV normL2Sqr<uchar,float> equivalent for CV_8U inputs: integer exact.        */
static inline float l2sqr_u8(const uint8_t* a, const uint8_t* b, int n)
{
    int32_t s = 0;
    for (int i = 0; i < n; i++) { int32_t v = (int32_t)a[i] - (int32_t)b[i]; s += v * v; }
    return (float)s;
}

/* kind: 0 = f32 unrolled4, 1 = f32 fma chain, 2 = f32 sse2x4, 3 = f32 simd4x4, 8 = u8 */
#define ORC_KIND_U8 8
#define ORC_ORDER_OK(o) ((o) >= 0 && (o) <= 3)
typedef struct { const void* base; int dim; int kind; } orc_mat;

static inline float dist_row(const orc_mat* A, int64_t i, const orc_mat* B, int64_t j)
{
    float d2;
    if (A->kind == ORC_KIND_U8)
        d2 = l2sqr_u8((const uint8_t*)A->base + i * A->dim, (const uint8_t*)B->base + j * B->dim, A->dim);
    else if (A->kind == 1)
        d2 = l2sqr_f32_fmachain((const float*)A->base + i * A->dim, (const float*)B->base + j * B->dim, A->dim);
    else if (A->kind == 2)
        d2 = l2sqr_f32_sse2x4((const float*)A->base + i * A->dim, (const float*)B->base + j * B->dim, A->dim);
    else if (A->kind == 3)
        d2 = l2sqr_f32_simd4x4((const float*)A->base + i * A->dim, (const float*)B->base + j * B->dim, A->dim);
    else
        d2 = l2sqr_f32_unrolled4((const float*)A->base + i * A->dim, (const float*)B->base + j * B->dim, A->dim);
    return sqrtf(d2);      /* batchDistance: dist = std::sqrt(normL2Sqr) -- Appendix A.1 */
}

/* ---- k-NN insertion (cv::batchDistance, Appendix A.2) -------------------- */
/* For query row i scan train rows j ascending; dist[] pre-filled FLT_MAX, idx[] -1.
 *   if (d < dist[K-1]) { k = K-2; while (k >= 0 && dist[k] > d) shift; insert at k+1 }
 * Strict '<' on entry and strict '>' while shifting: among equal distances the lower
 * train index stays first.                                                          */
static void knn_rows(const orc_mat* Q, int64_t nq, const orc_mat* T, int64_t nt, int K,
                     int32_t* idx, float* dist, int threads)
{
    /* small problems (expansion rounds are ~400 x 125) run on one thread: a fork-join
     * over every host core costs more than the work itself                              */
    {
        int64_t cap = (nq * nt) / 400000 + 1;
        if (cap < threads) threads = (int)cap;
        if (threads < 1) threads = 1;
    }
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int64_t i = 0; i < nq; i++) {
        float* bd = dist + i * K;
        int32_t* bi = idx + i * K;
        for (int k = 0; k < K; k++) { bd[k] = FLT_MAX; bi[k] = -1; }
        for (int64_t j = 0; j < nt; j++) {
            float d = dist_row(Q, i, T, j);
            if (d < bd[K - 1]) {
                int k = K - 2;
                for (; k >= 0 && bd[k] > d; k--) { bd[k + 1] = bd[k]; bi[k + 1] = bi[k]; }
                bd[k + 1] = d;
                bi[k + 1] = (int32_t)j;
            }
        }
        /* entries with idx < 0 are dropped by OpenCV (inner list shorter than k);
         * we report them as idx -1 / dist +inf                                       */
        for (int k = 0; k < K; k++) if (bi[k] < 0) bd[k] = INFINITY;
    }
}

static orc_mat mk(const void* p, int dim, int kind) { orc_mat m; m.base = p; m.dim = dim; m.kind = kind; return m; }

static int nthreads(int threads)
{
#ifdef _OPENMP
    return threads > 0 ? threads : omp_get_max_threads();
#else
    (void)threads; return 1;
#endif
}

/* cv2.BFMatcher(NORM_L2, crossCheck=False).knnMatch(Q, T, k)  -- matchutil.py:39-43 */
ORC_API int orc_bf_knn_f32(const float* Q, int64_t nq, const float* T, int64_t nt, int dim, int k,
                           int order, int32_t* idx, float* dist, int threads)
{
    if (k < 1 || dim < 1 || !ORC_ORDER_OK(order)) return -1;
    orc_mat q = mk(Q, dim, order), t = mk(T, dim, order);
    knn_rows(&q, nq, &t, nt, k, idx, dist, nthreads(threads));
    return 0;
}

ORC_API int orc_bf_knn_u8(const uint8_t* Q, int64_t nq, const uint8_t* T, int64_t nt, int dim, int k,
                          int32_t* idx, float* dist, int threads)
{
    if (k < 1 || dim < 1) return -1;
    orc_mat q = mk(Q, dim, ORC_KIND_U8), t = mk(T, dim, ORC_KIND_U8);
    knn_rows(&q, nq, &t, nt, k, idx, dist, nthreads(threads));
    return 0;
}

/* cv2.BFMatcher(NORM_L2, crossCheck=True).knnMatch(Q, T, k=1) -- Appendix A.3:
 *   batchDistance(T, Q, k=1): for each train row i, tidx[i] = argmin_q d(q, i)
 *   (lowest q on ties), tdist[i];
 *   then dist[q] = FLT_MAX, nidx[q] = -1;
 *   for i in 0..nt-1: q = tidx[i]; if tdist[i] < dist[q]: dist[q] = tdist[i]; nidx[q] = i
 * Output per query row: train index or -1 (empty inner list, filtered out by the
 * reference at fastmatch.pyx:123,162), and the distance (+inf when unmatched).       */
static void xcheck(const orc_mat* Q, int64_t nq, const orc_mat* T, int64_t nt,
                   int32_t* tidx_out, float* dist_out, int threads)
{
    int32_t* rq = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nt > 0 ? nt : 1));
    float* rd = (float*)malloc(sizeof(float) * (size_t)(nt > 0 ? nt : 1));
    knn_rows(T, nt, Q, nq, 1, rq, rd, threads);   /* reverse NN: rows = train, scan = query */
    for (int64_t q = 0; q < nq; q++) { dist_out[q] = FLT_MAX; tidx_out[q] = -1; }
    for (int64_t i = 0; i < nt; i++) {
        int32_t q = rq[i];
        if (q < 0) continue;
        if (rd[i] < dist_out[q]) { dist_out[q] = rd[i]; tidx_out[q] = (int32_t)i; }
    }
    for (int64_t q = 0; q < nq; q++) if (tidx_out[q] < 0) dist_out[q] = INFINITY;
    free(rq); free(rd);
}

/* ---- the same cross-check, vectorised (a second CPU baseline; results identical) ----------------------
 * The loops above are the faithful restatement: one pair at a time, ~1 multiply-add per cycle and thread, every
 * output row streaming the whole other bank through the caches.  A CPU would not be programmed that way either, so
 * bench.py reports a second figure from this form: uint8 rows widened to int16, differences squared and summed with
 * vpmaddwd (AVX2: 16 products per instruction), eight output rows per pass over the other bank.  Same semantics,
 * bit for bit: candidates in ascending index order, compared on the float32 root with a strict '<' (the integer test
 * d2 <= best d2 in front of it only skips pairs that cannot pass: sqrtf is monotone).                                */
#if defined(__AVX2__)
#include <immintrin.h>
#define ORC_SIMD_BLOCK 8
static void knn1_rows_u8_simd(const uint8_t* A, int64_t na, const uint8_t* B, int64_t nb, int dim,
                              int32_t* idx, float* dist, int threads)
{
    const int dpad = (dim + 15) & ~15;
    {
        int64_t cap = (na * nb) / 400000 + 1;
        if (cap < threads) threads = (int)cap;
        if (threads < 1) threads = 1;
    }
    const int64_t nblk = (na + ORC_SIMD_BLOCK - 1) / ORC_SIMD_BLOCK;
#pragma omp parallel num_threads(threads)
    {
        int16_t* a16 = (int16_t*)aligned_alloc(32, sizeof(int16_t) * (size_t)ORC_SIMD_BLOCK * (size_t)dpad);
        int16_t* b16 = (int16_t*)aligned_alloc(32, sizeof(int16_t) * (size_t)dpad);
#pragma omp for schedule(dynamic, 4)
        for (int64_t blk = 0; blk < nblk; blk++) {
            const int64_t i0 = blk * ORC_SIMD_BLOCK;
            const int rows = (int)((na - i0) < ORC_SIMD_BLOCK ? (na - i0) : ORC_SIMD_BLOCK);
            float bd[ORC_SIMD_BLOCK];
            int32_t bi[ORC_SIMD_BLOCK], b2[ORC_SIMD_BLOCK];
            for (int r = 0; r < ORC_SIMD_BLOCK; r++) {
                bd[r] = FLT_MAX; bi[r] = -1; b2[r] = INT32_MAX;
                for (int k = 0; k < dpad; k++)
                    a16[r * dpad + k] = (r < rows && k < dim) ? (int16_t)A[(i0 + r) * dim + k] : 0;
            }
            for (int64_t j = 0; j < nb; j++) {
                {   /* the row widened to int16: 16 bytes per vpmovzxbw, the tail (dim not a multiple of 16) by hand */
                    int k = 0;
                    for (; k + 16 <= dim; k += 16)
                        _mm256_store_si256((__m256i*)(b16 + k), _mm256_cvtepu8_epi16(_mm_loadu_si128((const __m128i*)(B + j * dim + k))));
                    for (; k < dpad; k++) b16[k] = k < dim ? (int16_t)B[j * dim + k] : 0;
                }
                for (int r = 0; r < rows; r++) {
                    __m256i acc = _mm256_setzero_si256();
                    for (int k = 0; k < dpad; k += 16) {
                        const __m256i d = _mm256_sub_epi16(_mm256_load_si256((const __m256i*)(a16 + r * dpad + k)),
                                                           _mm256_load_si256((const __m256i*)(b16 + k)));
                        acc = _mm256_add_epi32(acc, _mm256_madd_epi16(d, d));
                    }
                    __m128i s = _mm_add_epi32(_mm256_castsi256_si128(acc), _mm256_extracti128_si256(acc, 1));
                    s = _mm_add_epi32(s, _mm_shuffle_epi32(s, 0x4e));
                    s = _mm_add_epi32(s, _mm_shuffle_epi32(s, 0xb1));
                    const int32_t d2 = _mm_cvtsi128_si32(s);
                    if (d2 <= b2[r]) {
                        const float d = sqrtf((float)d2);
                        if (d < bd[r]) { bd[r] = d; bi[r] = (int32_t)j; b2[r] = d2; }
                    }
                }
            }
            for (int r = 0; r < rows; r++) { idx[i0 + r] = bi[r]; dist[i0 + r] = bi[r] < 0 ? INFINITY : bd[r]; }
        }
        free(a16); free(b16);
    }
}
#endif

ORC_API int orc_have_simd(void)
{
#if defined(__AVX2__)
    return 1;
#else
    return 0;
#endif
}

/* orc_bf_xcheck1_u8 through the vectorised reverse-NN scan; -2 where the build has no AVX2. */
ORC_API int orc_bf_xcheck1_u8_simd(const uint8_t* Q, int64_t nq, const uint8_t* T, int64_t nt, int dim,
                                   int32_t* tidx_out, float* dist_out, int threads)
{
#if defined(__AVX2__)
    if (dim < 1) return -1;
    int32_t* rq = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nt > 0 ? nt : 1));
    float* rd = (float*)malloc(sizeof(float) * (size_t)(nt > 0 ? nt : 1));
    knn1_rows_u8_simd(T, nt, Q, nq, dim, rq, rd, nthreads(threads));
    for (int64_t q = 0; q < nq; q++) { dist_out[q] = FLT_MAX; tidx_out[q] = -1; }
    for (int64_t i = 0; i < nt; i++) {
        const int32_t q = rq[i];
        if (q < 0) continue;
        if (rd[i] < dist_out[q]) { dist_out[q] = rd[i]; tidx_out[q] = (int32_t)i; }
    }
    for (int64_t q = 0; q < nq; q++) if (tidx_out[q] < 0) dist_out[q] = INFINITY;
    free(rq); free(rd);
    return 0;
#else
    (void)Q; (void)nq; (void)T; (void)nt; (void)dim; (void)tidx_out; (void)dist_out; (void)threads;
    return -2;
#endif
}

/* ---- the same cross-check, blocked for the caches and the dot-product units (a third CPU baseline; results identical) ----
 * What a CPU programmer with AVX-512 VNNI would write, so that bench.py's GPU / CPU ratio has an honest denominator:
 *   d2(a, c) = |a|^2 + |c|^2 - 2 a.c,   a.c = sum a_k (c_k - 128) + 128 sum a_k  -- vpdpbusd (u8 x s8 -> int32, 64 products per
 *   instruction, no saturation) on the candidates' bytes with the top bit flipped;
 *   output rows packed 16 to a register (lane = output row, 4 consecutive bytes of it), the candidate's 4 bytes broadcast:
 *   one vpdpbusd advances 16 pairs by 4 dimensions, nothing is summed across lanes;
 *   register block: 64 output rows x 6 candidates = 24 accumulators; a block's 64 rows stream the candidate bank once;
 *   the reduction keeps (d2, float32 root, index) of the best candidate per lane: an integer compare d2 <= best d2 per
 *   accumulator, and only where a lane passes the float32 roots are formed and compared with the strict '<' of the scan
 *   above -- candidates in ascending index order, so the same winner, bit for bit (tests/test_oracle_kat.py).
 * Compiled for the instruction set by a target attribute, chosen at run time (orc_have_vnni).                            */
#if defined(__x86_64__)
#include <immintrin.h>
#define ORC_VNNI_TARGET __attribute__((target("avx512f,avx512bw,avx512vl,avx512dq,avx512vnni")))
#define ORC_VG 4      /* groups of 16 output rows per block */
#define ORC_VC 6      /* candidates per block (vnni_block names its 4 x 6 accumulators: change both) */
#define ORC_UNROLL _Pragma("GCC unroll 8")

/* (no-tree-pre: with partial-redundancy elimination gcc 11 keeps a copy of every accumulator on the stack inside the loop) */
ORC_VNNI_TARGET __attribute__((optimize("no-tree-pre")))
static void vnni_block(const uint8_t* ap /* [kg][ORC_VG][64] */, const int32_t* P /* [ORC_VG][16] */, int kg,
                       const int8_t* const* cand /* ORC_VC rows, kg * 4 bytes each */, const int32_t* nc, int valid, int64_t j0,
                       int32_t* best2, float* bestd, int32_t* besti /* each [ORC_VG][16] */)
{
    /* (24 named accumulators: as an array gcc keeps copying them between registers inside the loop) */
#define ORC_ROW(c) __m512i acc##c##0 = _mm512_setzero_si512(), acc##c##1 = acc##c##0, acc##c##2 = acc##c##0, acc##c##3 = acc##c##0;
    ORC_ROW(0) ORC_ROW(1) ORC_ROW(2) ORC_ROW(3) ORC_ROW(4) ORC_ROW(5)
#undef ORC_ROW
    const int8_t* const cp0 = cand[0]; const int8_t* const cp1 = cand[1]; const int8_t* const cp2 = cand[2];
    const int8_t* const cp3 = cand[3]; const int8_t* const cp4 = cand[4]; const int8_t* const cp5 = cand[5];
    for (int k = 0; k < kg; k++) {
        const __m512i a0 = _mm512_load_si512((const void*)(ap + ((size_t)k * ORC_VG + 0) * 64));
        const __m512i a1 = _mm512_load_si512((const void*)(ap + ((size_t)k * ORC_VG + 1) * 64));
        const __m512i a2 = _mm512_load_si512((const void*)(ap + ((size_t)k * ORC_VG + 2) * 64));
        const __m512i a3 = _mm512_load_si512((const void*)(ap + ((size_t)k * ORC_VG + 3) * 64));
#define ORC_STEP(c) { int32_t w; memcpy(&w, cp##c + 4 * k, 4); const __m512i b = _mm512_set1_epi32(w); \
        acc##c##0 = _mm512_dpbusd_epi32(acc##c##0, a0, b); acc##c##1 = _mm512_dpbusd_epi32(acc##c##1, a1, b); \
        acc##c##2 = _mm512_dpbusd_epi32(acc##c##2, a2, b); acc##c##3 = _mm512_dpbusd_epi32(acc##c##3, a3, b); }
        ORC_STEP(0) ORC_STEP(1) ORC_STEP(2) ORC_STEP(3) ORC_STEP(4) ORC_STEP(5)
#undef ORC_STEP
    }
    /* d2 = P + |c|^2 - 2 acc, lane by lane; does any lane of any accumulator reach its best d2?
     * (named values again: an array here makes gcc keep a copy of the accumulators in memory through the loop above) */
    const __m512i pg0 = _mm512_load_si512((const void*)(P + 0)), pg1 = _mm512_load_si512((const void*)(P + 16));
    const __m512i pg2 = _mm512_load_si512((const void*)(P + 32)), pg3 = _mm512_load_si512((const void*)(P + 48));
    const __m512i bb0 = _mm512_load_si512((const void*)(best2 + 0)), bb1 = _mm512_load_si512((const void*)(best2 + 16));
    const __m512i bb2 = _mm512_load_si512((const void*)(best2 + 32)), bb3 = _mm512_load_si512((const void*)(best2 + 48));
    __mmask16 any = 0;
#define ORC_D2(c, g) const __m512i d##c##g = _mm512_sub_epi32(_mm512_add_epi32(pg##g, n##c), _mm512_slli_epi32(acc##c##g, 1)); \
    any |= _mm512_cmple_epi32_mask(d##c##g, bb##g);
#define ORC_D2ROW(c) const __m512i n##c = _mm512_set1_epi32(nc[c]); ORC_D2(c, 0) ORC_D2(c, 1) ORC_D2(c, 2) ORC_D2(c, 3)
    ORC_D2ROW(0) ORC_D2ROW(1) ORC_D2ROW(2) ORC_D2ROW(3) ORC_D2ROW(4) ORC_D2ROW(5)
#undef ORC_D2ROW
#undef ORC_D2
    if (!any) return;
    __m512i d2[ORC_VC][ORC_VG];
#define ORC_ST(c) d2[c][0] = d##c##0; d2[c][1] = d##c##1; d2[c][2] = d##c##2; d2[c][3] = d##c##3;
    ORC_ST(0) ORC_ST(1) ORC_ST(2) ORC_ST(3) ORC_ST(4) ORC_ST(5)
#undef ORC_ST
    for (int c = 0; c < valid; c++) {          /* ascending candidate index */
        for (int g = 0; g < ORC_VG; g++) {
            const __m512i b2 = _mm512_load_si512((const void*)(best2 + 16 * g));
            const __mmask16 m = _mm512_cmple_epi32_mask(d2[c][g], b2);
            if (!m) continue;
            const __m512 bd = _mm512_load_ps(bestd + 16 * g);
            const __m512 d = _mm512_sqrt_ps(_mm512_cvtepi32_ps(d2[c][g]));      /* d2 < 2^24: exact conversion; IEEE root = sqrtf */
            const __mmask16 w = _mm512_mask_cmp_ps_mask(m, d, bd, _CMP_LT_OQ);
            if (!w) continue;
            _mm512_store_ps(bestd + 16 * g, _mm512_mask_mov_ps(bd, w, d));
            _mm512_store_si512((void*)(best2 + 16 * g), _mm512_mask_mov_epi32(b2, w, d2[c][g]));
            _mm512_store_si512((void*)(besti + 16 * g), _mm512_mask_mov_epi32(_mm512_load_si512((const void*)(besti + 16 * g)), w,
                                                                            _mm512_set1_epi32((int32_t)(j0 + c))));
        }
    }
}

ORC_VNNI_TARGET
static void knn1_rows_u8_vnni(const uint8_t* A, int64_t na, const uint8_t* B, int64_t nb, int dim,
                              int32_t* idx, float* dist, int threads)
{
    const int kg = (dim + 3) / 4, dpad = 4 * kg;
    const int rows_per_block = 16 * ORC_VG;
    const int64_t nblk = (na + rows_per_block - 1) / rows_per_block;
    /* the candidates once: bytes with the top bit flipped (c - 128 as int8), rows padded to dpad (+ 4 bytes of slack), norms */
    int8_t* bs = (int8_t*)aligned_alloc(64, (((size_t)(nb > 0 ? nb : 1) * (size_t)dpad + 4 + 63) / 64) * 64);
    int32_t* ncv = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nb > 0 ? nb : 1));
    {
        int64_t cap = (na * nb) / 4000000 + 1;
        if (cap < threads) threads = (int)cap;
        if (threads < 1) threads = 1;
    }
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t j = 0; j < nb; j++) {
        int32_t n2 = 0;
        for (int k = 0; k < dpad; k++) {
            const int v = k < dim ? B[j * dim + k] : 128;          /* (padding: c - 128 = 0, and a's padding is 0 as well) */
            bs[j * dpad + k] = (int8_t)(v - 128);
            n2 += k < dim ? v * v : 0;
        }
        ncv[j] = n2;
    }
#pragma omp parallel num_threads(threads)
    {
        uint8_t* ap = (uint8_t*)aligned_alloc(64, (size_t)kg * ORC_VG * 64);
        int32_t* P = (int32_t*)aligned_alloc(64, sizeof(int32_t) * 16 * ORC_VG);
        int32_t* best2 = (int32_t*)aligned_alloc(64, sizeof(int32_t) * 16 * ORC_VG);
        int32_t* besti = (int32_t*)aligned_alloc(64, sizeof(int32_t) * 16 * ORC_VG);
        float* bestd = (float*)aligned_alloc(64, sizeof(float) * 16 * ORC_VG);
#pragma omp for schedule(dynamic, 1)
        for (int64_t blk = 0; blk < nblk; blk++) {
            const int64_t i0 = blk * rows_per_block;
            for (int g = 0; g < ORC_VG; g++)
                for (int l = 0; l < 16; l++) {
                    const int64_t i = i0 + 16 * g + l;
                    int32_t n2 = 0, s1 = 0;
                    for (int k = 0; k < dpad; k++) {
                        const int v = (i < na && k < dim) ? A[i * dim + k] : 0;
                        ap[((size_t)(k >> 2) * ORC_VG + g) * 64 + 4 * l + (k & 3)] = (uint8_t)v;
                        n2 += v * v; s1 += v;
                    }
                    P[16 * g + l] = n2 - 256 * s1;
                    best2[16 * g + l] = INT32_MAX; besti[16 * g + l] = -1; bestd[16 * g + l] = FLT_MAX;
                }
            for (int64_t j = 0; j < nb; j += ORC_VC) {
                const int8_t* cand[ORC_VC];
                int32_t nc[ORC_VC];
                const int valid = (int)((nb - j) < ORC_VC ? (nb - j) : ORC_VC);
                for (int c = 0; c < ORC_VC; c++) {        /* (beyond the bank: row j again, never visited by the reduction) */
                    cand[c] = bs + (c < valid ? j + c : j) * dpad;
                    nc[c] = c < valid ? ncv[j + c] : (1 << 30);
                }
                vnni_block(ap, P, kg, cand, nc, valid, j, best2, bestd, besti);
            }
            for (int r = 0; r < rows_per_block && i0 + r < na; r++) {
                idx[i0 + r] = besti[r];
                dist[i0 + r] = besti[r] < 0 ? INFINITY : bestd[r];
            }
        }
        free(ap); free(P); free(best2); free(besti); free(bestd);
    }
    free(bs); free(ncv);
}
#endif

ORC_API int orc_have_vnni(void)
{
#if defined(__x86_64__)
    return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl") &&
           __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512vnni");
#else
    return 0;
#endif
}

/* orc_bf_xcheck1_u8 through the blocked VNNI scan; -2 where the host has no AVX-512 VNNI. */
ORC_API int orc_bf_xcheck1_u8_blocked(const uint8_t* Q, int64_t nq, const uint8_t* T, int64_t nt, int dim,
                                      int32_t* tidx_out, float* dist_out, int threads)
{
#if defined(__x86_64__)
    if (dim < 1) return -1;
    if (!orc_have_vnni()) return -2;
    int32_t* rq = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nt > 0 ? nt : 1));
    float* rd = (float*)malloc(sizeof(float) * (size_t)(nt > 0 ? nt : 1));
    knn1_rows_u8_vnni(T, nt, Q, nq, dim, rq, rd, nthreads(threads));
    for (int64_t q = 0; q < nq; q++) { dist_out[q] = FLT_MAX; tidx_out[q] = -1; }
    for (int64_t i = 0; i < nt; i++) {
        const int32_t q = rq[i];
        if (q < 0) continue;
        if (rd[i] < dist_out[q]) { dist_out[q] = rd[i]; tidx_out[q] = (int32_t)i; }
    }
    for (int64_t q = 0; q < nq; q++) if (tidx_out[q] < 0) dist_out[q] = INFINITY;
    free(rq); free(rd);
    return 0;
#else
    (void)Q; (void)nq; (void)T; (void)nt; (void)dim; (void)tidx_out; (void)dist_out; (void)threads;
    return -2;
#endif
}

ORC_API int orc_bf_xcheck1_f32(const float* Q, int64_t nq, const float* T, int64_t nt, int dim,
                               int order, int32_t* tidx, float* dist, int threads)
{
    if (dim < 1 || !ORC_ORDER_OK(order)) return -1;
    orc_mat q = mk(Q, dim, order), t = mk(T, dim, order);
    xcheck(&q, nq, &t, nt, tidx, dist, nthreads(threads));
    return 0;
}

ORC_API int orc_bf_xcheck1_u8(const uint8_t* Q, int64_t nq, const uint8_t* T, int64_t nt, int dim,
                              int32_t* tidx, float* dist, int threads)
{
    if (dim < 1) return -1;
    orc_mat q = mk(Q, dim, ORC_KIND_U8), t = mk(T, dim, ORC_KIND_U8);
    xcheck(&q, nq, &t, nt, tidx, dist, nthreads(threads));
    return 0;
}

/* ratio + threshold -- fastmatch.pyx:124,165 (ratio) and :50,75,82 (threshold).
 * ratio = float64(float32 distance) / float64 selfdist[queryIdx]; accepted = ratio < tau.
 * NumPy-scalar semantics: x/0 -> inf, 0/0 -> nan, both rejected by '<'.
 * qrows == NULL means queryIdx == position.                                          */
ORC_API int orc_ratio_filter(const float* dist, const double* selfdist, const int32_t* qrows,
                             int64_t n, double tau, double* ratio, uint8_t* pass, int64_t* n_pass)
{
    int64_t c = 0;
    for (int64_t i = 0; i < n; i++) {
        int64_t q = qrows ? qrows[i] : i;
        double r = (double)dist[i] / selfdist[q];
        if (ratio) ratio[i] = r;
        uint8_t p = (r < tau) ? 1 : 0;
        if (pass) pass[i] = p;
        c += p;
    }
    if (n_pass) *n_pass = c;
    return 0;
}

/* Classic Ratio-Match (Classic Matching.ipynb cell 3): m[0].distance / m[1].distance
 * in Python floats (float64).  The notebook would raise ZeroDivisionError when
 * d2 == 0; we report +inf (d1 > 0) or nan (d1 == 0) instead, both rejected.          */
ORC_API int orc_lowe_ratio(const float* dist2 /*[n*2]*/, int64_t n, double* ratio)
{
    for (int64_t i = 0; i < n; i++) ratio[i] = (double)dist2[2 * i] / (double)dist2[2 * i + 1];
    return 0;
}

ORC_API int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
