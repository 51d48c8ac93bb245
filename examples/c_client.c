/* A plain-C (C99) client of the drop-in boundary, include/fastmatch_hip.h -- what a binding in any language does:
 *   the reference's match_position round (fastmatch.pyx:161-165: cross-checked 1-NN of a query subset against a cell's
 *   descriptors, ratio against the query's self distances, threshold) through fm_self_dist + fm_match_ratio.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/c_client.c -Lfast-match_amd -lfastmatch_hip -Wl,-rpath,$PWD/fast-match_amd -lm -o c_client
 *   ./c_client [nq nt seed]
 *
 * Prints one line per accepted match ("q t distance ratio") after a header line "accepted <n> of <nq>".
 * Exit codes: 0 ok, 2 no usable device (the library has no CPU fallback), 3 a call failed, 4 the small built-in check
 * (the device's matches against a scalar loop in this file) disagreed.  tests/test_c_client*.py drive it. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "fastmatch_hip.h"

static uint32_t rng_state;
static uint32_t rng_next(void) { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

/* SIFT-like rows: non-negative bytes, a few large entries */
static void fill(uint8_t* rows, int64_t n)
{
    for (int64_t i = 0; i < n * 128; ++i) {
        const uint32_t r = rng_next();
        rows[i] = (uint8_t)((r & 7) == 0 ? 40 + (r >> 4) % 120 : (r >> 4) % 48);
    }
}

static int die(fm_ctx* ctx, const char* what, int rc)
{
    fprintf(stderr, "%s failed (%d): %s\n", what, rc, fm_last_error(ctx));
    return 3;
}

/* float32 distance the way cv::batchDistance stores it: exact integer d2, then sqrtf */
static float dist_u8(const uint8_t* a, const uint8_t* b)
{
    int32_t s = 0;
    for (int k = 0; k < 128; ++k) { const int32_t d = (int32_t)a[k] - (int32_t)b[k]; s += d * d; }
    return sqrtf((float)s);
}

int main(int argc, char** argv)
{
    const int64_t nq = argc > 1 ? atoll(argv[1]) : 300, nt = argc > 2 ? atoll(argv[2]) : 120;
    rng_state = argc > 3 ? (uint32_t)atoll(argv[3]) : 12345u;
    const double tau = 0.9;
    if (fm_abi_version() != FM_ABI_VERSION) { fprintf(stderr, "header / library ABI mismatch\n"); return 3; }

    uint8_t* Q = (uint8_t*)malloc((size_t)nq * 128);
    uint8_t* T = (uint8_t*)malloc((size_t)nt * 128);
    fill(Q, nq);
    fill(T, nt);
    for (int64_t i = 0; i < nt && i < nq; i += 3)                   /* plant near-copies so that something is accepted */
        for (int k = 0; k < 128; ++k) T[i * 128 + k] = (uint8_t)(Q[i * 128 + k] ^ ((k % 17) == 0));

    fm_ctx* ctx = NULL;
    int rc = fm_ctx_create(0, &ctx);
    if (rc != FM_OK) { fprintf(stderr, "no device: %s\n", fm_last_error(NULL)); return 2; }

    fm_bank *qb = NULL, *tb = NULL;
    if ((rc = fm_bank_create_u8(ctx, Q, nq, 128, &qb)) != FM_OK) return die(ctx, "fm_bank_create_u8(q)", rc);
    if ((rc = fm_bank_create_u8(ctx, T, nt, 128, &tb)) != FM_OK) return die(ctx, "fm_bank_create_u8(t)", rc);

    double* selfdist = (double*)malloc((size_t)nq * sizeof(double));
    if ((rc = fm_self_dist(ctx, qb, selfdist)) != FM_OK) return die(ctx, "fm_self_dist", rc);
    if ((rc = fm_bank_set_selfdist(ctx, qb, selfdist)) != FM_OK) return die(ctx, "fm_bank_set_selfdist", rc);

    int32_t* tidx = (int32_t*)malloc((size_t)nq * sizeof(int32_t));
    float* dist = (float*)malloc((size_t)nq * sizeof(float));
    double* ratio = (double*)malloc((size_t)nq * sizeof(double));
    uint8_t* pass = (uint8_t*)malloc((size_t)nq);
    int64_t n_pass = 0;
    if ((rc = fm_match_ratio(ctx, qb, tb, tau, tidx, dist, ratio, pass, &n_pass)) != FM_OK) return die(ctx, "fm_match_ratio", rc);

    /* the same by a scalar loop: reverse nearest neighbour per train row (lowest query index on ties), then per query
     * the closest train row that elected it (lowest train index on ties) -- cv::BFMatcher's cross-check; self distance
     * = nearest other row of the query bank */
    int bad = 0;
    int32_t* want = (int32_t*)malloc((size_t)nq * sizeof(int32_t));
    float* wdist = (float*)malloc((size_t)nq * sizeof(float));
    for (int64_t q = 0; q < nq; ++q) { want[q] = -1; wdist[q] = INFINITY; }
    for (int64_t t = 0; t < nt; ++t) {
        int64_t best = -1;
        float bd = INFINITY;
        for (int64_t q = 0; q < nq; ++q) {
            const float d = dist_u8(Q + q * 128, T + t * 128);
            if (d < bd) { bd = d; best = q; }
        }
        if (best >= 0 && bd < wdist[best]) { wdist[best] = bd; want[best] = (int32_t)t; }
    }
    int64_t want_pass = 0;
    for (int64_t q = 0; q < nq; ++q) {
        float sd = INFINITY;
        for (int64_t j = 0; j < nq; ++j)
            if (j != q) { const float d = dist_u8(Q + q * 128, Q + j * 128); if (d < sd) sd = d; }
        if ((double)sd != selfdist[q] || want[q] != tidx[q]) ++bad;
        if (want[q] >= 0) {
            const double r = (double)wdist[q] / (double)sd;
            if (wdist[q] != dist[q] || !(r == ratio[q] || (r != r && ratio[q] != ratio[q]))) ++bad;
            if (r < tau) ++want_pass;
            if ((r < tau) != (pass[q] != 0)) ++bad;
        }
    }
    if (want_pass != n_pass) ++bad;

    printf("accepted %lld of %lld\n", (long long)n_pass, (long long)nq);
    for (int64_t q = 0; q < nq; ++q)
        if (pass[q]) printf("%lld %d %.9g %.17g\n", (long long)q, tidx[q], (double)dist[q], ratio[q]);

    fm_stats st;
    if (fm_get_stats(ctx, &st) == FM_OK) fprintf(stderr, "device time %.3f ms\n", st.kernel_ms);
    fm_bank_destroy(ctx, qb);
    fm_bank_destroy(ctx, tb);
    fm_ctx_destroy(ctx);
    free(Q); free(T); free(selfdist); free(tidx); free(dist); free(ratio); free(pass); free(want); free(wdist);
    if (bad) { fprintf(stderr, "%d disagreements with the scalar loop\n", bad); return 4; }
    return 0;
}
