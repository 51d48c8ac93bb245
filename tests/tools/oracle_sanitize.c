/* ASAN / UBSAN harness for the CPU oracle's C restatement (oracle/bfmatch_oracle.c, compiled into this program): random
 * shapes incl. 0 and 1 rows, k = 2 against one row, dim < 128, the vectorised and the blocked (VNNI) cross-checks against the scalar one, all four
 * float32 accumulation orders.  Prints "ok" when no sanitizer report fired and the internal cross-checks held. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int orc_bf_knn_f32(const float*, int64_t, const float*, int64_t, int, int, int, int32_t*, float*, int);
int orc_bf_knn_u8(const uint8_t*, int64_t, const uint8_t*, int64_t, int, int, int32_t*, float*, int);
int orc_bf_xcheck1_u8_simd(const uint8_t*, int64_t, const uint8_t*, int64_t, int, int32_t*, float*, int);
int orc_bf_xcheck1_f32(const float*, int64_t, const float*, int64_t, int, int, int32_t*, float*, int);
int orc_bf_xcheck1_u8(const uint8_t*, int64_t, const uint8_t*, int64_t, int, int32_t*, float*, int);
int orc_have_simd(void);
int orc_bf_xcheck1_u8_blocked(const uint8_t*, int64_t, const uint8_t*, int64_t, int, int32_t*, float*, int);
int orc_have_vnni(void);

static uint64_t s = 88172645463325252ull;
static uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }

int main(void)
{
    static const int dims[] = {1, 3, 16, 64, 127, 128};
    long calls = 0;
    for (int it = 0; it < 600; ++it) {
        const int64_t nq = (int64_t)(rnd() % 70), nt = (int64_t)(rnd() % 70);
        const int dim = dims[rnd() % 6], k = 1 + (int)(rnd() % 2), thr = 1 + (int)(rnd() % 3);
        /* exact-size heap blocks: an access one element past the end is a report */
        uint8_t* Q = malloc((size_t)(nq * dim) + 1); uint8_t* T = malloc((size_t)(nt * dim) + 1);
        float* Qf = malloc(sizeof(float) * (size_t)(nq * dim) + 4); float* Tf = malloc(sizeof(float) * (size_t)(nt * dim) + 4);
        const int lowrange = (int)(rnd() % 3) == 0;
        for (int64_t i = 0; i < nq * dim; ++i) { Q[i] = (uint8_t)(lowrange ? rnd() % 3 : rnd() % 256); Qf[i] = (float)Q[i] + (float)(rnd() % 1000) / 2048.f; }
        for (int64_t i = 0; i < nt * dim; ++i) { T[i] = (uint8_t)(lowrange ? rnd() % 3 : rnd() % 256); Tf[i] = (float)T[i] + (float)(rnd() % 1000) / 2048.f; }
        int32_t* idx = malloc(sizeof(int32_t) * (size_t)(nq * k) + 4); float* dist = malloc(sizeof(float) * (size_t)(nq * k) + 4);
        int32_t* ti = malloc(sizeof(int32_t) * (size_t)nq + 4); float* td = malloc(sizeof(float) * (size_t)nq + 4);
        int32_t* ti2 = malloc(sizeof(int32_t) * (size_t)nq + 4); float* td2 = malloc(sizeof(float) * (size_t)nq + 4);
        if (orc_bf_knn_u8(Q, nq, T, nt, dim, k, idx, dist, thr) != 0) return 2;
        for (int64_t i = 0; i < nq * k; ++i)
            if (idx[i] < -1 || idx[i] >= nt || (idx[i] >= 0 && !(dist[i] >= 0.f))) { printf("knn_u8 out of range\n"); return 1; }
        for (int order = 0; order < 4; ++order) {
            if (orc_bf_knn_f32(Qf, nq, Tf, nt, dim, k, order, idx, dist, thr) != 0) return 2;
            if (orc_bf_xcheck1_f32(Qf, nq, Tf, nt, dim, order, ti, td, thr) != 0) return 2;
            calls += 2;
        }
        if (orc_bf_xcheck1_u8(Q, nq, T, nt, dim, ti, td, thr) != 0) return 2;
        if (orc_have_simd() && dim == 128) {
            if (orc_bf_xcheck1_u8_simd(Q, nq, T, nt, dim, ti2, td2, thr) != 0) return 2;
            if (memcmp(ti, ti2, sizeof(int32_t) * (size_t)nq) || memcmp(td, td2, sizeof(float) * (size_t)nq)) { printf("simd != scalar\n"); return 1; }
        }
        if (orc_have_vnni()) {          /* (every dim: rows are padded to 4-byte groups inside) */
            if (orc_bf_xcheck1_u8_blocked(Q, nq, T, nt, dim, ti2, td2, thr) != 0) return 2;
            if (memcmp(ti, ti2, sizeof(int32_t) * (size_t)nq) || memcmp(td, td2, sizeof(float) * (size_t)nq)) { printf("blocked != scalar\n"); return 1; }
        }
        /* cross-check property: a matched query is the reverse nearest neighbour of its train row (ties: an equal distance) */
        for (int64_t q = 0; q < nq; ++q)
            if (ti[q] >= 0) {
                if (ti[q] >= nt) { printf("xcheck index out of range\n"); return 1; }
                double d2 = 0.0;
                for (int c = 0; c < dim; ++c) { const double d = (double)Q[q * dim + c] - (double)T[(int64_t)ti[q] * dim + c]; d2 += d * d; }
                if (td[q] != sqrtf((float)d2)) { printf("xcheck distance differs from its definition\n"); return 1; }
            }
        calls += 3;
        free(Q); free(T); free(Qf); free(Tf); free(idx); free(dist); free(ti); free(td); free(ti2); free(td2);
    }
    printf("ok: %ld calls\n", calls);
    return 0;
}
