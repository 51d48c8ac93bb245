// K4 -- one workgroup per expansion round: the whole cross-checked match of
// fastmatch.pyx:161-167 (match_position) in a single launch, for many rounds at once.
//
// Round b:  query rows  q_rows[q_off[b] .. q_off[b+1])  gathered from the resident query
//           bank (the radius subset Metric_Cache.get returns, cache.pyx:173-188), against
//           the train rows [t_off[b], t_off[b+1]) of the resident target bank (one
//           Grid_Cache cell, cache.pyx:124-138).
// Per round the block
//   1. reverse-NN: for every train row t, argmin over the gathered query slots of d2
//      (int8 MFMA tiles, same in-lane reduction as rowreduce.hip; query rows are gathered
//      128 at a time into a swizzled LDS image, their accumulator-init words are built on
//      the fly from the bank's row norms),
//   2. scatter-min of (d2 << 32 | t) into an LDS table indexed by query slot
//      (ds_min_u64; SURVEY.md Appendix A.3: lowest t wins ties),
//   3. decode: tidx, dist = sqrtf(d2), ratio = double(dist) / selfdist[q_row]  (float64,
//      fastmatch.pyx:165).
// Rounds are tiny (~400 x 125 descriptors), so the kernel is latency bound; what matters
// is that a round costs one launch and no host round trip between its three steps.
#include "tile_ops.h"

namespace fm {

constexpr int kRoundQCap = 4096;     // query slots per round held in LDS (32 KiB)

struct RoundParams {
    const int8_t*  q_rows8;
    const int32_t* q_norm;
    const double*  q_selfdist;    // may be null
    const int32_t* q_rows;        // device [tot]
    const int64_t* q_off;         // device [B+1]
    const int8_t*  t_rows8;
    const int32_t* t_norm;
    const int64_t* t_off;         // device [B+1]
    int32_t*       tidx;          // device [tot]
    float*         dist;
    double*        ratio;         // may be null
};

template <int NB>
__global__ __launch_bounds__(256)
void round_kernel(RoundParams p)
{
    __shared__ __attribute__((aligned(16))) char smem[kStageBytes];
    __shared__ unsigned long long qbest[kRoundQCap];

    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int h    = lane >> 5;
    const int b    = blockIdx.x;
    const int64_t q0 = p.q_off[b];
    const int nq = (int)(p.q_off[b + 1] - q0);
    const int64_t t0 = p.t_off[b];
    const int nt = (int)(p.t_off[b + 1] - t0);

    for (int i = tid; i < nq; i += 256) qbest[i] = ~0ull;

    const int sw = ((lane & 31) >> 1) & 7;
    int aoff[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) aoff[c] = (lane & 31) * kDim + 16 * ((2 * c + h) ^ sw);
    const int xoff = kStageRowBytes + h * 64;
    const int nstages = (nq + kStageRows - 1) / kStageRows;

    for (int cb0 = 0; cb0 < nt; cb0 += 128 * NB) {
        const int cb = cb0 + wave * (32 * NB);
        v4i bf[NB][4];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int n = cb + 32 * j + (lane & 31);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (n < nt) bf[j][c] = *(const v4i*)(p.t_rows8 + (size_t)(t0 + n) * kDim + 32 * c + 16 * h);
                else        bf[j][c] = v4i{0, 0, 0, 0};
            }
        }
        TopK<1> top[NB];
        int thr[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) { thr[j] = INT32_MIN; top[j].init(); }

        for (int st = 0; st < nstages; ++st) {
            __syncthreads();                       // previous stage fully consumed
            // gather 128 query rows (16 B per thread x 4) into the swizzled image
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int g   = wave * 4 + i;
                const int row = g * 8 + (lane >> 3);
                const int slot = st * kStageRows + row;
                v4i v = v4i{0, 0, 0, 0};
                if (slot < nq) {
                    const int qi = p.q_rows[q0 + slot];
                    v = *(const v4i*)(p.q_rows8 + (size_t)qi * kDim + 16 * ((lane & 7) ^ ((row >> 1) & 7)));
                }
                *(v4i*)(smem + g * 1024 + lane * 16) = v;
            }
            if (tid < kStageRows) {
                const int slot = st * kStageRows + tid;
                const int tile = tid >> 5, mm = tid & 31;
                const int hh = (mm >> 2) & 1, reg = (mm & 3) + 4 * (mm >> 3);
                int cinit = kPadCinit, low = 15 - reg;
                if (slot < nq) {
                    const int nm = p.q_norm[p.q_rows[q0 + slot]];
                    cinit = -(nm >> 1);
                    low = ((1 - (nm & 1)) << 4) | (15 - reg);
                }
                int* aux = (int*)(smem + kStageRowBytes) + tile * kAuxPerTile;
                aux[16 * hh + reg] = cinit;
                aux[32 + 16 * hh + reg] = low;
            }
            __syncthreads();
            const int ntiles = min(kStageRows / kTileRows, (nq - st * kStageRows + kTileRows - 1) / kTileRows);
            for (int tt = 0; tt < ntiles; ++tt) {
                v4i af[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) af[c] = *(const v4i*)(smem + tt * (kTileRows * kDim) + aoff[c]);
                const v16i ci = lds_read16(smem + xoff + tt * (kAuxPerTile * 4));
                v16i acc[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[0], bf[j][0], ci, 0, 0, 0);
#pragma unroll
                for (int c = 1; c < 4; ++c)
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[c], bf[j][c], acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int tmax = max16(acc[j]);
                    if (__builtin_amdgcn_ballot_w64(tmax >= thr[j]) != 0ull) {
                        const v16i low = lds_read16(smem + xoff + tt * (kAuxPerTile * 4) + 128);
                        top[j].update(acc[j], low, st * (kStageRows / kTileRows) + tt);
                        thr[j] = top[j].own_threshold();
                    }
                }
            }
        }
        // cross-half merge, then scatter-min into the per-slot table
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int mh = top[j].hi(0);
            int mi = (top[j].tile[0] >= 0) ? top[j].index(0, h) : -1;
            if (mi >= nq) mi = -1;                  // padding slot
            const int oh = __shfl_xor(mh, 32);
            const int oi = __shfl_xor(mi, 32);
            const bool mine = !better(oh, oi, mh, mi);
            const int rh = mine ? mh : oh;
            const int ri = mine ? mi : oi;
            const int n = cb + 32 * j + (lane & 31);
            if (h == 0 && n < nt && ri >= 0) {
                const unsigned d2 = (unsigned)(p.t_norm[t0 + n] + 1 - rh);
                atomicMin(&qbest[ri], ((unsigned long long)d2 << 32) | (unsigned)n);
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < nq; i += 256) {
        const unsigned long long key = qbest[i];
        int32_t ti = -1;
        float d = INFINITY;
        double r = NAN;
        if (key != ~0ull) {
            ti = (int32_t)(unsigned)key;
            d = sqrtf((float)(unsigned)(key >> 32));
            if (p.q_selfdist) r = (double)d / p.q_selfdist[p.q_rows[q0 + i]];
        }
        p.tidx[q0 + i] = ti;
        p.dist[q0 + i] = d;
        if (p.ratio) p.ratio[q0 + i] = r;
    }
}

hipError_t launch_rounds(const Bank& q, const Bank& t, const int32_t* d_q_rows, const int64_t* d_q_off,
                         const int64_t* d_t_off, int64_t n_rounds, int32_t* d_tidx, float* d_dist,
                         double* d_ratio, hipStream_t stream)
{
    RoundParams p;
    p.q_rows8 = q.rows8; p.q_norm = q.norm; p.q_selfdist = q.selfdist;
    p.q_rows = d_q_rows; p.q_off = d_q_off;
    p.t_rows8 = t.rows8; p.t_norm = t.norm; p.t_off = d_t_off;
    p.tidx = d_tidx; p.dist = d_dist; p.ratio = d_ratio;
    hipLaunchKernelGGL((round_kernel<1>), dim3((unsigned)n_rounds), dim3(256), 0, stream, p);
    return hipGetLastError();
}

int round_qcap() { return kRoundQCap; }

}  // namespace fm
