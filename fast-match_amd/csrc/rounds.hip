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
//   2. scatter-min of (float32 bits of sqrtf(d2) << 32 | t) into an LDS table indexed by query
//      slot (ds_min_u64; SURVEY.md Appendix A.3: lowest t wins ties -- ties of the float32
//      distance, which is what OpenCV compares),
//   3. decode: tidx, dist, ratio = double(dist) / selfdist[q_row]  (float64, fastmatch.pyx:165).
// Rounds are tiny (~400 x 125 descriptors), so the kernel is latency bound; what matters
// is that a round costs one launch and no host round trip between its three steps.
#include "round_body.h"
#include "round_body_f32.h"

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
    int            tie_guard;     // the banks' norms allow d2 >= kSqrtTieMin (tile_ops.h)
};

__global__ __launch_bounds__(256)
void round_kernel(RoundParams p)
{
    __shared__ __attribute__((aligned(16))) char smem[kStageBytes];
    __shared__ unsigned long long qbest[kRoundQCap];
    __shared__ unsigned long long tbest[kTbestWords];

    const int tid  = threadIdx.x;
    const int b    = blockIdx.x;
    const int64_t q0 = p.q_off[b];
    const int nq = (int)(p.q_off[b + 1] - q0);
    const int64_t t0 = p.t_off[b];
    const int nt = (int)(p.t_off[b + 1] - t0);

    for (int i = tid; i < nq; i += 256) qbest[i] = ~0ull;

    __syncthreads();
    if (nq > 0 && nt > 0)
        x1_round_wsplit<kStageRows>((gptr<const int8_t>)p.q_rows8, (gptr<const int32_t>)p.q_norm, p.q_rows + q0, nq,
                                    (gptr<const int8_t>)p.t_rows8, (gptr<const int32_t>)p.t_norm, t0, nt, smem, qbest, tbest, p.tie_guard);
    __syncthreads();
    for (int i = tid; i < nq; i += 256) {
        const unsigned long long key = qbest[i];
        int32_t ti = -1;
        float d = INFINITY;
        double r = NAN;
        if (key != ~0ull) {
            ti = (int32_t)(unsigned)key;
            d = x1_key_distance((unsigned)(key >> 32), p.tie_guard);
            if (p.q_selfdist) r = (double)d / p.q_selfdist[p.q_rows[q0 + i]];
        }
        p.tidx[q0 + i] = ti;
        p.dist[q0 + i] = d;
        if (p.ratio) p.ratio[q0 + i] = r;
    }
}

// The same launch for banks that are not integer valued (float32 route, round_body_f32.h).  A
// round whose candidate list overflows (pathological: thousands of query rows within the fp16
// margin of a train row's nearest) reports tidx = -2 for all its slots; the caller redoes it on
// the dense route.
constexpr int kRoundF32Clist = 8192;
constexpr int kRoundF32Lds = kRF_StageBytes + kRoundQCap * 8 + kRoundF32Clist * 4 + 128 * 8 + 64;

struct RoundF32Params {
    RoundF32       rf;
    const double*  q_selfdist;
    const int32_t* q_rows;
    const int64_t* q_off;
    const int64_t* t_off;
    int32_t*       tidx;
    float*         dist;
    double*        ratio;
};

__global__ __launch_bounds__(256)
void round_f32_kernel(RoundF32Params p)
{
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    char* smem = dyn;
    unsigned long long* qbest = (unsigned long long*)(dyn + kRF_StageBytes);
    unsigned* clist = (unsigned*)(qbest + kRoundQCap);
    unsigned long long* tbest = (unsigned long long*)(clist + kRoundF32Clist);
    int* sh = (int*)(tbest + 128);

    const int tid = threadIdx.x;
    const int b   = blockIdx.x;
    const int64_t q0 = p.q_off[b];
    const int nq = (int)(p.q_off[b + 1] - q0);
    const int64_t t0 = p.t_off[b];
    const int nt = (int)(p.t_off[b + 1] - t0);
    for (int i = tid; i < nq; i += 256) qbest[i] = ~0ull;
    __syncthreads();
    bool ok = true;
    if (nq > 0 && nt > 0) ok = x1_round_f32(RoundF32G(p.rf), p.q_rows + q0, nq, t0, nt, smem, qbest, clist, kRoundF32Clist, tbest, sh);
    __syncthreads();
    for (int i = tid; i < nq; i += 256) {
        const unsigned long long key = qbest[i];
        int32_t ti = ok ? -1 : -2;
        float d = INFINITY;
        double r = NAN;
        if (ok && key != ~0ull) {
            ti = (int32_t)(unsigned)key;
            d = __uint_as_float((unsigned)(key >> 32));
            if (p.q_selfdist) r = (double)d / p.q_selfdist[p.q_rows[q0 + i]];
        }
        p.tidx[q0 + i] = ti;
        p.dist[q0 + i] = d;
        if (p.ratio) p.ratio[q0 + i] = r;
    }
}

hipError_t launch_rounds_f32(const RoundF32& rf, const double* q_selfdist, const int32_t* d_q_rows, const int64_t* d_q_off,
                             const int64_t* d_t_off, int64_t n_rounds, int32_t* d_tidx, float* d_dist,
                             double* d_ratio, hipStream_t stream)
{
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)round_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kRoundF32Lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    RoundF32Params p;
    p.rf = rf; p.q_selfdist = q_selfdist;
    p.q_rows = d_q_rows; p.q_off = d_q_off; p.t_off = d_t_off;
    p.tidx = d_tidx; p.dist = d_dist; p.ratio = d_ratio;
    hipLaunchKernelGGL(round_f32_kernel, dim3((unsigned)n_rounds), dim3(256), kRoundF32Lds, stream, p);
    return hipGetLastError();
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
    p.tie_guard = sqrt_tie_possible(q, t) ? 1 : 0;
    hipLaunchKernelGGL(round_kernel, dim3((unsigned)n_rounds), dim3(256), 0, stream, p);
    return hipGetLastError();
}

int round_qcap() { return kRoundQCap; }

}  // namespace fm
