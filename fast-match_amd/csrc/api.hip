// C-ABI of libfastmatch_hip.so (include/fastmatch_hip.h) and the small kernels around K1:
//   K6  bank upload: bytes XOR 0x80, row norms, accumulator-order aux words
//   K2  cross-check finaliser: reverse-NN partials -> packed (d2,idx) 64-bit scatter-min
//   K3  float64 ratio + threshold
//   top-2 merge of the split partials for knnMatch(k=2)
// Reference call sites are cited in the header next to each entry point.
#include "fm_internal.h"
#include "expand_pair.h"
#include "round_body_f32.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <vector>
#include <map>

using namespace fm;

// ---------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------
static std::mutex g_err_mu;
static std::string g_err;   // last context-less error

struct fm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_call0 = nullptr, ev_call1 = nullptr, ev_k0 = nullptr, ev_k1 = nullptr;
    std::string err;
    std::string devname;
    // growable device workspaces
    void*  ws_partial = nullptr; size_t ws_partial_bytes = 0;
    void*  ws_out = nullptr;     size_t ws_out_bytes = 0;
    void*  ws_in = nullptr;      size_t ws_in_bytes = 0;
    // configuration: fm_ctx_set_option (the FM_* environment variables seed it at creation)
    Tuning tune;
    int* d_counters = nullptr;   // device words of the fp16 filter (layout: fm_internal.h, launch_filter)
    int64_t filter_launches = 0;
    unsigned long long* h_scratch = nullptr;   // pinned host words the kernels can write (counts)
    // page-locked staging for results that go to pageable caller memory (d2h below)
    char*  h_stage = nullptr; size_t h_stage_bytes = 0, h_stage_used = 0;
    struct StagedCopy { void* dst; size_t off, bytes; };
    std::vector<StagedCopy> staged;
    // calls enqueued without a synchronisation (fm_match_accepted_async): their events, read at fm_sync
    struct PendingTimer { hipEvent_t c0, c1, k0, k1; bool timed; int64_t pairs; bool call_timed = true; };
    int64_t async_calls = 0;
    std::vector<PendingTimer> pending;       // in flight
    std::vector<PendingTimer> timer_pool;    // idle event sets
    // fm_match_accepted_async: K1 launches follow each other on `stream`; the small kernels behind a
    // K1 (election, decode + ratio, compaction) run on `stream_tail` and overlap the NEXT call's K1.
    // Two workspace slots alternate; a slot's tail kernels leave its bound[] and qbest[] arrays in
    // the state the next K1 / election expects, so no fill operations sit between two K1 launches.
    hipStream_t stream_tail = nullptr;      // = tails[0]
    static constexpr int kTails = 3;
    hipStream_t tails[kTails] = {nullptr, nullptr, nullptr};   // fm_match_accepted_batch spreads the pairs' tails over these
    hipStream_t rows_stream = nullptr;      // stream that produced the last device-resident rows (fm_gather_matches follows it)
    hipEvent_t ev_consumer = nullptr;
    hipEvent_t ev_tail_end[3] = {nullptr, nullptr, nullptr};   // one per tail stream (an event re-recorded on another stream
                                                               // before its waiters ran is not a safe handshake)
    struct AsyncSlot {
        void* ws = nullptr; size_t bytes = 0;
        int64_t nq = -1, ncols_alloc = -1, partial_bytes = -1;   // layout the arrays were initialised for
        hipEvent_t tail_done = nullptr, k_done = nullptr;
        bool in_use = false;
    } aslot[2];
    int aslot_next = 0;
    std::vector<AsyncSlot> bslot;           // fm_match_accepted_batch: a ring of kBatchSlots workspaces
    static constexpr int kBatchSlots = 32;  // (two launches of up to 16 pairs in flight; a slot is re-used behind its tail's event)
    int64_t bslot_next = 0;
    // fm_mark / fm_wait: points in the enqueued work a caller can wait for without draining what follows
    static constexpr int kMarks = 8;
    struct Mark { hipEvent_t ev[1 + kTails] = {nullptr, nullptr, nullptr, nullptr}; int64_t id = -1; } marks[kMarks];
    int64_t next_mark = 0;
    void* comm = nullptr;        // RCCL communicator of the result gather (fm_comm_init)
    int   comm_ranks = 0;
    fm_stats stats{};
    bool kernel_timed = false;
    int64_t pending_pairs = 0;
};

static int fail(fm_ctx* ctx, int code, const std::string& msg)
{
    if (ctx) ctx->err = msg;
    else { std::lock_guard<std::mutex> lk(g_err_mu); g_err = msg; }
    return code;
}

#define HIP_TRY(ctx, expr)                                                                  \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            char _b[512];                                                                   \
            snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                     __FILE__, __LINE__);                                                   \
            (void)hipGetLastError();                                                        \
            return fail(ctx, _e == hipErrorOutOfMemory ? FM_ENOMEM : FM_EDEVICE, _b);       \
        }                                                                                   \
    } while (0)

// Ablation builds only (-DFM_ABLATE, scripts/ablate): FM_ABLATE_KEEP_BOUNDS leaves the bounds of a finished
// run in place (measures what the exact path costs).  The product build always resets them.
static inline bool ablate_keep_bounds()
{
#ifdef FM_ABLATE
    return getenv("FM_ABLATE_KEEP_BOUNDS") != nullptr;
#else
    return false;
#endif
}

static int ws_ensure(fm_ctx* ctx, void** p, size_t* cap, size_t need)
{
    if (need <= *cap && *p) return FM_OK;
    if (*p) { HIP_TRY(ctx, hipFree(*p)); *p = nullptr; *cap = 0; }
    size_t sz = need + need / 4 + 4096;
    HIP_TRY(ctx, hipMalloc(p, sz));
    *cap = sz;
    return FM_OK;
}

// ---------------------------------------------------------------------------------------
// K6: bank preparation
// ---------------------------------------------------------------------------------------
// One 256-thread block per 32-row tile; thread (r = tid>>3, c = tid&7) owns the 16 bytes
// [16c, 16c+16) of tile row r.  SRC_F32: source rows are float32; values are converted to
// uint8 and nonint[0] is raised if any value is not an integer in [0,255]; nonint[1] = max over the
// rows of the squared norm of the uint8 row (Bank::usq_max).
template <bool SRC_F32>
__global__ __launch_bounds__(256)
void bank_prep_kernel(const void* __restrict__ src, int64_t n, int dim,
                      int8_t* __restrict__ rows8, int32_t* __restrict__ norm,
                      int32_t* __restrict__ aux, int* __restrict__ nonint)
{
    const int tid = threadIdx.x;
    const int r = tid >> 3, c = tid & 7;
    const int64_t tile = blockIdx.x;
    const int64_t row = tile * kTileRows + r;
    unsigned w[4] = {0, 0, 0, 0};
    int sumsq = 0, usq = 0;
    bool bad = false;
    if (row < n) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned word = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int k = 16 * c + 4 * q + b;
                int u = 128;                         // padding beyond dim: 0 after the shift
                if (k < dim) {
                    if constexpr (SRC_F32) {
                        const float f = ((const float*)src)[row * dim + k];
                        const float fr = rintf(f);
                        if (!(f == fr) || f < 0.f || f > 255.f) { bad = true; u = 128; }
                        else u = (int)fr;
                    } else {
                        u = ((const uint8_t*)src)[row * dim + k];
                    }
                    usq += u * u;
                }
                const int s = u - 128;               // == (int8)(u ^ 0x80)
                sumsq += s * s;
                word |= (unsigned)(s & 0xff) << (8 * b);
            }
            w[q] = word;
        }
    }
    *(uint4*)(rows8 + row * kDim + 16 * c) = make_uint4(w[0], w[1], w[2], w[3]);
    sumsq += __shfl_xor(sumsq, 1);
    sumsq += __shfl_xor(sumsq, 2);
    sumsq += __shfl_xor(sumsq, 4);
    usq += __shfl_xor(usq, 1);
    usq += __shfl_xor(usq, 2);
    usq += __shfl_xor(usq, 4);
    usq = max(usq, __shfl_xor(usq, 8));
    usq = max(usq, __shfl_xor(usq, 16));
    usq = max(usq, __shfl_xor(usq, 32));
    if ((tid & 63) == 0 && usq > 0) atomicMax(nonint + 1, usq);
    if (c == 0) {
        // aux words of the 32-row unit in the accumulator order of v_mfma_i32_16x16x64_i8
        // (two 16-row tiles; tile row rr sits in lane group rr >> 2, register rr & 3)
        const int sub = r >> 4, rr = r & 15;
        const int id = 4 * sub + (rr & 3);
        int32_t* a = aux + tile * kAuxPerTile + 32 * sub;
        if (row < n) {
            norm[row] = sumsq;
            a[rr]      = -(sumsq >> 1);
            a[16 + rr] = ((1 - (sumsq & 1)) << 4) | (15 - id);
        } else {
            norm[row] = 0;
            a[rr]      = kPadCinit;
            a[16 + rr] = 15 - id;
        }
    }
    if constexpr (SRC_F32) {
        // one atomic per wave at most, and none once the flag is up (a bank that is not integer
        // valued would otherwise send one atomic per element to the same address)
        if (__builtin_amdgcn_ballot_w64(bad) != 0ull && (tid & 63) == 0 && *(volatile int*)nonint == 0) atomicOr(nonint, 1);
    }
}

// float32 bank for the general (non-integer) route: zero-padded copy [n_pad][128].
__global__ void bank_copy_f32_kernel(const float* __restrict__ src, int64_t n, int dim,
                                     float* __restrict__ dst, int64_t n_pad)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad * kDim) return;
    const int64_t row = i / kDim;
    const int k = (int)(i % kDim);
    dst[i] = (row < n && k < dim) ? src[row * dim + k] : 0.f;
}

// fp16 rows, norms and accumulator inits of a float32 bank for the fp16 filter (filter_f16.hip).
// Pass 1: stat[0] = max |value| (float bits), stat[1] |= 1 if a value is not finite.
__global__ __launch_bounds__(256)
void bank_absmax_kernel(const float* __restrict__ rowsf, int64_t total, int* __restrict__ stat)
{
    float m = 0.f;
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const float v = fabsf(rowsf[i]);
        bad |= !(v <= 3.0e38f);
        m = fmaxf(m, v);
    }
#pragma unroll
    for (int mask = 1; mask < 64; mask <<= 1) m = fmaxf(m, __shfl_xor(m, mask));
    if (__builtin_amdgcn_ballot_w64(bad) != 0ull) { if ((threadIdx.x & 63) == 0) atomicOr(stat + 1, 1); }
    else if ((threadIdx.x & 63) == 0) atomicMax(stat, (int)__float_as_uint(m));     // m >= 0: bit order = value order
}

// Pass 2: 16 lanes per row, 8 dims per lane; rows scaled by 2^k (exact) and rounded to fp16
// (nearest even); norms of the scaled rows in float64 -> float32.  stat[0] = max norm.
__global__ __launch_bounds__(256)
void bank_prep_f16_kernel(const float* __restrict__ rowsf, int64_t n, int64_t n_pad, int k,
                          uint16_t* __restrict__ rowsh, float* __restrict__ normf,
                          float* __restrict__ auxf, int* __restrict__ stat)
{
    const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int c = threadIdx.x & 15;
    if (row >= n_pad) return;
    const float4 v0 = *(const float4*)(rowsf + row * kDim + 8 * c);
    const float4 v1 = *(const float4*)(rowsf + row * kDim + 8 * c + 4);
    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    unsigned h[8];
    double ss = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float x = ldexpf(v[i], k);
        const _Float16 hx = (_Float16)x;
        h[i] = (unsigned)__builtin_bit_cast(unsigned short, hx);
        ss += (double)x * (double)x;
    }
    *(uint4*)(rowsh + row * kDim + 8 * c) =
        make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
    ss += __shfl_xor(ss, 1);
    ss += __shfl_xor(ss, 2);
    ss += __shfl_xor(ss, 4);
    ss += __shfl_xor(ss, 8);
    if (c == 0) {
        const float nm = (float)ss;
        if (row < n) {
            normf[row] = nm;
            auxf[row] = -0.5f * nm;
        } else {
            normf[row] = 0.f;
            auxf[row] = -3.4e38f;
        }
    }
    // max norm: one atomic per wave (4 rows), not per row
    float m = (c == 0 && row < n) ? (float)ss : 0.f;
#pragma unroll
    for (int mask = 16; mask < 64; mask <<= 1) m = fmaxf(m, __shfl_xor(m, mask));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(stat, (int)__float_as_uint(m));
}

// ---------------------------------------------------------------------------------------
// merge / finalise kernels
// ---------------------------------------------------------------------------------------
// knnMatch(k=2): merge nsplit partial top-2 lists per query row (keys are (d2<<32)|idx,
// ascending = cv::batchDistance order) and emit idx / sqrtf(d2).
// Partial keys carry the exact integer d2 (int8 route) or the float32 bits of the distance
// itself (float32 route) in their high word.  The d2 order is OpenCV's (float32 distance, index)
// order as long as the second best d2 stays below kSqrtTieMin (tile_ops.h); output rows beyond that
// are listed in fix[] (fix[0] = count, rows from fix[4] on) and redone by sqrt_fix_kernel<2>.
__device__ __forceinline__ float key_dist(unsigned long long key, int f32)
{
    const unsigned hi = (unsigned)(key >> 32);
    return f32 ? __uint_as_float(hi) : sqrtf((float)hi);
}

__global__ void knn2_merge_kernel(const unsigned long long* __restrict__ partial, int nsplit,
                                  int ncols_alloc, int64_t n, int32_t* __restrict__ idx,
                                  float* __restrict__ dist, int f32, unsigned* __restrict__ fix = nullptr)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long b0 = ~0ull, b1 = ~0ull;
    for (int s = 0; s < nsplit; ++s) {
        const unsigned long long* p = partial + ((size_t)s * ncols_alloc + i) * 2;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const unsigned long long v = p[k];
            if (v < b0) { b1 = b0; b0 = v; }
            else if (v < b1) { b1 = v; }
        }
    }
    idx[2 * i]     = (b0 == ~0ull) ? -1 : (int32_t)(unsigned)b0;
    idx[2 * i + 1] = (b1 == ~0ull) ? -1 : (int32_t)(unsigned)b1;
    dist[2 * i]     = (b0 == ~0ull) ? INFINITY : key_dist(b0, f32);
    dist[2 * i + 1] = (b1 == ~0ull) ? INFINITY : key_dist(b1, f32);
    if (fix && b1 != ~0ull && (unsigned)(b1 >> 32) >= kSqrtTieMin) fix[4 + atomicAdd(fix, 1u)] = (unsigned)i;
}

// Exact repair of the output rows listed in fix[] (see knn2_merge_kernel / xcheck_scatter_kernel): one
// workgroup per listed row scans ALL reduced rows with exact integer arithmetic and orders them the way
// cv::batchDistance does, by (float32 bits of sqrtf(d2), index).  KTOP = 2: rewrites the row's 2-NN
// list; KTOP = 1: the row is a train row, the winner is the query row it elects -> scatter-min into
// qbest (the step xcheck_scatter_kernel left out for this row).  Cold: rows whose K-th best d2 is
// >= kSqrtTieMin = 4 197 200 (SIFT descriptors: d2 <= 1.05e6).
template <int KTOP>
__global__ __launch_bounds__(256)
void sqrt_fix_kernel(const unsigned* __restrict__ fix, const int8_t* __restrict__ col_rows,
                     const int32_t* __restrict__ col_norm, const int8_t* __restrict__ red_rows,
                     const int32_t* __restrict__ red_norm, int nred,
                     unsigned long long* __restrict__ qbest, unsigned t_offset,
                     int32_t* __restrict__ idx, float* __restrict__ dist)
{
    __shared__ unsigned long long best[2];
    const int tid = threadIdx.x;
    const unsigned n = fix[0];
    for (unsigned e = blockIdx.x; e < n; e += gridDim.x) {
        const unsigned c = fix[4 + e];
        if (tid == 0) { best[0] = ~0ull; best[1] = ~0ull; }
        __syncthreads();
        v4i cr[kDim / 16];
#pragma unroll
        for (int w = 0; w < kDim / 16; ++w) cr[w] = *(const v4i*)(col_rows + (size_t)c * kDim + 16 * w);
        const int cn = col_norm[c];
        unsigned long long k0 = ~0ull, k1 = ~0ull;
        for (int m = tid; m < nred; m += 256) {
            int dot = 0;
#pragma unroll
            for (int w = 0; w < kDim / 16; ++w) {
                const v4i y = *(const v4i*)(red_rows + (size_t)m * kDim + 16 * w);
#pragma unroll
                for (int u = 0; u < 4; ++u) dot = __builtin_amdgcn_sdot4(cr[w][u], y[u], dot, false);
            }
            const unsigned d2 = (unsigned)(cn + red_norm[m] - 2 * dot);
            const unsigned long long key = ((unsigned long long)sqrt_bits(d2) << 32) | (unsigned)m;
            if (key < k0) { k1 = k0; k0 = key; }
            else if (key < k1) k1 = key;
        }
        if (k0 != ~0ull) atomicMin(&best[0], k0);
        __syncthreads();
        const unsigned long long g0 = best[0];
        if constexpr (KTOP == 2) {
            const unsigned long long mine = (k0 == g0) ? k1 : k0;
            if (mine != ~0ull) atomicMin(&best[1], mine);
            __syncthreads();
            if (tid == 0) {
                const unsigned long long g1 = best[1];
                idx[2 * (size_t)c]     = (g0 == ~0ull) ? -1 : (int32_t)(unsigned)g0;
                idx[2 * (size_t)c + 1] = (g1 == ~0ull) ? -1 : (int32_t)(unsigned)g1;
                dist[2 * (size_t)c]     = (g0 == ~0ull) ? INFINITY : __uint_as_float((unsigned)(g0 >> 32));
                dist[2 * (size_t)c + 1] = (g1 == ~0ull) ? INFINITY : __uint_as_float((unsigned)(g1 >> 32));
            }
        } else {
            if (tid == 0 && g0 != ~0ull)
                atomicMin(&qbest[(unsigned)g0], (g0 & 0xffffffff00000000ull) | (unsigned long long)(c + t_offset));
        }
        __syncthreads();
    }
}

// Cross-check step 2: train row t elects rq = argmin_q d(q,t)
// (lowest q on ties) = min over the split partials; then scatter-min of (d2<<32 | t)
// into qbest[rq]: q keeps the closest electing train row, lowest t on ties.  64-bit
// atomicMin is order independent, so the result is deterministic.  The key's high word is the float32
// distance (bits of sqrtf(d2); f32: the partial keys carry those bits already): OpenCV compares the
// distances, and above kSqrtTieMin two d2 can share one.  fix != null: a train row whose best d2 shares
// its root with d2 + 1 may elect a LOWER query index that sits at d2 + 1 -- it is listed in fix[] and
// left to sqrt_fix_kernel<1>.
// bound_reset (async calls): the K1 that produced `partial` is complete, so its bound[] array is
// put back to "no bound" here for the next K1 that uses this workspace slot.
__global__ void xcheck_scatter_kernel(const unsigned long long* __restrict__ partial, int nsplit,
                                      int ncols_alloc, int64_t nt,
                                      unsigned long long* __restrict__ qbest, unsigned t_offset, int f32,
                                      int* __restrict__ bound_reset, unsigned* __restrict__ fix)
{
    // four lanes per train row, each takes every 4th split: short independent load chains
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (bound_reset && gid < ncols_alloc) bound_reset[gid] = INT32_MIN;
    const int64_t t = gid >> 2;
    const int part = (int)(gid & 3);
    unsigned long long b = ~0ull;
    if (t < nt) {
        for (int s = part; s < nsplit; s += 4) {
            const unsigned long long v = partial[(size_t)s * ncols_alloc + t];
            b = v < b ? v : b;
        }
    }
    unsigned long long o = __shfl_xor(b, 1);
    b = o < b ? o : b;
    o = __shfl_xor(b, 2);
    b = o < b ? o : b;
    if (t >= nt || part != 0 || b == ~0ull) return;
    const unsigned q = (unsigned)b;
    unsigned hi = (unsigned)(b >> 32);
    if (!f32) {
        if (fix && hi >= kSqrtTieMin && sqrt_ties_up(hi)) { fix[4 + atomicAdd(fix, 1u)] = (unsigned)t; return; }
        hi = sqrt_bits(hi);
    }
    // (t_offset: global index of this bank's first row when the train set is sharded over ranks)
    const unsigned long long key = ((unsigned long long)hi << 32) | (unsigned long long)((unsigned)t + t_offset);
    atomicMin(&qbest[q], key);
}

// Cross-check step 3 + optional R1: decode qbest (high word = float32 distance bits), ratio = (double)dist / selfdist[q] in float64, pass = ratio < tau
// (fastmatch.pyx:124,165; :50,75,82).
__global__ void xcheck_finalize_kernel(const unsigned long long* qbest, int64_t nq,
                                       const double* __restrict__ selfdist, double tau,
                                       int32_t* __restrict__ tidx, float* __restrict__ dist,
                                       double* __restrict__ ratio, uint8_t* __restrict__ pass,
                                       unsigned long long* __restrict__ npass,
                                       int* __restrict__ block_counts,
                                       unsigned long long* qbest_reset = nullptr)
{
    __shared__ int wave_cnt[4];
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool p = false;
    if (q < nq) {
        const unsigned long long key = qbest[q];
        if (qbest_reset) qbest_reset[q] = ~0ull;     // (async calls: the slot's table is left empty for its next election)
        int32_t ti = -1;
        float d = INFINITY;
        double r = NAN;
        if (key != ~0ull) {
            ti = (int32_t)(unsigned)key;
            d = __uint_as_float((unsigned)(key >> 32));
            if (selfdist) { r = (double)d / selfdist[q]; p = r < tau; }
        }
        tidx[q] = ti;
        dist[q] = d;
        if (ratio) ratio[q] = r;
        if (pass) pass[q] = p ? 1 : 0;
    }
    const unsigned long long m = __ballot(p);
    if (npass && !block_counts) {            // (with block counts the total comes from compact_kernel)
        if ((threadIdx.x & 63) == 0 && m) atomicAdd(npass, (unsigned long long)__popcll(m));
    }
    if (block_counts) {                      // for the ordered compaction (compact_kernel)
        if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = __popcll(m);
        __syncthreads();
        if (threadIdx.x == 0) block_counts[blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    }
}

// Ordered stream compaction of the accepted matches (ascending query index): block b sums
// the counts of the blocks before it, then every accepted row writes itself at
// offset + rank.  Deterministic (no atomics).
__global__ void compact_kernel(const int32_t* __restrict__ tidx, const float* __restrict__ dist,
                               const double* __restrict__ ratio, const uint8_t* __restrict__ pass,
                               const int* __restrict__ block_counts, int64_t nq, int64_t cap,
                               int32_t* __restrict__ o_q, int32_t* __restrict__ o_t,
                               float* __restrict__ o_d, double* __restrict__ o_r,
                               unsigned long long* __restrict__ npass)
{
    __shared__ int red[256];
    __shared__ int wave_base[4];
    const int tid = threadIdx.x;
    int s = 0;
    for (int b = tid; b < (int)blockIdx.x; b += 256) s += block_counts[b];
    red[tid] = s;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (tid < d) red[tid] += red[tid + d];
        __syncthreads();
    }
    const int64_t base = red[0];
    const int64_t q = (int64_t)blockIdx.x * 256 + tid;
    const bool p = q < nq && pass[q];
    const unsigned long long m = __ballot(p);
    const int lane = tid & 63, wave = tid >> 6;
    if (lane == 0) wave_base[wave] = __popcll(m);
    __syncthreads();
    int wb = 0;
    for (int w = 0; w < wave; ++w) wb += wave_base[w];
    if (blockIdx.x == gridDim.x - 1 && tid == 0)       // total = everything before the last block + its own
        *npass = (unsigned long long)(base + wave_base[0] + wave_base[1] + wave_base[2] + wave_base[3]);
    if (p) {
        const int64_t dst = base + wb + __popcll(m & ((1ull << lane) - 1ull));
        if (dst < cap) { o_q[dst] = (int32_t)q; o_t[dst] = tidx[q]; o_d[dst] = dist[q]; o_r[dst] = ratio[q]; }
    }
}

// Same ordered compaction, but into the 12-byte rows the multi-GPU result gather ships
// (query index, train index, float32 distance bits) in a caller-supplied DEVICE buffer, with
// the count as a device word next to it (fm_match_accepted_dev): nothing crosses to the host.
__global__ void compact_rows_kernel(const int32_t* __restrict__ tidx, const float* __restrict__ dist,
                                    const uint8_t* __restrict__ pass, const int* __restrict__ block_counts,
                                    int64_t nq, int64_t cap, int32_t* __restrict__ o_rows,
                                    long long* __restrict__ o_count, unsigned long long* __restrict__ h_count)
{
    __shared__ int red[256];
    __shared__ int wave_base[4];
    const int tid = threadIdx.x;
    int s = 0;
    for (int b = tid; b < (int)blockIdx.x; b += 256) s += block_counts[b];
    red[tid] = s;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (tid < d) red[tid] += red[tid + d];
        __syncthreads();
    }
    const int64_t base = red[0];
    const int64_t q = (int64_t)blockIdx.x * 256 + tid;
    const bool p = q < nq && pass[q];
    const unsigned long long m = __ballot(p);
    const int lane = tid & 63, wave = tid >> 6;
    if (lane == 0) wave_base[wave] = __popcll(m);
    __syncthreads();
    int wb = 0;
    for (int w = 0; w < wave; ++w) wb += wave_base[w];
    if (blockIdx.x == gridDim.x - 1 && tid == 0) {
        // the device word counts the rows that are THERE (what a consumer of o_rows may index: the gather ships
        // it next to the rows); the host word keeps the full number of accepted matches
        const long long tot = (long long)(base + wave_base[0] + wave_base[1] + wave_base[2] + wave_base[3]);
        *o_count = tot < (long long)cap ? tot : (long long)cap;
        if (h_count) *h_count = (unsigned long long)tot;
    }
    if (p) {
        const int64_t dst = base + wb + __popcll(m & ((1ull << lane) - 1ull));
        if (dst < cap) {
            o_rows[3 * dst] = (int32_t)q;
            o_rows[3 * dst + 1] = tidx[q];
            o_rows[3 * dst + 2] = (int32_t)__float_as_uint(dist[q]);
        }
    }
}

__global__ void ratio_filter_kernel(const float* __restrict__ dist, const double* __restrict__ selfdist,
                                    const int32_t* __restrict__ qrows, int64_t n, double tau,
                                    double* __restrict__ ratio, uint8_t* __restrict__ pass,
                                    unsigned long long* __restrict__ npass)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool p = false;
    if (i < n) {
        const int64_t q = qrows ? qrows[i] : i;
        const double r = (double)dist[i] / selfdist[q];
        p = r < tau;
        if (ratio) ratio[i] = r;
        if (pass) pass[i] = p ? 1 : 0;
    }
    const unsigned long long m = __ballot(p);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(npass, (unsigned long long)__popcll(m));
}

// Classic Ratio-Match (Classic Matching.ipynb cell 3): ratio = float64(d1) / float64(d2) of
// the 2-NN list, accepted when ratio < tau (d2 == 0 gives inf / nan: rejected, where the
// notebook's Python division would raise).  Feeds compact_kernel.
__global__ void lowe_kernel(const int32_t* __restrict__ idx2, const float* __restrict__ dist2, int64_t nq,
                            double tau, int32_t* __restrict__ tidx, float* __restrict__ dist,
                            double* __restrict__ ratio, uint8_t* __restrict__ pass,
                            int* __restrict__ block_counts)
{
    __shared__ int wave_cnt[4];
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool p = false;
    if (q < nq) {
        const float d1 = dist2[2 * q], d2 = dist2[2 * q + 1];
        const double r = (idx2[2 * q + 1] >= 0) ? (double)d1 / (double)d2 : NAN;
        p = r < tau;
        tidx[q] = idx2[2 * q];
        dist[q] = d1;
        ratio[q] = r;
        pass[q] = p ? 1 : 0;
    }
    const unsigned long long m = __ballot(p);
    if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

__global__ void selfdist_from_knn_kernel(const float* __restrict__ dist2, int64_t n, double* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (double)dist2[2 * i + 1];
}

// ---------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------
extern "C" const char* fm_last_error(const fm_ctx* ctx)
{
    if (ctx) return ctx->err.c_str();
    std::lock_guard<std::mutex> lk(g_err_mu);
    static thread_local std::string copy;
    copy = g_err;
    return copy.c_str();
}

// Options of a context by name (include/fastmatch_hip.h lists them).
struct OptionDef { const char* name; int Tuning::* field; int lo, hi; const char* env; };
static const OptionDef kOptions[] = {
    {"nb", &Tuning::nb, 0, 8, "FM_NB"}, {"nsplit", &Tuning::nsplit, 0, 1 << 20, "FM_NSPLIT"}, {"nw", &Tuning::nw, 0, 16, "FM_NW"},
    {"nbuf", &Tuning::nbuf, 0, 3, "FM_NBUF"}, {"prio", &Tuning::prio, 0, 1, "FM_PRIO"},
    {"glds", &Tuning::glds, 0, 1, "FM_GLDS"}, {"coop", &Tuning::coop, 0, 1, "FM_COOP"},
    {"f32_filter", &Tuning::f32_filter, 0, 2, "FM_F32_FILTER"}, {"f32_nw", &Tuning::f32_nw, 0, 8, "FM_F32_NW"},
    {"f32_nsplit", &Tuning::f32_nsplit, 0, 1 << 20, "FM_F32_NSPLIT"}, {"f32_fused", &Tuning::f32_fused, -1, 1, "FM_F32_FUSED"},
    {"f32_lpc", &Tuning::f32_lpc, 0, 64, "FM_F32_LPC"},
    {"batch_group", &Tuning::batch_group, 1, kRRBatchMax, "FM_BATCH_GROUP"}, {"batch_tail", &Tuning::batch_tail, 0, kRRBatchMax, "FM_BATCH_TAIL"},
    {"async_time_every", &Tuning::async_time_every, 0, 1 << 20, "FM_ASYNC_TIME_EVERY"},
    {"expand_big", &Tuning::expand_big, 0, 1, nullptr}, {"expand_grow", &Tuning::expand_grow, 0, 4, nullptr}, {"expand_prof", &Tuning::expand_prof, 0, 1, "FM_EXPAND_PROF"},
};

extern "C" int fm_ctx_set_option(fm_ctx* ctx, const char* name, int64_t value)
{
    if (!ctx || !name) return fail(ctx, FM_EINVAL, "fm_ctx_set_option: NULL argument");
    for (const OptionDef& o : kOptions) {
        if (strcmp(o.name, name) != 0) continue;
        if (value < o.lo || value > o.hi) return fail(ctx, FM_EINVAL, std::string("fm_ctx_set_option: value out of range for ") + name);
        if (o.field == &Tuning::f32_filter && value != 0 && !ctx->d_counters)
            return fail(ctx, FM_EDEVICE, "fm_ctx_set_option: the fp16 filter's counters could not be allocated on this context");
        ctx->tune.*(o.field) = (int)value;
        return FM_OK;
    }
    return fail(ctx, FM_EINVAL, std::string("fm_ctx_set_option: unknown option ") + name);
}

extern "C" int fm_ctx_get_option(fm_ctx* ctx, const char* name, int64_t* value)
{
    if (!ctx || !name || !value) return fail(ctx, FM_EINVAL, "fm_ctx_get_option: NULL argument");
    for (const OptionDef& o : kOptions)
        if (strcmp(o.name, name) == 0) { *value = ctx->tune.*(o.field); return FM_OK; }
    return fail(ctx, FM_EINVAL, std::string("fm_ctx_get_option: unknown option ") + name);
}

extern "C" int fm_ctx_destroy(fm_ctx* ctx);

extern "C" int fm_ctx_create(int device_id, fm_ctx** out)
{
    if (!out) return fail(nullptr, FM_EINVAL, "fm_ctx_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return fail(nullptr, FM_EDEVICE,
                    std::string("fm_ctx_create: no HIP device available (") +
                    (e != hipSuccess ? hipGetErrorString(e) : "device count 0") +
                    "); libfastmatch_hip has no CPU fallback");
    }
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, FM_EINVAL, "fm_ctx_create: bad device id");
    fm_ctx* ctx = new (std::nothrow) fm_ctx();
    if (!ctx) return fail(nullptr, FM_ENOMEM, "fm_ctx_create: out of host memory");
    ctx->device = device_id;
    hipDeviceProp_t prop;
    if ((e = hipSetDevice(device_id)) != hipSuccess || (e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess) {
        delete ctx;
        return fail(nullptr, FM_EDEVICE, std::string("fm_ctx_create: ") + hipGetErrorString(e));
    }
    ctx->devname = std::string(prop.gcnArchName) + " " + prop.name;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        std::string m = "fm_ctx_create: device is " + ctx->devname + "; this library is built for gfx950 only";
        delete ctx;
        return fail(nullptr, FM_EUNSUPPORTED, m);
    }
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&ctx->ev_call0)) != hipSuccess || (e = hipEventCreate(&ctx->ev_call1)) != hipSuccess ||
        (e = hipEventCreate(&ctx->ev_k0)) != hipSuccess || (e = hipEventCreate(&ctx->ev_k1)) != hipSuccess) {
        delete ctx;
        return fail(nullptr, FM_EDEVICE, std::string("fm_ctx_create: ") + hipGetErrorString(e));
    }
    {
        // the tail stream gets the highest priority: its workgroups are few and short and should not
        // queue behind the thousands of workgroups of the K1 they overlap
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); least = greatest = 0; }
        if ((e = hipStreamCreateWithPriority(&ctx->tails[0], hipStreamNonBlocking, greatest)) != hipSuccess ||
            (e = hipStreamCreateWithPriority(&ctx->tails[1], hipStreamNonBlocking, greatest)) != hipSuccess ||
            (e = hipStreamCreateWithPriority(&ctx->tails[2], hipStreamNonBlocking, greatest)) != hipSuccess ||
            ((ctx->stream_tail = ctx->tails[0]), false) ||
            (e = hipEventCreateWithFlags(&ctx->aslot[0].tail_done, hipEventDisableTiming)) != hipSuccess ||
            (e = hipEventCreateWithFlags(&ctx->aslot[1].tail_done, hipEventDisableTiming)) != hipSuccess ||
            (e = hipEventCreateWithFlags(&ctx->ev_consumer, hipEventDisableTiming)) != hipSuccess ||
            (e = hipEventCreateWithFlags(&ctx->ev_tail_end[0], hipEventDisableTiming)) != hipSuccess ||
            (e = hipEventCreateWithFlags(&ctx->ev_tail_end[1], hipEventDisableTiming)) != hipSuccess ||
            (e = hipEventCreateWithFlags(&ctx->ev_tail_end[2], hipEventDisableTiming)) != hipSuccess ||
            (e = hipEventCreateWithFlags(&ctx->aslot[0].k_done, hipEventDisableTiming)) != hipSuccess ||
            (e = hipEventCreateWithFlags(&ctx->aslot[1].k_done, hipEventDisableTiming)) != hipSuccess) {
            fm_ctx_destroy(ctx);
            return fail(nullptr, FM_EDEVICE, std::string("fm_ctx_create: ") + hipGetErrorString(e));
        }
    }
    if (hipHostMalloc((void**)&ctx->h_scratch, 64, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); ctx->h_scratch = nullptr; }
    // defaults from the environment (experiments; a value out of range is ignored), then per context
    // through fm_ctx_set_option
    for (const OptionDef& o : kOptions)
        if (o.env) if (const char* s = getenv(o.env)) { const long v = atol(s); if (v >= o.lo && v <= o.hi) ctx->tune.*(o.field) = (int)v; }
    if (getenv("FM_EXPAND_NO_BIG")) ctx->tune.expand_big = 0;
    if (hipMalloc((void**)&ctx->d_counters, filter_flag_bytes()) != hipSuccess || hipMemset(ctx->d_counters, 0, filter_flag_bytes()) != hipSuccess) {
        (void)hipGetLastError();
        ctx->d_counters = nullptr;
        ctx->tune.f32_filter = 0;
    }
    *out = ctx;
    return FM_OK;
}

extern "C" int fm_ctx_destroy(fm_ctx* ctx)
{
    if (!ctx) return FM_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (hipStream_t ts : ctx->tails) if (ts) (void)hipStreamSynchronize(ts);
    auto free_slot = [](fm_ctx::AsyncSlot& sl) {
        if (sl.ws) (void)hipFree(sl.ws);
        if (sl.tail_done) (void)hipEventDestroy(sl.tail_done);
        if (sl.k_done) (void)hipEventDestroy(sl.k_done);
    };
    for (auto& m : ctx->marks) for (hipEvent_t ev : m.ev) if (ev) (void)hipEventDestroy(ev);
    for (auto& sl : ctx->aslot) free_slot(sl);
    for (auto& sl : ctx->bslot) free_slot(sl);
    if (ctx->ev_consumer) (void)hipEventDestroy(ctx->ev_consumer);
    for (hipEvent_t ev : ctx->ev_tail_end) if (ev) (void)hipEventDestroy(ev);
    if (ctx->comm) { comm_destroy(ctx->comm); ctx->comm = nullptr; }      // (before the streams it was used on)
    for (hipStream_t ts : ctx->tails) if (ts) (void)hipStreamDestroy(ts);
    for (auto* v : {&ctx->pending, &ctx->timer_pool})
        for (auto& t : *v) { (void)hipEventDestroy(t.c0); (void)hipEventDestroy(t.c1); (void)hipEventDestroy(t.k0); (void)hipEventDestroy(t.k1); }
    if (ctx->ws_partial) (void)hipFree(ctx->ws_partial);
    if (ctx->ws_out) (void)hipFree(ctx->ws_out);
    if (ctx->ws_in) (void)hipFree(ctx->ws_in);
    if (ctx->h_scratch) (void)hipHostFree(ctx->h_scratch);
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    if (ctx->d_counters) (void)hipFree(ctx->d_counters);
    if (ctx->ev_call0) (void)hipEventDestroy(ctx->ev_call0);
    if (ctx->ev_call1) (void)hipEventDestroy(ctx->ev_call1);
    if (ctx->ev_k0) (void)hipEventDestroy(ctx->ev_k0);
    if (ctx->ev_k1) (void)hipEventDestroy(ctx->ev_k1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return FM_OK;
}

// Account the calls that were enqueued without a synchronisation; the stream must be idle.
static int drain_pending(fm_ctx* ctx)
{
    for (auto& t : ctx->pending) {
        float ms = 0.f;
        if (t.call_timed) {      // enqueue-to-results latency of the call (overlapped calls: not additive)
            if (hipEventElapsedTime(&ms, t.k0, t.c1) == hipSuccess) { ctx->stats.total_ms += ms; ctx->stats.calls += 1; }
            else (void)hipGetLastError();
        }
        if (t.timed) {
            if (hipEventElapsedTime(&ms, t.k0, t.k1) == hipSuccess) {
                ctx->stats.kernel_ms += ms;
                ctx->stats.kernel_launches += 1;
                ctx->stats.pairs += t.pairs;
            } else (void)hipGetLastError();
        }
        ctx->timer_pool.push_back(t);
    }
    ctx->pending.clear();
    return FM_OK;
}

extern "C" int fm_sync(fm_ctx* ctx)
{
    if (!ctx) return fail(nullptr, FM_EINVAL, "fm_sync: ctx is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (hipStream_t ts : ctx->tails) HIP_TRY(ctx, hipStreamSynchronize(ts));
    return drain_pending(ctx);
}

// fm_mark: remember "everything enqueued on this context so far"; fm_wait: block until that point is
// reached.  Work enqueued after the mark keeps running: a consumer can read the results of batch i while
// batch i + 1 is already on the device (double-buffered outputs) -- fm_sync would drain both.
extern "C" int fm_mark(fm_ctx* ctx, int64_t* ticket)
{
    if (!ctx || !ticket) return fail(ctx, FM_EINVAL, "fm_mark: NULL argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    fm_ctx::Mark& m = ctx->marks[ctx->next_mark % fm_ctx::kMarks];
    for (int u = 0; u < 1 + fm_ctx::kTails; ++u) {
        if (!m.ev[u]) HIP_TRY(ctx, hipEventCreateWithFlags(&m.ev[u], hipEventDisableTiming));
        HIP_TRY(ctx, hipEventRecord(m.ev[u], u == 0 ? ctx->stream : ctx->tails[u - 1]));
    }
    m.id = ctx->next_mark;
    *ticket = ctx->next_mark++;
    return FM_OK;
}

extern "C" int fm_wait(fm_ctx* ctx, int64_t ticket)
{
    if (!ctx) return fail(nullptr, FM_EINVAL, "fm_wait: ctx is NULL");
    if (ticket < 0 || ticket >= ctx->next_mark) return fail(ctx, FM_EINVAL, "fm_wait: unknown ticket");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    fm_ctx::Mark& m = ctx->marks[ticket % fm_ctx::kMarks];
    if (m.id != ticket) {            // the slot has been re-used by a later mark: everything older is covered by it
        if (m.id < ticket) return fail(ctx, FM_EINVAL, "fm_wait: unknown ticket");
    }
    for (hipEvent_t ev : m.ev) if (ev) HIP_TRY(ctx, hipEventSynchronize(ev));
    return FM_OK;
}

extern "C" int fm_get_stats(fm_ctx* ctx, fm_stats* out)
{
    if (!ctx || !out) return fail(ctx, FM_EINVAL, "fm_get_stats: NULL argument");
    if (!ctx->pending.empty()) {
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (hipStream_t ts : ctx->tails) HIP_TRY(ctx, hipStreamSynchronize(ts));
        drain_pending(ctx);
    }
    *out = ctx->stats;
    return FM_OK;
}

extern "C" int fm_reset_stats(fm_ctx* ctx)
{
    if (!ctx) return fail(nullptr, FM_EINVAL, "fm_reset_stats: ctx is NULL");
    ctx->stats = fm_stats{};
    return FM_OK;
}

extern "C" int fm_f32_filter_stats(fm_ctx* ctx, int64_t* launches, int64_t* fallbacks)
{
    if (!ctx) return fail(nullptr, FM_EINVAL, "fm_f32_filter_stats: ctx is NULL");
    int c[4] = {0, 0, 0, 0};
    if (ctx->d_counters) {
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipMemcpy(c, ctx->d_counters, 16, hipMemcpyDeviceToHost));
    }
    if (getenv("FM_F32_DEBUG")) fprintf(stderr, "[fm] filter: launches %lld, redone by K5 %d, output rows rescanned %d\n",
                                        (long long)ctx->filter_launches, c[2], c[3]);
    if (launches) *launches = ctx->filter_launches;
    if (fallbacks) *fallbacks = c[2];
    return FM_OK;
}

extern "C" int fm_device_name(fm_ctx* ctx, char* buf, int buflen)
{
    if (!ctx || !buf || buflen <= 0) return fail(ctx, FM_EINVAL, "fm_device_name: bad argument");
    snprintf(buf, (size_t)buflen, "%s", ctx->devname.c_str());
    return FM_OK;
}

// Page-locked allocations made through fm_host_alloc, with their device-side aliases: the async entry
// points look up to fifty output pointers per call, and a runtime query per pointer (microseconds each)
// would sit in front of the first launch of a batch.
struct PinnedRange { size_t bytes; char* dev; };
static std::mutex g_pinned_mu;
static std::map<uintptr_t, PinnedRange> g_pinned;

extern "C" int fm_host_alloc(fm_ctx* ctx, int64_t bytes, void** out)
{
    if (!ctx || !out || bytes < 0) return fail(ctx, FM_EINVAL, "fm_host_alloc: bad argument");
    *out = nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t sz = (size_t)(bytes > 0 ? bytes : 1);
    HIP_TRY(ctx, hipHostMalloc(out, sz, hipHostMallocDefault));
    void* dev = nullptr;
    if (hipHostGetDevicePointer(&dev, *out, 0) == hipSuccess && dev) {
        std::lock_guard<std::mutex> lk(g_pinned_mu);
        g_pinned[(uintptr_t)*out] = PinnedRange{sz, (char*)dev};
    } else (void)hipGetLastError();
    return FM_OK;
}

extern "C" int fm_host_free(fm_ctx* ctx, void* p)
{
    if (!p) return FM_OK;
    if (ctx) (void)hipSetDevice(ctx->device);
    { std::lock_guard<std::mutex> lk(g_pinned_mu); g_pinned.erase((uintptr_t)p); }
    hipError_t e = hipHostFree(p);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(ctx, FM_EDEVICE, std::string("fm_host_free: ") + hipGetErrorString(e)); }
    return FM_OK;
}

// Device-side alias of a page-locked host buffer (fm_host_alloc / hipHostMalloc), or NULL for
// ordinary pageable memory: kernels can then write results straight into the caller's buffer.
static void* pinned_device_alias(const void* host)
{
    if (!host) return nullptr;
    {
        std::lock_guard<std::mutex> lk(g_pinned_mu);
        auto it = g_pinned.upper_bound((uintptr_t)host);
        if (it != g_pinned.begin()) {
            --it;
            const size_t off = (uintptr_t)host - it->first;
            if (off < it->second.bytes) return it->second.dev + off;
        }
    }
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, host) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (at.type != hipMemoryTypeHost) return nullptr;
    return at.devicePointer;
}

// Copy through the kernel's own stores (dst is the device alias of page-locked host memory).
__global__ void copy_out_kernel(unsigned char* __restrict__ dst, const unsigned char* __restrict__ src, size_t bytes)
{
    const size_t words = bytes / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += stride)
        ((unsigned*)dst)[i] = ((const unsigned*)src)[i];
    if (blockIdx.x == 0 && threadIdx.x < (bytes & 3)) dst[words * 4 + threadIdx.x] = src[words * 4 + threadIdx.x];
}

// Device -> caller memory on the context's stream.  A copy into pageable memory makes the
// runtime pin the destination pages for the transfer (milliseconds for results of ~100 KB and
// up), so such results land in the context's own page-locked staging buffer and are moved to
// the caller by CallScope::finish() after the call's single synchronisation.
static hipError_t copy_out(fm_ctx* ctx, unsigned char* dst_alias, const void* src, size_t bytes)
{
    const unsigned grid = (unsigned)((bytes / 4 + 255) / 256 < 1024 ? (bytes / 4 + 255) / 256 + 1 : 1024);
    hipLaunchKernelGGL(copy_out_kernel, dim3(grid), dim3(256), 0, ctx->stream, dst_alias, (const unsigned char*)src, bytes);
    return hipGetLastError();
}

static hipError_t d2h(fm_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    if (bytes == 0) return hipSuccess;
    // A copy kernel rather than hipMemcpyAsync for anything but tiny results: the runtime hands
    // device-to-host copies of 64 KiB and more to a DMA queue behind a host-side wait for the
    // stream, which was seen to add 1-7 ms of idle time after multi-millisecond kernels.
    const bool kernel_ok = bytes >= 4096 && bytes <= ((size_t)256 << 20) && ((uintptr_t)src & 3) == 0;
    if (kernel_ok) {
        if (unsigned char* direct = (unsigned char*)pinned_device_alias(dst))       // page-locked destination
            return copy_out(ctx, direct, src, bytes);
        size_t off = (ctx->h_stage_used + 63) & ~(size_t)63;
        if (off + bytes > ctx->h_stage_bytes && ctx->staged.empty()) {
            if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
            ctx->h_stage = nullptr;
            ctx->h_stage_bytes = 0;
            const size_t want = bytes * 3 + (1 << 20);
            if (hipHostMalloc((void**)&ctx->h_stage, want, hipHostMallocDefault) == hipSuccess) ctx->h_stage_bytes = want;
            else { (void)hipGetLastError(); ctx->h_stage = nullptr; }
            off = 0;
        }
        if (ctx->h_stage && off + bytes <= ctx->h_stage_bytes) {
            if (unsigned char* alias = (unsigned char*)pinned_device_alias(ctx->h_stage + off)) {
                hipError_t e = copy_out(ctx, alias, src, bytes);
                if (e != hipSuccess) return e;
                ctx->staged.push_back({dst, off, bytes});
                ctx->h_stage_used = off + bytes;
                return hipSuccess;
            }
        }
    }
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream);
}

// Brackets one API call: events for total time, stats accounting after the final sync.
struct CallScope {
    fm_ctx* ctx;
    ~CallScope() { ctx->staged.clear(); ctx->h_stage_used = 0; }
    explicit CallScope(fm_ctx* c) : ctx(c)
    {
        // entries left behind by a call that failed half way point at host memory that is gone
        ctx->staged.clear();
        ctx->h_stage_used = 0;
        ctx->kernel_timed = false;
        ctx->pending_pairs = 0;
        (void)hipEventRecord(ctx->ev_call0, ctx->stream);
    }
    int finish()
    {
        HIP_TRY(ctx, hipEventRecord(ctx->ev_call1, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        // (async calls still in flight finish on the tail stream: a synchronous call completes them too,
        // as the header promises; they are accounted at the next fm_sync / fm_get_stats)
        if (!ctx->pending.empty()) for (hipStream_t ts : ctx->tails) HIP_TRY(ctx, hipStreamSynchronize(ts));
        for (const auto& c : ctx->staged) memcpy(c.dst, ctx->h_stage + c.off, c.bytes);
        ctx->staged.clear();
        ctx->h_stage_used = 0;
        float ms = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_call0, ctx->ev_call1));
        ctx->stats.total_ms += ms;
        ctx->stats.calls += 1;
        if (ctx->kernel_timed) {
            float kms = 0.f;
            HIP_TRY(ctx, hipEventElapsedTime(&kms, ctx->ev_k0, ctx->ev_k1));
            ctx->stats.kernel_ms += kms;
            ctx->stats.kernel_launches += 1;
            ctx->stats.pairs += ctx->pending_pairs;
        }
        return FM_OK;
    }
};

// ---------------------------------------------------------------------------------------
// banks
// ---------------------------------------------------------------------------------------
static void bank_free(Bank* b)
{
    if (b->rows8) (void)hipFree(b->rows8);
    if (b->norm) (void)hipFree(b->norm);
    if (b->aux) (void)hipFree(b->aux);
    if (b->rowsf) (void)hipFree(b->rowsf);
    if (b->rowsh) (void)hipFree(b->rowsh);
    if (b->normf) (void)hipFree(b->normf);
    if (b->auxf) (void)hipFree(b->auxf);
    if (b->selfdist) (void)hipFree(b->selfdist);
    b->rows8 = nullptr; b->norm = nullptr; b->aux = nullptr; b->rowsf = nullptr; b->selfdist = nullptr;
    b->rowsh = nullptr; b->normf = nullptr; b->auxf = nullptr;
}

static int bank_create(fm_ctx* ctx, const void* rows, int64_t n, int dim, bool f32, fm_bank** out, bool keep_f32 = false)
{
    if (!ctx) return fail(nullptr, FM_EINVAL, "fm_bank_create: ctx is NULL");
    if (!out) return fail(ctx, FM_EINVAL, "fm_bank_create: bank out pointer is NULL");
    *out = nullptr;
    if (n < 0 || dim < 1 || (n > 0 && !rows)) return fail(ctx, FM_EINVAL, "fm_bank_create: bad rows/n/dim");
    if (dim > kDim) return fail(ctx, FM_EUNSUPPORTED, "fm_bank_create: dim > 128 is not supported");
    if (n > (int64_t)INT32_MAX - 2 * kStageRows) return fail(ctx, FM_EUNSUPPORTED, "fm_bank_create: n too large");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    fm_bank* b = new (std::nothrow) fm_bank();
    if (!b) return fail(ctx, FM_ENOMEM, "fm_bank_create: out of host memory");
    b->n = n;
    b->dim = dim;
    b->n_pad = ((n + kStageRows - 1) / kStageRows) * kStageRows;
    if (b->n_pad == 0) b->n_pad = kStageRows;
    b->kind = FM_BANK_I8;
    const size_t elt = f32 ? 4 : 1;
    const size_t src_bytes = (size_t)n * dim * elt;
    int rc = FM_OK;
    auto bail = [&](int code) { bank_free(b); delete b; return code; };

    const size_t flag_off = (src_bytes + 15) & ~(size_t)15;
    if ((rc = ws_ensure(ctx, &ctx->ws_in, &ctx->ws_in_bytes, flag_off + 32)) != FM_OK) return bail(rc);
    int* d_flag = (int*)((char*)ctx->ws_in + flag_off);
#define BTRY(expr)                                                                               \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            (void)hipGetLastError();                                                             \
            return bail(fail(ctx, _e == hipErrorOutOfMemory ? FM_ENOMEM : FM_EDEVICE,            \
                             std::string(#expr " failed: ") + hipGetErrorString(_e)));           \
        }                                                                                        \
    } while (0)
    BTRY(hipMalloc((void**)&b->rows8, (size_t)b->n_pad * kDim));
    BTRY(hipMalloc((void**)&b->norm, (size_t)b->n_pad * 4));
    BTRY(hipMalloc((void**)&b->aux, (size_t)(b->n_pad / kTileRows) * kAuxPerTile * 4));
    if (src_bytes) BTRY(hipMemcpyAsync(ctx->ws_in, rows, src_bytes, hipMemcpyHostToDevice, ctx->stream));
    BTRY(hipMemsetAsync(d_flag, 0, 8, ctx->stream));
    const int ntiles = (int)(b->n_pad / kTileRows);
    if (f32)
        hipLaunchKernelGGL(bank_prep_kernel<true>, dim3(ntiles), dim3(256), 0, ctx->stream,
                           (const void*)ctx->ws_in, n, dim, b->rows8, b->norm, b->aux, d_flag);
    else
        hipLaunchKernelGGL(bank_prep_kernel<false>, dim3(ntiles), dim3(256), 0, ctx->stream,
                           (const void*)ctx->ws_in, n, dim, b->rows8, b->norm, b->aux, d_flag);
    BTRY(hipGetLastError());
    int flags[2] = {0, 0};
    BTRY(hipMemcpyAsync(flags, d_flag, 8, hipMemcpyDeviceToHost, ctx->stream));
    BTRY(hipStreamSynchronize(ctx->stream));
    const int flag = flags[0];
    b->usq_max = flags[1];
    if (f32 && (flag || (keep_f32 && n > 0))) {
        // not integer-valued (or the caller wants the float32 route): keep a float32 bank for the fma-chain route
        b->kind = FM_BANK_F32;
        (void)hipFree(b->rows8); b->rows8 = nullptr;
        (void)hipFree(b->aux); b->aux = nullptr;
        BTRY(hipMalloc((void**)&b->rowsf, (size_t)b->n_pad * kDim * 4));
        const int64_t tot = b->n_pad * kDim;
        hipLaunchKernelGGL(bank_copy_f32_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const float*)ctx->ws_in, n, dim, b->rowsf, b->n_pad);
        BTRY(hipGetLastError());
        // rows for the fp16 filter, scaled by the power of two that puts the largest magnitude
        // of the bank in [2^13, 2^14)
        BTRY(hipMalloc((void**)&b->rowsh, (size_t)b->n_pad * kDim * 2));
        BTRY(hipMalloc((void**)&b->normf, (size_t)b->n_pad * 4));
        BTRY(hipMalloc((void**)&b->auxf, (size_t)b->n_pad * 4));
        BTRY(hipMemsetAsync(d_flag, 0, 8, ctx->stream));
        hipLaunchKernelGGL(bank_absmax_kernel, dim3(1024), dim3(256), 0, ctx->stream, (const float*)b->rowsf, tot, d_flag);
        BTRY(hipGetLastError());
        int stat[2] = {0, 0};
        BTRY(hipMemcpyAsync(stat, d_flag, 8, hipMemcpyDeviceToHost, ctx->stream));
        BTRY(hipStreamSynchronize(ctx->stream));
        float vmax = 0.f;
        memcpy(&vmax, &stat[0], 4);
        b->filt_ok = stat[1] == 0;
        if (b->filt_ok) {
            int ex = 0;
            if (vmax > 0.f) (void)frexpf(vmax, &ex);         // vmax = m 2^ex, m in [0.5, 1)
            b->kscale = vmax > 0.f ? 14 - ex : 0;
            BTRY(hipMemsetAsync(d_flag, 0, 8, ctx->stream));
            hipLaunchKernelGGL(bank_prep_f16_kernel, dim3((unsigned)(b->n_pad / 16)), dim3(256), 0, ctx->stream,
                               (const float*)b->rowsf, n, b->n_pad, b->kscale, b->rowsh, b->normf, b->auxf, d_flag);
            BTRY(hipGetLastError());
            BTRY(hipMemcpyAsync(stat, d_flag, 8, hipMemcpyDeviceToHost, ctx->stream));
            BTRY(hipStreamSynchronize(ctx->stream));
            memcpy(&b->nm_max, &stat[0], 4);
        }
    }
#undef BTRY
    *out = b;
    return FM_OK;
}

extern "C" int fm_bank_create_u8(fm_ctx* ctx, const uint8_t* rows, int64_t n, int dim, fm_bank** bank)
{
    return bank_create(ctx, rows, n, dim, false, bank);
}

extern "C" int fm_bank_create_f32(fm_ctx* ctx, const float* rows, int64_t n, int dim, fm_bank** bank)
{
    return bank_create(ctx, rows, n, dim, true, bank);
}

extern "C" int fm_bank_create_f32_route(fm_ctx* ctx, const float* rows, int64_t n, int dim, fm_bank** bank)
{
    return bank_create(ctx, rows, n, dim, true, bank, true);
}

// Everything enqueued on the context -- its own stream and the tail streams the async entry points use.
static void sync_all_streams(fm_ctx* ctx)
{
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (hipStream_t ts : ctx->tails) if (ts) (void)hipStreamSynchronize(ts);
}

extern "C" int fm_bank_destroy(fm_ctx* ctx, fm_bank* bank)
{
    if (!bank) return FM_OK;
    if (ctx) sync_all_streams(ctx);          // (tail kernels of async calls read the bank's self distances)
    bank_free(bank);
    delete bank;
    return FM_OK;
}

extern "C" int fm_bank_info(const fm_bank* bank, int64_t* n, int* dim, int* kind)
{
    if (!bank) return fail(nullptr, FM_EINVAL, "fm_bank_info: bank is NULL");
    if (n) *n = bank->n;
    if (dim) *dim = bank->dim;
    if (kind) *kind = bank->kind;
    return FM_OK;
}

extern "C" int fm_bank_set_selfdist(fm_ctx* ctx, fm_bank* bank, const double* selfdist)
{
    if (!ctx || !bank) return fail(ctx, FM_EINVAL, "fm_bank_set_selfdist: NULL argument");
    if (bank->n > 0 && !selfdist) return fail(ctx, FM_EINVAL, "fm_bank_set_selfdist: selfdist is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!bank->selfdist) HIP_TRY(ctx, hipMalloc((void**)&bank->selfdist, (size_t)(bank->n > 0 ? bank->n : 1) * 8));
    if (bank->n > 0) {
        HIP_TRY(ctx, hipMemcpyAsync(bank->selfdist, selfdist, (size_t)bank->n * 8, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return FM_OK;
}

static int check_pair(fm_ctx* ctx, const fm_bank* q, const fm_bank* t, const char* who)
{
    if (!ctx) return fail(nullptr, FM_EINVAL, std::string(who) + ": ctx is NULL");
    if (!q || !t) return fail(ctx, FM_EINVAL, std::string(who) + ": bank is NULL");
    if (q->dim != t->dim) return fail(ctx, FM_EINVAL, std::string(who) + ": query/train dim mismatch");
    if (q->kind != t->kind && q->n > 0 && t->n > 0)      // (an empty bank has no kind of its own)
        return fail(ctx, FM_EINVAL, std::string(who) + ": query/train kind mismatch (one bank is integer-valued, the other is not)");
    return FM_OK;
}

// ---------------------------------------------------------------------------------------
// float32 route: K5 alone, or the fp16 filter (K8) with K5 as its conditional fallback
// ---------------------------------------------------------------------------------------
// Leaves the packed keys in ws_partial in `pl`'s layout (K5's plan) either way.
static int rowreduce_f32_route(fm_ctx* ctx, const fm_bank* cols, const fm_bank* red, int ktop, RowReducePlan* pl_out)
{
    const RowReducePlan pl = plan_rowreduce_f32(cols->n_pad, red->n_pad, ctx->tune.nsplit);
    *pl_out = pl;
    const bool filter = ctx->tune.f32_filter != 0 && filter_usable(*cols, *red) &&
                        (ctx->tune.f32_filter >= 2 || (double)cols->n * (double)red->n >= 4.0e6);
    const size_t part = (pl.partial_bytes(ktop) + 255) & ~(size_t)255;
    if (!filter) {
        int rc = ws_ensure(ctx, &ctx->ws_partial, &ctx->ws_partial_bytes, part);
        if (rc != FM_OK) return rc;
        HIP_TRY(ctx, hipEventRecord(ctx->ev_k0, ctx->stream));
        HIP_TRY(ctx, launch_rowreduce_f32(*cols, *red, ktop, pl, (unsigned long long*)ctx->ws_partial, nullptr, ctx->stream));
        HIP_TRY(ctx, hipEventRecord(ctx->ev_k1, ctx->stream));
        return FM_OK;
    }
    const FilterPlan fp = plan_filter(cols->n_pad, red->n_pad, ctx->tune);
    int rc = ws_ensure(ctx, &ctx->ws_partial, &ctx->ws_partial_bytes, part + fp.slots_bytes() + fp.bound_bytes() + 64);
    if (rc != FM_OK) return rc;
    unsigned long long* d_part = (unsigned long long*)ctx->ws_partial;
    unsigned long long* d_slots = (unsigned long long*)((char*)ctx->ws_partial + part);
    int* d_bound = (int*)((char*)d_slots + fp.slots_bytes());
    HIP_TRY(ctx, hipMemsetAsync(d_part, 0xff, pl.partial_bytes(ktop), ctx->stream));
    HIP_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)d_bound, filter_empty_bound(), (size_t)fp.ncols_alloc * 2, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_counters, 0, 8, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_k0, ctx->stream));
    HIP_TRY(ctx, launch_filter(*cols, *red, ktop, fp, d_slots, d_bound, ctx->d_counters, d_part, ctx->stream));
    HIP_TRY(ctx, launch_rowreduce_f32(*cols, *red, ktop, pl, d_part, ctx->d_counters, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_k1, ctx->stream));
    ctx->filter_launches += 1;
    return FM_OK;
}

// ---------------------------------------------------------------------------------------
// float32-root ties of the integer route (tile_ops.h: kSqrtTieMin)
// ---------------------------------------------------------------------------------------
constexpr int kFixGrid = 1024;                 // workgroups of a sqrt_fix_kernel launch (each walks the list)
static inline size_t fix_bytes(int64_t rows) { return ((size_t)rows * 4 + 16 + 15) & ~(size_t)15; }

// Election of the cross-check on stream s: per train row the minimum over K1's split partials, scatter-min
// into qbest; for bank pairs whose norms allow d2 >= kSqrtTieMin, the listed rows are then redone exactly.
// fix: device words [4 + nt] (only touched for such pairs).
static int enqueue_election(fm_ctx* ctx, hipStream_t s, const fm_bank* q, const fm_bank* t,
                            const unsigned long long* partial, const RowReducePlan& pl,
                            unsigned long long* qbest, unsigned t_offset, int* bound_reset, unsigned* fix)
{
    const int64_t nt = t->n;
    const int f32 = q->kind == FM_BANK_F32;
    const bool guard = !f32 && fix && sqrt_tie_possible(*q, *t);
    if (guard) HIP_TRY(ctx, hipMemsetAsync(fix, 0, 16, s));
    const int64_t sthreads = (bound_reset && (int64_t)pl.ncols_alloc > nt * 4) ? (int64_t)pl.ncols_alloc : nt * 4;
    hipLaunchKernelGGL(xcheck_scatter_kernel, dim3((unsigned)((sthreads + 255) / 256)), dim3(256), 0, s,
                       partial, pl.nsplit, pl.ncols_alloc, nt, qbest, t_offset, f32, bound_reset, guard ? fix : (unsigned*)nullptr);
    HIP_TRY(ctx, hipGetLastError());
    if (guard) {
        hipLaunchKernelGGL(sqrt_fix_kernel<1>, dim3(kFixGrid), dim3(256), 0, s, (const unsigned*)fix,
                           (const int8_t*)t->rows8, (const int32_t*)t->norm, (const int8_t*)q->rows8, (const int32_t*)q->norm,
                           (int)q->n, qbest, t_offset, (int32_t*)nullptr, (float*)nullptr);
        HIP_TRY(ctx, hipGetLastError());
    }
    return FM_OK;
}

// ---------------------------------------------------------------------------------------
// K2 entry points
// ---------------------------------------------------------------------------------------
// Device-side knn2 into d_idx/d_dist (device pointers).
static int knn2_device(fm_ctx* ctx, const fm_bank* q, const fm_bank* t, int32_t* d_idx, float* d_dist)
{
    const int64_t nq = q->n;
    if (nq == 0) return FM_OK;
    const int f32 = q->kind == FM_BANK_F32;
    if (t->n == 0) {                                     // no train rows: every slot is (-1, +inf)
        hipLaunchKernelGGL(knn2_merge_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const unsigned long long*)nullptr, 0, 0, nq, d_idx, d_dist, f32);
        HIP_TRY(ctx, hipGetLastError());
        return FM_OK;
    }
    RowReducePlan pl;
    int rc;
    if (f32) {
        if ((rc = rowreduce_f32_route(ctx, q, t, 2, &pl)) != FM_OK) return rc;
    } else {
        pl = plan_rowreduce(q->n_pad, t->n_pad, ctx->tune);
        if ((rc = ws_ensure(ctx, &ctx->ws_partial, &ctx->ws_partial_bytes, pl.partial_bytes(2) + pl.bound_bytes() + fix_bytes(nq))) != FM_OK) return rc;
        int* d_bound = nullptr;
        if ((ctx->tune.coop != 0) && pl.nsplit > 1) {
            d_bound = (int*)((char*)ctx->ws_partial + pl.partial_bytes(2));      // bound1 | bound2 (rowreduce.hip)
            if (!ablate_keep_bounds()) HIP_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)d_bound, (int)0x80000000, (size_t)pl.ncols_alloc * 2, ctx->stream));
        }
        HIP_TRY(ctx, hipEventRecord(ctx->ev_k0, ctx->stream));
        HIP_TRY(ctx, launch_rowreduce(*q, *t, 2, pl, (unsigned long long*)ctx->ws_partial, d_bound, (ctx->tune.glds != 0), ctx->stream));
        HIP_TRY(ctx, hipEventRecord(ctx->ev_k1, ctx->stream));
    }
    ctx->kernel_timed = true;
    ctx->pending_pairs += nq * t->n;
    // (output rows whose second best d2 reaches kSqrtTieMin are redone in OpenCV's float32 order)
    unsigned* d_fix = (!f32 && sqrt_tie_possible(*q, *t)) ? (unsigned*)((char*)ctx->ws_partial + pl.partial_bytes(2) + pl.bound_bytes()) : nullptr;
    if (d_fix) HIP_TRY(ctx, hipMemsetAsync(d_fix, 0, 16, ctx->stream));
    hipLaunchKernelGGL(knn2_merge_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const unsigned long long*)ctx->ws_partial, pl.nsplit, pl.ncols_alloc, nq, d_idx, d_dist, f32, d_fix);
    HIP_TRY(ctx, hipGetLastError());
    if (d_fix) {
        hipLaunchKernelGGL(sqrt_fix_kernel<2>, dim3(kFixGrid), dim3(256), 0, ctx->stream, (const unsigned*)d_fix,
                           (const int8_t*)q->rows8, (const int32_t*)q->norm, (const int8_t*)t->rows8, (const int32_t*)t->norm,
                           (int)t->n, (unsigned long long*)nullptr, 0u, d_idx, d_dist);
        HIP_TRY(ctx, hipGetLastError());
    }
    return FM_OK;
}

extern "C" int fm_knn2(fm_ctx* ctx, const fm_bank* q, const fm_bank* t, int32_t* idx, float* dist)
{
    int rc = check_pair(ctx, q, t, "fm_knn2");
    if (rc != FM_OK) return rc;
    const int64_t nq = q->n;
    if (nq > 0 && (!idx || !dist)) return fail(ctx, FM_EINVAL, "fm_knn2: output pointer is NULL");
    if (nq == 0) return FM_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if ((rc = ws_ensure(ctx, &ctx->ws_out, &ctx->ws_out_bytes, (size_t)nq * 16 + 64)) != FM_OK) return rc;
    int32_t* d_idx = (int32_t*)ctx->ws_out;
    float* d_dist = (float*)((char*)ctx->ws_out + (size_t)nq * 8);
    CallScope cs(ctx);
    if ((rc = knn2_device(ctx, q, t, d_idx, d_dist)) != FM_OK) return rc;
    HIP_TRY(ctx, d2h(ctx, idx, d_idx, (size_t)nq * 8));
    HIP_TRY(ctx, d2h(ctx, dist, d_dist, (size_t)nq * 8));
    return cs.finish();
}

extern "C" int fm_knn2_ratio(fm_ctx* ctx, const fm_bank* q, const fm_bank* t, double tau, int64_t cap,
                             int32_t* qidx, int32_t* tidx, float* dist, double* ratio, int64_t* n_accepted)
{
    int rc = check_pair(ctx, q, t, "fm_knn2_ratio");
    if (rc != FM_OK) return rc;
    if (n_accepted) *n_accepted = 0;
    const int64_t nq = q->n;
    if (nq == 0) return FM_OK;
    if (cap < 0 || !qidx || !tidx || !dist || !ratio) return fail(ctx, FM_EINVAL, "fm_knn2_ratio: bad output arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int nblk = (int)((nq + 255) / 256);
    const int64_t ccap = cap < nq ? cap : nq;
    // knn lists | per-q tidx, dist, ratio, pass | block counts, count | compacted outputs
    size_t off = 0;
    auto carve = [&](size_t b) { size_t o = off; off += (b + 15) & ~(size_t)15; return o; };
    const size_t o_i2 = carve((size_t)nq * 8), o_d2 = carve((size_t)nq * 8), o_ti = carve((size_t)nq * 4), o_di = carve((size_t)nq * 4);
    const size_t o_ra = carve((size_t)nq * 8), o_pa = carve((size_t)nq), o_bc = carve((size_t)nblk * 4), o_cnt = carve(16);
    const size_t o_cq = carve((size_t)ccap * 4), o_ct = carve((size_t)ccap * 4), o_cd = carve((size_t)ccap * 4), o_cr = carve((size_t)ccap * 8);
    if ((rc = ws_ensure(ctx, &ctx->ws_out, &ctx->ws_out_bytes, off + 64)) != FM_OK) return rc;
    char* b = (char*)ctx->ws_out;
    CallScope cs(ctx);
    if ((rc = knn2_device(ctx, q, t, (int32_t*)(b + o_i2), (float*)(b + o_d2))) != FM_OK) return rc;
    hipLaunchKernelGGL(lowe_kernel, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, (const int32_t*)(b + o_i2),
                       (const float*)(b + o_d2), nq, tau, (int32_t*)(b + o_ti), (float*)(b + o_di), (double*)(b + o_ra),
                       (uint8_t*)(b + o_pa), (int*)(b + o_bc));
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(compact_kernel, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, (const int32_t*)(b + o_ti),
                       (const float*)(b + o_di), (const double*)(b + o_ra), (const uint8_t*)(b + o_pa), (const int*)(b + o_bc),
                       nq, ccap, (int32_t*)(b + o_cq), (int32_t*)(b + o_ct), (float*)(b + o_cd), (double*)(b + o_cr),
                       (unsigned long long*)(b + o_cnt));
    HIP_TRY(ctx, hipGetLastError());
    unsigned long long cnt = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&cnt, b + o_cnt, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const size_t m = (size_t)((int64_t)cnt < ccap ? (int64_t)cnt : ccap);
    if (m) {
        HIP_TRY(ctx, d2h(ctx, qidx, b + o_cq, m * 4));
        HIP_TRY(ctx, d2h(ctx, tidx, b + o_ct, m * 4));
        HIP_TRY(ctx, d2h(ctx, dist, b + o_cd, m * 4));
        HIP_TRY(ctx, d2h(ctx, ratio, b + o_cr, m * 8));
    }
    rc = cs.finish();
    if (rc != FM_OK) return rc;
    if (n_accepted) *n_accepted = (int64_t)cnt;
    return FM_OK;
}

extern "C" int fm_self_dist(fm_ctx* ctx, const fm_bank* bank, double* selfdist)
{
    int rc = check_pair(ctx, bank, bank, "fm_self_dist");
    if (rc != FM_OK) return rc;
    const int64_t n = bank->n;
    if (n == 0) return FM_OK;
    if (!selfdist) return fail(ctx, FM_EINVAL, "fm_self_dist: output pointer is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if ((rc = ws_ensure(ctx, &ctx->ws_out, &ctx->ws_out_bytes, (size_t)n * 24 + 64)) != FM_OK) return rc;
    int32_t* d_idx = (int32_t*)ctx->ws_out;
    float* d_dist = (float*)((char*)ctx->ws_out + (size_t)n * 8);
    double* d_sd = (double*)((char*)ctx->ws_out + (size_t)n * 16);
    CallScope cs(ctx);
    if ((rc = knn2_device(ctx, bank, bank, d_idx, d_dist)) != FM_OK) return rc;
    hipLaunchKernelGGL(selfdist_from_knn_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const float*)d_dist, n, d_sd);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, d2h(ctx, selfdist, d_sd, (size_t)n * 8));
    return cs.finish();
}

// ---------------------------------------------------------------------------------------
// X1 (+R1) entry points
// ---------------------------------------------------------------------------------------
// Workspace of one bank pair in flight (async calls): partial | bound | qbest | tidx | dist | ratio | pass | block counts
// "no consumer stream": NULL is a stream (the null stream, which PyTorch's default stream is)
static const hipStream_t kNoStream = (hipStream_t)FM_NO_STREAM;

struct SlotLayout {
    size_t pbytes, a_qbest, a_tidx, a_dist, a_ratio, a_pass, a_bc, a_fix, a_end;
    int nblk;
};

static SlotLayout slot_layout(int64_t nq, int64_t nt, const RowReducePlan& pl)
{
    SlotLayout L;
    L.nblk = (int)((nq + 255) / 256);
    L.pbytes = (pl.partial_bytes(1) + 15) & ~(size_t)15;
    const size_t bbytes = ((size_t)pl.ncols_alloc * 4 + 15) & ~(size_t)15;
    L.a_qbest = L.pbytes + bbytes; L.a_tidx = L.a_qbest + (size_t)nq * 8; L.a_dist = L.a_tidx + (size_t)nq * 4;
    L.a_ratio = (L.a_dist + (size_t)nq * 4 + 7) & ~(size_t)7; L.a_pass = L.a_ratio + (size_t)nq * 8;
    L.a_bc = (L.a_pass + (size_t)nq + 15) & ~(size_t)15;
    L.a_fix = (L.a_bc + (size_t)L.nblk * 4 + 16 + 15) & ~(size_t)15;       // tie-repair list (enqueue_election)
    L.a_end = L.a_fix + fix_bytes(nt);
    return L;
}

// Make the slot ready for a K1 on stream ks: wait for the slot's previous tail (it reads partial /
// qbest and re-arms bound), (re)allocate and initialise the arrays when the shape changed.
static int slot_prepare(fm_ctx* ctx, fm_ctx::AsyncSlot& sl, const SlotLayout& L, int64_t nq, const RowReducePlan& pl, hipStream_t ks)
{
    if (sl.in_use) HIP_TRY(ctx, hipStreamWaitEvent(ks, sl.tail_done, 0));
    if (L.a_end > sl.bytes || sl.nq != nq || sl.ncols_alloc != pl.ncols_alloc || sl.partial_bytes != (int64_t)L.pbytes) {
        if (sl.in_use) HIP_TRY(ctx, hipEventSynchronize(sl.tail_done));
        if (L.a_end > sl.bytes) {
            if (sl.ws) { HIP_TRY(ctx, hipFree(sl.ws)); sl.ws = nullptr; sl.bytes = 0; }
            HIP_TRY(ctx, hipMalloc(&sl.ws, L.a_end + L.a_end / 4));
            sl.bytes = L.a_end + L.a_end / 4;
        }
        HIP_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)((char*)sl.ws + L.pbytes), (int)0x80000000, (size_t)pl.ncols_alloc, ks));
        HIP_TRY(ctx, hipMemsetAsync((char*)sl.ws + L.a_qbest, 0xff, (size_t)nq * 8, ks));
        sl.nq = nq; sl.ncols_alloc = pl.ncols_alloc; sl.partial_bytes = (int64_t)L.pbytes;
    }
    return FM_OK;
}

static int take_timer(fm_ctx* ctx, fm_ctx::PendingTimer* tm)
{
    if (!ctx->timer_pool.empty()) { *tm = ctx->timer_pool.back(); ctx->timer_pool.pop_back(); return FM_OK; }
    if (ctx->pending.size() >= 1024) {                 // bound the number of live events
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (hipStream_t ts : ctx->tails) if (ts) HIP_TRY(ctx, hipStreamSynchronize(ts));
        drain_pending(ctx);
        *tm = ctx->timer_pool.back(); ctx->timer_pool.pop_back();
        return FM_OK;
    }
    HIP_TRY(ctx, hipEventCreate(&tm->c0)); HIP_TRY(ctx, hipEventCreate(&tm->c1));
    HIP_TRY(ctx, hipEventCreate(&tm->k0)); HIP_TRY(ctx, hipEventCreate(&tm->k1));
    return FM_OK;
}

// The small kernels behind a K1 (election, decode + ratio test, ordered compaction), on stream ts,
// which must already wait for that K1.  Host outputs (a_* = device aliases of page-locked memory) or,
// with dev_rows, the 12-byte rows of the result gather.  Leaves the slot's bound[] / qbest[] clean.
static int enqueue_tail(fm_ctx* ctx, hipStream_t ts, fm_ctx::AsyncSlot& sl, const SlotLayout& L, const fm_bank* q,
                        const fm_bank* t, int64_t nq, int64_t nt, const RowReducePlan& pl, double tau, int64_t compact_cap,
                        void* a_q, void* a_t, void* a_d, void* a_r, void* a_c,
                        int32_t* dev_rows, long long* dev_count, hipStream_t consumer)
{
    char* sb = (char*)sl.ws;
    unsigned long long* s_partial = (unsigned long long*)sb;
    int* s_bound = (int*)(sb + L.pbytes);
    unsigned long long* s_qbest = (unsigned long long*)(sb + L.a_qbest);
    int32_t* s_tidx = (int32_t*)(sb + L.a_tidx);
    float* s_dist = (float*)(sb + L.a_dist);
    double* s_ratio = (double*)(sb + L.a_ratio);
    uint8_t* s_pass = (uint8_t*)(sb + L.a_pass);
    int* s_bc = (int*)(sb + L.a_bc);
    if (nt > 0) {
        int rc = enqueue_election(ctx, ts, q, t, s_partial, pl, s_qbest, 0u, s_bound, (unsigned*)(sb + L.a_fix));
        if (rc != FM_OK) return rc;
    }
    hipLaunchKernelGGL(xcheck_finalize_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, ts,
                       (const unsigned long long*)s_qbest, nq, (const double*)q->selfdist, tau, s_tidx, s_dist, s_ratio,
                       s_pass, (unsigned long long*)nullptr, s_bc, s_qbest);
    if (dev_rows) {
        // the rows go to the caller's device buffers, which a consumer stream (the result gather)
        // reads: the compaction waits for what that stream has been given so far (the gather that
        // last read these buffers), and the stream waits for the compaction
        if (consumer != kNoStream) {
            HIP_TRY(ctx, hipEventRecord(ctx->ev_consumer, consumer));
            HIP_TRY(ctx, hipStreamWaitEvent(ts, ctx->ev_consumer, 0));
        }
        hipLaunchKernelGGL(compact_rows_kernel, dim3((unsigned)L.nblk), dim3(256), 0, ts,
                           (const int32_t*)s_tidx, (const float*)s_dist, (const uint8_t*)s_pass,
                           (const int*)s_bc, nq, compact_cap, dev_rows, dev_count, (unsigned long long*)a_c);
    } else {
        hipLaunchKernelGGL(compact_kernel, dim3((unsigned)L.nblk), dim3(256), 0, ts,
                           (const int32_t*)s_tidx, (const float*)s_dist, (const double*)s_ratio, (const uint8_t*)s_pass,
                           (const int*)s_bc, nq, compact_cap < nq ? compact_cap : nq, (int32_t*)a_q, (int32_t*)a_t,
                           (float*)a_d, (double*)a_r, (unsigned long long*)a_c);
    }
    HIP_TRY(ctx, hipGetLastError());
    return FM_OK;
}

static int xcheck_common(fm_ctx* ctx, const fm_bank* q, const fm_bank* t, bool with_ratio, double tau,
                         int32_t* tidx, float* dist, double* ratio, uint8_t* pass, int64_t* n_pass,
                         const char* who, int64_t compact_cap = -1, int32_t* c_qidx = nullptr,
                         int32_t* dev_rows = nullptr, long long* dev_count = nullptr, bool async_mode = false,
                         hipStream_t consumer = kNoStream)
{
    const bool compact = compact_cap >= 0;
    const bool to_device = dev_rows != nullptr;
    int rc = check_pair(ctx, q, t, who);
    if (rc != FM_OK) return rc;
    const int f32 = q->kind == FM_BANK_F32;
    const int64_t nq = q->n, nt = t->n;
    if (n_pass) *n_pass = 0;
    if (nq == 0) return FM_OK;
    if (!to_device && (!tidx || !dist || (compact && (!c_qidx || !ratio)))) return fail(ctx, FM_EINVAL, std::string(who) + ": output pointer is NULL");
    if (with_ratio && !q->selfdist) return fail(ctx, FM_EINVAL, std::string(who) + ": query bank has no self distances (fm_bank_set_selfdist)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // outputs: qbest u64[nq] | tidx i32[nq] | dist f32[nq] | ratio f64[nq] | pass u8[nq] | count u64
    const size_t o_qbest = 0, o_tidx = (size_t)nq * 8, o_dist = o_tidx + (size_t)nq * 4;
    const size_t o_ratio = (o_dist + (size_t)nq * 4 + 7) & ~(size_t)7, o_pass = o_ratio + (size_t)nq * 8;
    const size_t o_cnt = (o_pass + (size_t)nq + 15) & ~(size_t)15;
    // compaction: block counts | compacted qidx, tidx, dist, ratio
    const int nblk = (int)((nq + 255) / 256);
    const int64_t ccap = (compact && !to_device) ? (compact_cap < nq ? compact_cap : nq) : 0;
    const size_t o_bc = o_cnt + 16, o_cq = (o_bc + (size_t)nblk * 4 + 15) & ~(size_t)15, o_ct = o_cq + (size_t)ccap * 4;
    const size_t o_cd = o_ct + (size_t)ccap * 4, o_cr = (o_cd + (size_t)ccap * 4 + 7) & ~(size_t)7;
    if ((rc = ws_ensure(ctx, &ctx->ws_out, &ctx->ws_out_bytes, o_cr + (size_t)ccap * 8 + 16)) != FM_OK) return rc;
    char* base = (char*)ctx->ws_out;
    unsigned long long* d_qbest = (unsigned long long*)(base + o_qbest);
    int32_t* d_tidx = (int32_t*)(base + o_tidx);
    float* d_dist = (float*)(base + o_dist);
    double* d_ratio = (double*)(base + o_ratio);
    uint8_t* d_pass = (uint8_t*)(base + o_pass);
    unsigned long long* d_cnt = (unsigned long long*)(base + o_cnt);

    // reverse NN: output rows = train rows, reduced over the query rows
    RowReducePlan pl;
    int* d_bound = nullptr;
    if (!f32) {
        pl = plan_rowreduce(t->n_pad, q->n_pad, ctx->tune);
        if ((rc = ws_ensure(ctx, &ctx->ws_partial, &ctx->ws_partial_bytes, pl.partial_bytes(1) + pl.bound_bytes() + fix_bytes(nt))) != FM_OK) return rc;
        if ((ctx->tune.coop != 0) && pl.nsplit > 1) d_bound = (int*)((char*)ctx->ws_partial + pl.partial_bytes(1));
    }

    if (async_mode) {
        // Enqueue and return: outputs (and the count) are page-locked caller memory the compaction
        // kernel writes directly; the events of this call are read at fm_sync.
        void* a_q = pinned_device_alias(c_qidx); void* a_t = pinned_device_alias(tidx);
        void* a_d = pinned_device_alias(dist);   void* a_r = pinned_device_alias(ratio);
        void* a_c = n_pass ? pinned_device_alias(n_pass) : nullptr;
        if (f32) return fail(ctx, FM_EINVAL, std::string(who) + ": needs integer-valued banks");
        if (to_device ? (n_pass && !a_c) : (!a_q || !a_t || !a_d || !a_r || !a_c))
            return fail(ctx, FM_EINVAL, std::string(who) + ": host outputs must be page-locked (fm_host_alloc)");
        fm_ctx::PendingTimer tm;
        if ((rc = take_timer(ctx, &tm)) != FM_OK) return rc;
        tm.timed = nt > 0;
        tm.pairs = nq * nt;
        fm_ctx::AsyncSlot& sl = ctx->aslot[ctx->aslot_next];
        ctx->aslot_next ^= 1;
        const SlotLayout L = slot_layout(nq, nt, pl);
        if ((rc = slot_prepare(ctx, sl, L, nq, pl, ctx->stream)) != FM_OK) return rc;
        const bool coop = (ctx->tune.coop != 0) && pl.nsplit > 1;
        // Every event record is a packet the K1 launches of consecutive calls queue behind; the
        // start-of-kernel event is therefore taken for every async_time_every-th call only (those
        // calls are the ones fm_get_stats accounts as timed K1 launches; FM_ASYNC_TIME_EVERY, default 4).
        const int time_every = ctx->tune.async_time_every;
        const bool timed_call = time_every > 0 && (ctx->async_calls++ % time_every) == 0;
        tm.timed = tm.timed && timed_call;
        tm.call_timed = timed_call;
        if (timed_call) HIP_TRY(ctx, hipEventRecord(tm.k0, ctx->stream));
        if (nt > 0)
            HIP_TRY(ctx, launch_rowreduce(*t, *q, 1, pl, (unsigned long long*)sl.ws, coop ? (int*)((char*)sl.ws + L.pbytes) : nullptr,
                                          (ctx->tune.glds != 0), ctx->stream));
        // (untimed calls hand over through the slot's own event, created without timing)
        hipEvent_t handover = timed_call ? tm.k1 : sl.k_done;
        HIP_TRY(ctx, hipEventRecord(handover, ctx->stream));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream_tail, handover, 0));
        if ((rc = enqueue_tail(ctx, ctx->stream_tail, sl, L, q, t, nq, nt, pl, tau, compact_cap, a_q, a_t, a_d, a_r, a_c,
                               to_device ? dev_rows : nullptr, dev_count, consumer)) != FM_OK) return rc;
        if (timed_call) HIP_TRY(ctx, hipEventRecord(tm.c1, ctx->stream_tail));
        HIP_TRY(ctx, hipEventRecord(sl.tail_done, ctx->stream_tail));
        if (to_device && consumer != kNoStream) HIP_TRY(ctx, hipStreamWaitEvent(consumer, sl.tail_done, 0));
        ctx->rows_stream = to_device ? ctx->stream_tail : ctx->stream;
        sl.in_use = true;
        ctx->pending.push_back(tm);
        return FM_OK;
    }
    CallScope cs(ctx);
    HIP_TRY(ctx, hipMemsetAsync(d_qbest, 0xff, (size_t)nq * 8, ctx->stream));
    if (!compact) HIP_TRY(ctx, hipMemsetAsync(d_cnt, 0, 8, ctx->stream));
    if (nt > 0) {
        if (d_bound)
            if (!ablate_keep_bounds()) HIP_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)d_bound, (int)0x80000000, (size_t)pl.ncols_alloc, ctx->stream));
        if (f32) {
            if ((rc = rowreduce_f32_route(ctx, t, q, 1, &pl)) != FM_OK) return rc;
        } else {
            HIP_TRY(ctx, hipEventRecord(ctx->ev_k0, ctx->stream));
            HIP_TRY(ctx, launch_rowreduce(*t, *q, 1, pl, (unsigned long long*)ctx->ws_partial, d_bound, (ctx->tune.glds != 0), ctx->stream));
            HIP_TRY(ctx, hipEventRecord(ctx->ev_k1, ctx->stream));
        }
        ctx->kernel_timed = true;
        ctx->pending_pairs += nq * nt;
        // (the float32 route's partial layout has no tie list behind it, and needs none)
        unsigned* d_fix = f32 ? nullptr : (unsigned*)((char*)ctx->ws_partial + pl.partial_bytes(1) + pl.bound_bytes());
        if ((rc = enqueue_election(ctx, ctx->stream, q, t, (const unsigned long long*)ctx->ws_partial, pl, d_qbest, 0u, nullptr, d_fix)) != FM_OK) return rc;
    }
    hipLaunchKernelGGL(xcheck_finalize_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const unsigned long long*)d_qbest, nq, with_ratio ? (const double*)q->selfdist : (const double*)nullptr,
                       tau, d_tidx, d_dist, with_ratio ? d_ratio : (double*)nullptr,
                       with_ratio ? d_pass : (uint8_t*)nullptr, with_ratio ? d_cnt : (unsigned long long*)nullptr,
                       compact ? (int*)(base + o_bc) : (int*)nullptr);
    HIP_TRY(ctx, hipGetLastError());
    unsigned long long cnt = 0;
    if (to_device) {
        ctx->rows_stream = ctx->stream;
        // accepted matches stay on the device as packed rows (multi-GPU gather input)
        void* a_c = ctx->h_scratch ? pinned_device_alias(ctx->h_scratch) : nullptr;
        hipLaunchKernelGGL(compact_rows_kernel, dim3((unsigned)nblk), dim3(256), 0, ctx->stream,
                           (const int32_t*)d_tidx, (const float*)d_dist, (const uint8_t*)d_pass,
                           (const int*)(base + o_bc), nq, compact_cap, dev_rows, dev_count, (unsigned long long*)a_c);
        HIP_TRY(ctx, hipGetLastError());
        if (!a_c) HIP_TRY(ctx, hipMemcpyAsync(&cnt, dev_count, 8, hipMemcpyDeviceToHost, ctx->stream));
        rc = cs.finish();
        if (rc != FM_OK) return rc;
        if (n_pass) *n_pass = a_c ? (int64_t)ctx->h_scratch[0] : (int64_t)cnt;
        return FM_OK;
    }
    if (compact) {
        // caller-owned page-locked outputs: the compaction writes them (and the count) directly,
        // no staging copies and a single synchronisation
        void* a_q = pinned_device_alias(c_qidx); void* a_t = pinned_device_alias(tidx);
        void* a_d = pinned_device_alias(dist);   void* a_r = pinned_device_alias(ratio);
        void* a_c = ctx->h_scratch ? pinned_device_alias(ctx->h_scratch) : nullptr;
        if (a_q && a_t && a_d && a_r && a_c && compact_cap <= nq) {
            hipLaunchKernelGGL(compact_kernel, dim3((unsigned)nblk), dim3(256), 0, ctx->stream,
                               (const int32_t*)d_tidx, (const float*)d_dist, (const double*)d_ratio, (const uint8_t*)d_pass,
                               (const int*)(base + o_bc), nq, ccap, (int32_t*)a_q, (int32_t*)a_t,
                               (float*)a_d, (double*)a_r, (unsigned long long*)a_c);
            HIP_TRY(ctx, hipGetLastError());
            rc = cs.finish();
            if (rc != FM_OK) return rc;
            if (n_pass) *n_pass = (int64_t)ctx->h_scratch[0];
            return FM_OK;
        }
        hipLaunchKernelGGL(compact_kernel, dim3((unsigned)nblk), dim3(256), 0, ctx->stream,
                           (const int32_t*)d_tidx, (const float*)d_dist, (const double*)d_ratio, (const uint8_t*)d_pass,
                           (const int*)(base + o_bc), nq, ccap, (int32_t*)(base + o_cq), (int32_t*)(base + o_ct),
                           (float*)(base + o_cd), (double*)(base + o_cr), d_cnt);
        HIP_TRY(ctx, hipGetLastError());
        // the count must be known before the copies can be sized: one tiny synchronous read
        HIP_TRY(ctx, hipMemcpyAsync(&cnt, d_cnt, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        const size_t m = (size_t)((int64_t)cnt < ccap ? (int64_t)cnt : ccap);
        if (m) {
            HIP_TRY(ctx, d2h(ctx, c_qidx, base + o_cq, m * 4));
            HIP_TRY(ctx, d2h(ctx, tidx, base + o_ct, m * 4));
            HIP_TRY(ctx, d2h(ctx, dist, base + o_cd, m * 4));
            HIP_TRY(ctx, d2h(ctx, ratio, base + o_cr, m * 8));
        }
        rc = cs.finish();
        if (rc != FM_OK) return rc;
        if (n_pass) *n_pass = (int64_t)cnt;
        return FM_OK;
    }
    HIP_TRY(ctx, d2h(ctx, tidx, d_tidx, (size_t)nq * 4));
    HIP_TRY(ctx, d2h(ctx, dist, d_dist, (size_t)nq * 4));
    if (with_ratio) {
        if (ratio) HIP_TRY(ctx, d2h(ctx, ratio, d_ratio, (size_t)nq * 8));
        if (pass) HIP_TRY(ctx, d2h(ctx, pass, d_pass, (size_t)nq));
        HIP_TRY(ctx, hipMemcpyAsync(&cnt, d_cnt, 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    rc = cs.finish();
    if (rc != FM_OK) return rc;
    if (n_pass) *n_pass = (int64_t)cnt;
    return FM_OK;
}

// X1 up to the election, for a train set sharded over ranks: keys[q] =
// (distance key << 32) | (t_offset + local train row) of the closest train row OF THIS BANK that
// elects q, ~0 if none.  The element-wise minimum of the ranks' key arrays is the key array of
// the unsharded call (the key carries the global index, so ties break as on one GPU).
static int xcheck1_keys_common(fm_ctx* ctx, const fm_bank* q, const fm_bank* t, int64_t t_offset, uint64_t* keys, bool keys_on_device);

extern "C" int fm_xcheck1_keys(fm_ctx* ctx, const fm_bank* q, const fm_bank* t, int64_t t_offset, uint64_t* keys)
{
    return xcheck1_keys_common(ctx, q, t, t_offset, keys, false);
}

extern "C" int fm_xcheck1_keys_dev(fm_ctx* ctx, const fm_bank* q, const fm_bank* t, int64_t t_offset, uint64_t* d_keys)
{
    if (ctx && d_keys) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, d_keys) != hipSuccess || at.type != hipMemoryTypeDevice) {
            (void)hipGetLastError();
            return fail(ctx, FM_EINVAL, "fm_xcheck1_keys_dev: d_keys must be device memory");
        }
    }
    return xcheck1_keys_common(ctx, q, t, t_offset, d_keys, true);
}

static int xcheck1_keys_common(fm_ctx* ctx, const fm_bank* q, const fm_bank* t, int64_t t_offset, uint64_t* keys, bool keys_on_device)
{
    int rc = check_pair(ctx, q, t, "fm_xcheck1_keys");
    if (rc != FM_OK) return rc;
    const int64_t nq = q->n, nt = t->n;
    if (nq == 0) return FM_OK;
    if (!keys) return fail(ctx, FM_EINVAL, "fm_xcheck1_keys: output pointer is NULL");
    if (t_offset < 0 || t_offset + nt > (int64_t)UINT32_MAX) return fail(ctx, FM_EINVAL, "fm_xcheck1_keys: t_offset + rows must fit 32 bits");
    const int f32 = q->kind == FM_BANK_F32;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if ((rc = ws_ensure(ctx, &ctx->ws_out, &ctx->ws_out_bytes, (size_t)nq * 8 + 64)) != FM_OK) return rc;
    unsigned long long* d_qbest = (unsigned long long*)ctx->ws_out;
    RowReducePlan pl;
    int* d_bound = nullptr;
    if (!f32) {
        pl = plan_rowreduce(t->n_pad, q->n_pad, ctx->tune);
        if ((rc = ws_ensure(ctx, &ctx->ws_partial, &ctx->ws_partial_bytes, pl.partial_bytes(1) + pl.bound_bytes() + fix_bytes(nt))) != FM_OK) return rc;
        if ((ctx->tune.coop != 0) && pl.nsplit > 1) d_bound = (int*)((char*)ctx->ws_partial + pl.partial_bytes(1));
    }
    CallScope cs(ctx);
    HIP_TRY(ctx, hipMemsetAsync(d_qbest, 0xff, (size_t)nq * 8, ctx->stream));
    if (nt > 0) {
        if (d_bound)
            if (!ablate_keep_bounds()) HIP_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)d_bound, (int)0x80000000, (size_t)pl.ncols_alloc, ctx->stream));
        if (f32) {
            if ((rc = rowreduce_f32_route(ctx, t, q, 1, &pl)) != FM_OK) return rc;
        } else {
            HIP_TRY(ctx, hipEventRecord(ctx->ev_k0, ctx->stream));
            HIP_TRY(ctx, launch_rowreduce(*t, *q, 1, pl, (unsigned long long*)ctx->ws_partial, d_bound, (ctx->tune.glds != 0), ctx->stream));
            HIP_TRY(ctx, hipEventRecord(ctx->ev_k1, ctx->stream));
        }
        ctx->kernel_timed = true;
        ctx->pending_pairs += nq * nt;
        unsigned* d_fix = f32 ? nullptr : (unsigned*)((char*)ctx->ws_partial + pl.partial_bytes(1) + pl.bound_bytes());
        if ((rc = enqueue_election(ctx, ctx->stream, q, t, (const unsigned long long*)ctx->ws_partial, pl, d_qbest, (unsigned)t_offset, nullptr, d_fix)) != FM_OK) return rc;
    }
    if (keys_on_device) HIP_TRY(ctx, hipMemcpyAsync(keys, d_qbest, (size_t)nq * 8, hipMemcpyDeviceToDevice, ctx->stream));
    else HIP_TRY(ctx, d2h(ctx, keys, d_qbest, (size_t)nq * 8));
    return cs.finish();
}

extern "C" int fm_xcheck1(fm_ctx* ctx, const fm_bank* q, const fm_bank* t, int32_t* tidx, float* dist)
{
    return xcheck_common(ctx, q, t, false, 0.0, tidx, dist, nullptr, nullptr, nullptr, "fm_xcheck1");
}

extern "C" int fm_match_ratio(fm_ctx* ctx, const fm_bank* q, const fm_bank* t, double tau, int32_t* tidx,
                              float* dist, double* ratio, uint8_t* pass, int64_t* n_pass)
{
    return xcheck_common(ctx, q, t, true, tau, tidx, dist, ratio, pass, n_pass, "fm_match_ratio");
}

extern "C" int fm_match_accepted(fm_ctx* ctx, const fm_bank* q, const fm_bank* t, double tau, int64_t cap,
                                 int32_t* qidx, int32_t* tidx, float* dist, double* ratio, int64_t* n_accepted)
{
    if (cap < 0) return fail(ctx, FM_EINVAL, "fm_match_accepted: cap < 0");
    return xcheck_common(ctx, q, t, true, tau, tidx, dist, ratio, nullptr, n_accepted, "fm_match_accepted", cap, qidx);
}

extern "C" int fm_match_accepted_async(fm_ctx* ctx, const fm_bank* q, const fm_bank* t, double tau, int64_t cap,
                                       int32_t* qidx, int32_t* tidx, float* dist, double* ratio, int64_t* n_accepted)
{
    if (cap < 0) return fail(ctx, FM_EINVAL, "fm_match_accepted_async: cap < 0");
    if (!n_accepted) return fail(ctx, FM_EINVAL, "fm_match_accepted_async: n_accepted is NULL");
    if (q && q->n == 0) { *n_accepted = 0; }
    return xcheck_common(ctx, q, t, true, tau, tidx, dist, ratio, nullptr, n_accepted, "fm_match_accepted_async", cap, qidx,
                         nullptr, nullptr, true);
}

// n image pairs in one call, enqueued like fm_match_accepted_async; runs of consecutive pairs of one
// shape go through K1 TOGETHER (rowreduce_batch_kernel: up to `batch_group` pairs per launch, at most 16), each
// pair's small kernels follow on one of three tail streams beside the next group's K1.
// d_rows != NULL: device outputs (fm_match_accepted_dev_batch): pair i's rows at d_rows + i * cap * 3, its
// count at d_counts + i, optionally also in the page-locked words h_counts[i]; host outputs otherwise.
static int batch_common(fm_ctx* ctx, int32_t n, const fm_bank* const* q, const fm_bank* const* t, double tau,
                        int64_t cap, int32_t* const* qidx, int32_t* const* tidx, float* const* dist,
                        double* const* ratio, int64_t* const* n_accepted,
                        int32_t* d_rows, int64_t* d_counts, int64_t* h_counts, hipStream_t consumer);

extern "C" int fm_match_accepted_batch(fm_ctx* ctx, int32_t n, const fm_bank* const* q, const fm_bank* const* t, double tau,
                                       int64_t cap, int32_t* const* qidx, int32_t* const* tidx, float* const* dist,
                                       double* const* ratio, int64_t* const* n_accepted)
{
    if (!ctx) return fail(nullptr, FM_EINVAL, "fm_match_accepted_batch: ctx is NULL");
    if (n > 0 && (!qidx || !tidx || !dist || !ratio || !n_accepted)) return fail(ctx, FM_EINVAL, "fm_match_accepted_batch: NULL argument");
    for (int i = 0; i < n; ++i) if (!n_accepted[i]) return fail(ctx, FM_EINVAL, "fm_match_accepted_batch: n_accepted is NULL");
    return batch_common(ctx, n, q, t, tau, cap, qidx, tidx, dist, ratio, n_accepted, nullptr, nullptr, nullptr, kNoStream);
}

extern "C" int fm_match_accepted_dev_batch(fm_ctx* ctx, int32_t n, const fm_bank* const* q, const fm_bank* const* t, double tau,
                                           int64_t cap, int32_t* d_rows, int64_t* d_counts, int64_t* h_counts, void* consumer_stream)
{
    if (!ctx) return fail(nullptr, FM_EINVAL, "fm_match_accepted_dev_batch: ctx is NULL");
    if (n > 0) {
        if (!d_rows || !d_counts) return fail(ctx, FM_EINVAL, "fm_match_accepted_dev_batch: device output pointer is NULL");
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, d_rows) != hipSuccess || at.type != hipMemoryTypeDevice ||
            hipPointerGetAttributes(&at, d_counts) != hipSuccess || at.type != hipMemoryTypeDevice) {
            (void)hipGetLastError();
            return fail(ctx, FM_EINVAL, "fm_match_accepted_dev_batch: d_rows / d_counts must be device memory");
        }
        if (h_counts && !pinned_device_alias(h_counts)) return fail(ctx, FM_EINVAL, "fm_match_accepted_dev_batch: h_counts must be page-locked (fm_host_alloc)");
    }
    return batch_common(ctx, n, q, t, tau, cap, nullptr, nullptr, nullptr, nullptr, nullptr, d_rows, d_counts, h_counts,
                        (hipStream_t)consumer_stream);
}

static int batch_common(fm_ctx* ctx, int32_t n, const fm_bank* const* q, const fm_bank* const* t, double tau,
                        int64_t cap, int32_t* const* qidx, int32_t* const* tidx, float* const* dist,
                        double* const* ratio, int64_t* const* n_accepted,
                        int32_t* d_rows, int64_t* d_counts, int64_t* h_counts, hipStream_t consumer)
{
    const bool to_dev = d_rows != nullptr;
    if (n < 0 || cap < 0) return fail(ctx, FM_EINVAL, "fm_match_accepted_batch: n < 0 or cap < 0");
    if (n == 0) return FM_OK;
    if (!q || !t) return fail(ctx, FM_EINVAL, "fm_match_accepted_batch: NULL argument");
    int rc;
    // every pair is checked BEFORE anything is enqueued: a bad pair in the middle must not leave earlier pairs
    // in flight and outputs half written
    for (int i = 0; i < n; ++i) {
        if ((rc = check_pair(ctx, q[i], t[i], "fm_match_accepted_batch")) != FM_OK) return rc;
        if (q[i]->n > 0 && !q[i]->selfdist)
            return fail(ctx, FM_EINVAL, "fm_match_accepted_batch: a query bank has no self distances (fm_bank_set_selfdist)");
        if (!to_dev && q[i]->n > 0) {
            if (!qidx[i] || !tidx[i] || !dist[i] || !ratio[i]) return fail(ctx, FM_EINVAL, "fm_match_accepted_batch: output pointer is NULL");
            if (!pinned_device_alias(qidx[i]) || !pinned_device_alias(tidx[i]) || !pinned_device_alias(dist[i]) ||
                !pinned_device_alias(ratio[i]) || !pinned_device_alias(n_accepted[i]))
                return fail(ctx, FM_EINVAL, "fm_match_accepted_batch: host outputs must be page-locked (fm_host_alloc)");
        }
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (to_dev && consumer != kNoStream) {
        // the compactions write buffers the consumer stream reads (the previous step's collective): they wait
        // for what that stream has been given so far
        HIP_TRY(ctx, hipEventRecord(ctx->ev_consumer, consumer));
        for (hipStream_t ts : ctx->tails) HIP_TRY(ctx, hipStreamWaitEvent(ts, ctx->ev_consumer, 0));
    }
    // options batch_group: most pairs per launch (default 8); batch_tail: size of the short launch a run ends
    // with (default 2; 0 = none, for callers that enqueue the next batch before they wait for this one:
    // the small kernels of the last launch then overlap the next batch, see fm_mark / fm_wait)
    const int group_max = ctx->tune.batch_group, tail_n = ctx->tune.batch_tail;
    auto batchable = [&](int i) {
        return q[i]->kind != FM_BANK_F32 && q[i]->n > 0 && t[i]->n > 0 && q[i]->selfdist != nullptr;
    };
    static_assert(fm_ctx::kBatchSlots >= 2 * kRRBatchMax, "two full launches must find distinct workspaces");
    if ((int)ctx->bslot.size() < fm_ctx::kBatchSlots) ctx->bslot.resize((size_t)fm_ctx::kBatchSlots);
    int i = 0;
    while (i < n) {
        // run of same-shape pairs from i on, then this launch's share of it: a launch's small kernels overlap
        // the NEXT launch, so only the last launch's are exposed -- the run ends with a short launch (2 pairs)
        int run = 1;
        if (batchable(i))
            while (i + run < n && batchable(i + run) && q[i + run]->n_pad == q[i]->n_pad && t[i + run]->n_pad == t[i]->n_pad) ++run;
        int g = run;
        if (run > group_max + tail_n) g = group_max;
        else if (run > 4 && tail_n > 0) g = run - tail_n < group_max ? run - tail_n : group_max;
        if (g > group_max) g = group_max;
        if (g < 1) g = 1;
        RowReducePlan pl;
        if (g > 1) {
            pl = plan_rowreduce(t[i]->n_pad, q[i]->n_pad, ctx->tune);
            if (pl.nb != 4 || pl.nw != 8 || !(ctx->tune.glds != 0) || pl.nbuf == 2) g = 1;     // shapes the batched kernel is not built for
        }
        if (g == 1) {                  // an odd pair: the single-pair call (which also reports its errors)
            // float32-route pairs have no enqueue-only form: they run synchronously, in place (their outputs
            // are complete when the batch call returns; the pairs around them stay asynchronous)
            const bool in_place = q[i]->kind == FM_BANK_F32 && q[i]->n > 0;
            if (to_dev) {
                int64_t* hc = h_counts ? h_counts + i : nullptr;
                if (q[i]->n == 0) {
                    HIP_TRY(ctx, hipMemsetAsync(d_counts + i, 0, 8, ctx->stream_tail));
                    if (hc) *hc = 0;
                } else {
                    rc = xcheck_common(ctx, q[i], t[i], true, tau, nullptr, nullptr, nullptr, nullptr, hc, "fm_match_accepted_dev_batch",
                                       cap, nullptr, d_rows + (size_t)i * cap * 3, (long long*)(d_counts + i), !in_place, kNoStream);
                    if (rc != FM_OK) return rc;
                }
            } else {
                if (q[i]->n == 0) *n_accepted[i] = 0;
                rc = xcheck_common(ctx, q[i], t[i], true, tau, tidx[i], dist[i], ratio[i], nullptr, n_accepted[i],
                                   "fm_match_accepted_batch", cap, qidx[i], nullptr, nullptr, !in_place);
                if (rc != FM_OK) return rc;
            }
            ++i;
            continue;
        }
        const bool coop = (ctx->tune.coop != 0) && pl.nsplit > 1;
        void* al[kRRBatchMax][5];
        int slot_of[kRRBatchMax];
        SlotLayout L[kRRBatchMax];
        const Bank* cols[kRRBatchMax]; const Bank* red[kRRBatchMax];
        unsigned long long* part[kRRBatchMax]; int* bnd[kRRBatchMax];
        fm_ctx::PendingTimer tm;
        if ((rc = take_timer(ctx, &tm)) != FM_OK) return rc;
        tm.timed = true; tm.call_timed = true; tm.pairs = 0;
        for (int j = 0; j < g; ++j) {
            const int k = i + j;
            if (to_dev) {
                for (int u = 0; u < 4; ++u) al[j][u] = nullptr;
                al[j][4] = h_counts ? pinned_device_alias(h_counts + k) : nullptr;
                if (h_counts) h_counts[k] = 0;
            } else {
                al[j][0] = pinned_device_alias(qidx[k]); al[j][1] = pinned_device_alias(tidx[k]);
                al[j][2] = pinned_device_alias(dist[k]); al[j][3] = pinned_device_alias(ratio[k]);
                al[j][4] = pinned_device_alias(n_accepted[k]);
                for (int u = 0; u < 5; ++u)
                    if (!al[j][u]) { ctx->timer_pool.push_back(tm); return fail(ctx, FM_EINVAL, "fm_match_accepted_batch: host outputs must be page-locked (fm_host_alloc)"); }
                *n_accepted[k] = 0;
            }
            slot_of[j] = (int)(ctx->bslot_next++ % fm_ctx::kBatchSlots);
            fm_ctx::AsyncSlot& sl = ctx->bslot[(size_t)slot_of[j]];
            if (!sl.tail_done) HIP_TRY(ctx, hipEventCreateWithFlags(&sl.tail_done, hipEventDisableTiming));
            L[j] = slot_layout(q[k]->n, t[k]->n, pl);
            if ((rc = slot_prepare(ctx, sl, L[j], q[k]->n, pl, ctx->stream)) != FM_OK) { ctx->timer_pool.push_back(tm); return rc; }
            cols[j] = t[k]; red[j] = q[k];            // reverse NN: output rows = train rows, reduced over the query rows
            part[j] = (unsigned long long*)sl.ws;
            bnd[j] = coop ? (int*)((char*)sl.ws + L[j].pbytes) : nullptr;
            tm.pairs += q[k]->n * t[k]->n;
        }
        HIP_TRY(ctx, hipEventRecord(tm.k0, ctx->stream));
        HIP_TRY(ctx, launch_rowreduce_batch(g, cols, red, pl, part, bnd, ctx->stream));
        HIP_TRY(ctx, hipEventRecord(tm.k1, ctx->stream));
        for (int j = 0; j < g; ++j) {
            const int k = i + j;
            hipStream_t ts = ctx->tails[j % fm_ctx::kTails];
            fm_ctx::AsyncSlot& sl = ctx->bslot[(size_t)slot_of[j]];
            HIP_TRY(ctx, hipStreamWaitEvent(ts, tm.k1, 0));
            if ((rc = enqueue_tail(ctx, ts, sl, L[j], q[k], t[k], q[k]->n, t[k]->n, pl, tau, cap, al[j][0], al[j][1], al[j][2], al[j][3],
                                   al[j][4], to_dev ? d_rows + (size_t)k * cap * 3 : nullptr, (long long*)(to_dev ? d_counts + k : nullptr),
                                   kNoStream)) != FM_OK) { ctx->pending.push_back(tm); return rc; }     // (events are in flight: drained at fm_sync)
            HIP_TRY(ctx, hipEventRecord(sl.tail_done, ts));
            sl.in_use = true;
            if (j == g - 1) HIP_TRY(ctx, hipEventRecord(tm.c1, ts));
        }
        ctx->pending.push_back(tm);
        i += g;
    }
    if (to_dev) {
        // everything the tails were given is in front of these records: the consumer waits for all of it
        // (without a consumer stream the first tail stream collects the others: fm_gather_matches follows it)
        for (int u = 0; u < fm_ctx::kTails; ++u) {
            hipStream_t ts = ctx->tails[u];
            HIP_TRY(ctx, hipEventRecord(ctx->ev_tail_end[u], ts));
            if (consumer != kNoStream) HIP_TRY(ctx, hipStreamWaitEvent(consumer, ctx->ev_tail_end[u], 0));
            else if (ts != ctx->stream_tail) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream_tail, ctx->ev_tail_end[u], 0));
        }
        ctx->rows_stream = ctx->stream_tail;
    }
    return FM_OK;
}

extern "C" int fm_match_accepted_dev(fm_ctx* ctx, const fm_bank* q, const fm_bank* t, double tau, int64_t cap,
                                     int32_t* d_rows, int64_t* d_count, int64_t* n_accepted)
{
    if (!ctx) return fail(nullptr, FM_EINVAL, "fm_match_accepted_dev: ctx is NULL");
    if (cap < 0) return fail(ctx, FM_EINVAL, "fm_match_accepted_dev: cap < 0");
    if (!d_rows || !d_count) return fail(ctx, FM_EINVAL, "fm_match_accepted_dev: device output pointer is NULL");
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, d_rows) != hipSuccess || at.type != hipMemoryTypeDevice ||
        hipPointerGetAttributes(&at, d_count) != hipSuccess || at.type != hipMemoryTypeDevice) {
        (void)hipGetLastError();
        return fail(ctx, FM_EINVAL, "fm_match_accepted_dev: d_rows / d_count must be device memory");
    }
    if (q && q->n == 0) HIP_TRY(ctx, hipMemset(d_count, 0, 8));
    return xcheck_common(ctx, q, t, true, tau, nullptr, nullptr, nullptr, nullptr, n_accepted, "fm_match_accepted_dev",
                         cap, nullptr, d_rows, (long long*)d_count);
}

extern "C" int fm_match_accepted_dev_async(fm_ctx* ctx, const fm_bank* q, const fm_bank* t, double tau, int64_t cap,
                                           int32_t* d_rows, int64_t* d_count, int64_t* h_count, void* consumer_stream)
{
    if (!ctx) return fail(nullptr, FM_EINVAL, "fm_match_accepted_dev_async: ctx is NULL");
    if (cap < 0) return fail(ctx, FM_EINVAL, "fm_match_accepted_dev_async: cap < 0");
    if (!d_rows || !d_count) return fail(ctx, FM_EINVAL, "fm_match_accepted_dev_async: device output pointer is NULL");
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, d_rows) != hipSuccess || at.type != hipMemoryTypeDevice ||
        hipPointerGetAttributes(&at, d_count) != hipSuccess || at.type != hipMemoryTypeDevice) {
        (void)hipGetLastError();
        return fail(ctx, FM_EINVAL, "fm_match_accepted_dev_async: d_rows / d_count must be device memory");
    }
    if (q && q->n == 0) {
        HIP_TRY(ctx, hipMemsetAsync(d_count, 0, 8, ctx->stream_tail));
        if (h_count) *h_count = 0;
        if (consumer_stream != FM_NO_STREAM) {
            HIP_TRY(ctx, hipEventRecord(ctx->ev_consumer, ctx->stream_tail));
            HIP_TRY(ctx, hipStreamWaitEvent((hipStream_t)consumer_stream, ctx->ev_consumer, 0));
        }
        ctx->rows_stream = ctx->stream_tail;
    }
    return xcheck_common(ctx, q, t, true, tau, nullptr, nullptr, nullptr, nullptr, h_count, "fm_match_accepted_dev_async",
                         cap, nullptr, d_rows, (long long*)d_count, true, (hipStream_t)consumer_stream);
}

extern "C" int fm_ratio_filter(fm_ctx* ctx, const float* dist, const double* selfdist, const int32_t* qrows,
                               int64_t n, double tau, double* ratio, uint8_t* pass, int64_t* n_pass)
{
    if (!ctx) return fail(nullptr, FM_EINVAL, "fm_ratio_filter: ctx is NULL");
    if (n_pass) *n_pass = 0;
    if (n < 0) return fail(ctx, FM_EINVAL, "fm_ratio_filter: n < 0");
    if (n == 0) return FM_OK;
    if (!dist || !selfdist) return fail(ctx, FM_EINVAL, "fm_ratio_filter: NULL input");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // selfdist is indexed by qrows (or by i): upload max(qrows)+1 entries
    int64_t nsd = n;
    if (qrows) { nsd = 0; for (int64_t i = 0; i < n; ++i) { if (qrows[i] < 0) return fail(ctx, FM_EINVAL, "fm_ratio_filter: negative qrow"); if (qrows[i] + 1 > nsd) nsd = qrows[i] + 1; } }
    const size_t i_dist = 0, i_sd = ((size_t)n * 4 + 7) & ~(size_t)7, i_qr = i_sd + (size_t)nsd * 8;
    const size_t in_bytes = i_qr + (qrows ? (size_t)n * 4 : 0);
    int rc;
    if ((rc = ws_ensure(ctx, &ctx->ws_in, &ctx->ws_in_bytes, in_bytes + 16)) != FM_OK) return rc;
    const size_t o_ratio = 0, o_pass = (size_t)n * 8, o_cnt = (o_pass + (size_t)n + 15) & ~(size_t)15;
    if ((rc = ws_ensure(ctx, &ctx->ws_out, &ctx->ws_out_bytes, o_cnt + 16)) != FM_OK) return rc;
    char* ib = (char*)ctx->ws_in;
    char* ob = (char*)ctx->ws_out;
    CallScope cs(ctx);
    HIP_TRY(ctx, hipMemcpyAsync(ib + i_dist, dist, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ib + i_sd, selfdist, (size_t)nsd * 8, hipMemcpyHostToDevice, ctx->stream));
    if (qrows) HIP_TRY(ctx, hipMemcpyAsync(ib + i_qr, qrows, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ob + o_cnt, 0, 8, ctx->stream));
    hipLaunchKernelGGL(ratio_filter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const float*)(ib + i_dist), (const double*)(ib + i_sd),
                       qrows ? (const int32_t*)(ib + i_qr) : (const int32_t*)nullptr, n, tau,
                       (double*)(ob + o_ratio), (uint8_t*)(ob + o_pass), (unsigned long long*)(ob + o_cnt));
    HIP_TRY(ctx, hipGetLastError());
    unsigned long long cnt = 0;
    if (ratio) HIP_TRY(ctx, d2h(ctx, ratio, ob + o_ratio, (size_t)n * 8));
    if (pass) HIP_TRY(ctx, d2h(ctx, pass, ob + o_pass, (size_t)n));
    HIP_TRY(ctx, hipMemcpyAsync(&cnt, ob + o_cnt, 8, hipMemcpyDeviceToHost, ctx->stream));
    rc = cs.finish();
    if (rc != FM_OK) return rc;
    if (n_pass) *n_pass = (int64_t)cnt;
    return FM_OK;
}

// Planes and scale terms of a (query = reduced, train = output rows) pair of float32 banks for
// x1_round_f32: the same margin / accumulator-init factors launch_filter (filter_f16.hip) uses.
static void fill_round_f32(RoundF32* r, const Bank& q, const Bank& t)
{
    const float eps = 1.1f / 1024.0f;
    const int dk = t.kscale - q.kscale;               // acc units are 2^(kt + kq)
    r->q_rowsh = (const char*)q.rowsh; r->q_auxf = q.auxf; r->q_rowsf = q.rowsf;
    r->t_rowsh = (const char*)t.rowsh; r->t_normf = t.normf; r->t_rowsf = t.rowsf;
    r->eps_c = ldexpf(eps, -dk);
    r->eps_nm = ldexpf(eps * q.nm_max, dk);
    r->aux_mul = ldexpf(1.0f, dk);
}

extern "C" int fm_xcheck1_batched(fm_ctx* ctx, const fm_bank* q, const int32_t* q_rows, const int64_t* q_off,
                                  const fm_bank* t, const int64_t* t_off, int64_t n_rounds,
                                  int32_t* tidx, float* dist, double* ratio)
{
    int rc = check_pair(ctx, q, t, "fm_xcheck1_batched");
    if (rc != FM_OK) return rc;
    const bool f32 = q->kind == FM_BANK_F32;
    if (f32 && !filter_usable(*t, *q))
        return fail(ctx, FM_EUNSUPPORTED, "fm_xcheck1_batched: float32 banks without usable fp16 filter planes take the dense path (fm_xcheck1 / fm_match_ratio)");
    if (n_rounds < 0) return fail(ctx, FM_EINVAL, "fm_xcheck1_batched: n_rounds < 0");
    if (n_rounds == 0) return FM_OK;
    if (!q_off || !t_off) return fail(ctx, FM_EINVAL, "fm_xcheck1_batched: NULL offsets");
    if (q_off[0] != 0) return fail(ctx, FM_EINVAL, "fm_xcheck1_batched: q_off[0] must be 0");
    int64_t pairs = 0;
    for (int64_t b = 0; b < n_rounds; ++b) {
        const int64_t nq = q_off[b + 1] - q_off[b], nt = t_off[b + 1] - t_off[b];
        if (nq < 0 || nt < 0 || t_off[b] < 0 || t_off[b + 1] > t->n)
            return fail(ctx, FM_EINVAL, "fm_xcheck1_batched: bad round offsets");
        if (nq > round_qcap())
            return fail(ctx, FM_EUNSUPPORTED, "fm_xcheck1_batched: a round has more than 4096 query rows; use fm_xcheck1 on gathered banks");
        pairs += nq * nt;
    }
    const int64_t tot = q_off[n_rounds];
    if (tot == 0) return FM_OK;
    if (!q_rows || !tidx || !dist) return fail(ctx, FM_EINVAL, "fm_xcheck1_batched: NULL rows/outputs");
    for (int64_t i = 0; i < tot; ++i)
        if (q_rows[i] < 0 || q_rows[i] >= q->n) return fail(ctx, FM_EINVAL, "fm_xcheck1_batched: q_rows index out of range");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t i_qoff = 0, i_toff = (size_t)(n_rounds + 1) * 8, i_rows = i_toff + (size_t)(n_rounds + 1) * 8;
    if ((rc = ws_ensure(ctx, &ctx->ws_in, &ctx->ws_in_bytes, i_rows + (size_t)tot * 4 + 16)) != FM_OK) return rc;
    const size_t o_tidx = 0, o_dist = (size_t)tot * 4, o_ratio = ((size_t)tot * 8 + 7) & ~(size_t)7;
    if ((rc = ws_ensure(ctx, &ctx->ws_out, &ctx->ws_out_bytes, o_ratio + (size_t)tot * 8 + 16)) != FM_OK) return rc;
    char* ib = (char*)ctx->ws_in;
    char* ob = (char*)ctx->ws_out;
    CallScope cs(ctx);
    HIP_TRY(ctx, hipMemcpyAsync(ib + i_qoff, q_off, (size_t)(n_rounds + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ib + i_toff, t_off, (size_t)(n_rounds + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ib + i_rows, q_rows, (size_t)tot * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_k0, ctx->stream));
    if (f32) {
        RoundF32 rf;
        fill_round_f32(&rf, *q, *t);
        HIP_TRY(ctx, launch_rounds_f32(rf, q->selfdist, (const int32_t*)(ib + i_rows), (const int64_t*)(ib + i_qoff),
                                       (const int64_t*)(ib + i_toff), n_rounds, (int32_t*)(ob + o_tidx),
                                       (float*)(ob + o_dist), (double*)(ob + o_ratio), ctx->stream));
    } else {
        HIP_TRY(ctx, launch_rounds(*q, *t, (const int32_t*)(ib + i_rows), (const int64_t*)(ib + i_qoff),
                                   (const int64_t*)(ib + i_toff), n_rounds, (int32_t*)(ob + o_tidx),
                                   (float*)(ob + o_dist), (double*)(ob + o_ratio), ctx->stream));
    }
    HIP_TRY(ctx, hipEventRecord(ctx->ev_k1, ctx->stream));
    ctx->kernel_timed = true;
    ctx->pending_pairs += pairs;
    HIP_TRY(ctx, d2h(ctx, tidx, ob + o_tidx, (size_t)tot * 4));
    HIP_TRY(ctx, d2h(ctx, dist, ob + o_dist, (size_t)tot * 4));
    if (ratio) HIP_TRY(ctx, d2h(ctx, ratio, ob + o_ratio, (size_t)tot * 8));
    return cs.finish();
}

// ---------------------------------------------------------------------------------------
// K7 entry points
// ---------------------------------------------------------------------------------------
// An fm_expand holds what an image pair's runs SHARE and never change (banks, position index, cell
// offsets, target positions) plus a pool of run states (pending stack, seen / found tables, result
// arrays, seed buffer).  One launch may hold several runs of one pair -- the reference is driven as
// pairs x thresholds (turntable.py:59-60) -- each run in a state of its own, one workgroup each.
struct ExpandRun {
    void* blob = nullptr;          // stack | seen | found | m_index | m_pos | m_ratio | result
    double* stack = nullptr;
    unsigned long long* seen = nullptr;
    unsigned long long* found = nullptr;
    int32_t* m_index = nullptr;
    double* m_pos = nullptr;
    double* m_ratio = nullptr;
    long long* result = nullptr;
    double* d_seeds = nullptr;     // grown on demand
    int64_t seeds_cap = 0;
    // capacities of THIS run state: they start at the pair's defaults and are multiplied by four when a run
    // ends with the matching FM_EXPAND_*_FULL status (fm_expand_run then repeats the run)
    int64_t match_cap = 0, stack_cap = 0, seen_cap = 0, found_cap = 0;
};

struct fm_expand {
    ExpandPair dev{};              // device pointers + parameters (run state, seeds and tau filled per run)
    void* blob = nullptr;          // the shared arrays
    int64_t nq = 0;
    int64_t match_cap = 0, stack_cap = 0, seen_cap = 0;       // defaults of a new run state
    std::vector<ExpandRun> runs;   // run slot k = the k-th run of this pair inside one launch
};

static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
static inline int64_t pow2_at_least(int64_t x) { int64_t p = 1; while (p < x) p <<= 1; return p; }

static void expand_run_free(ExpandRun& r)
{
    if (r.blob) (void)hipFree(r.blob);
    if (r.d_seeds) (void)hipFree(r.d_seeds);
    r = ExpandRun{};
}

// (Re)allocate the arrays of a run state for its current capacities.
static int expand_run_alloc(fm_ctx* ctx, ExpandRun& r)
{
    if (r.blob) { (void)hipFree(r.blob); r.blob = nullptr; }
    r.found_cap = pow2_at_least(4 * r.match_cap);
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += al256(bytes > 0 ? bytes : 1); return o; };
    // (seen and found are neighbours: one fill resets both)
    const size_t o_stack = carve((size_t)r.stack_cap * 32), o_seen = carve((size_t)r.seen_cap * 8), o_found = carve((size_t)r.found_cap * 16);
    const size_t o_mi = carve((size_t)r.match_cap * 4), o_mp = carve((size_t)r.match_cap * 32), o_mr = carve((size_t)r.match_cap * 8);
    const size_t o_res = carve(256);
    hipError_t e = hipMalloc(&r.blob, off);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        r.blob = nullptr;
        return fail(ctx, FM_ENOMEM, std::string("fm_expand: run state (") + std::to_string(off >> 20) + " MiB): " + hipGetErrorString(e));
    }
    char* b = (char*)r.blob;
    r.stack = (double*)(b + o_stack); r.seen = (unsigned long long*)(b + o_seen); r.found = (unsigned long long*)(b + o_found);
    r.m_index = (int32_t*)(b + o_mi); r.m_pos = (double*)(b + o_mp); r.m_ratio = (double*)(b + o_mr);
    r.result = (long long*)(b + o_res);
    return FM_OK;
}

// Run slot `slot` of the pair exists after this call (slots are created in order).
static int expand_ensure_run(fm_ctx* ctx, fm_expand* ex, size_t slot)
{
    while (ex->runs.size() <= slot) {
        ExpandRun r;
        r.match_cap = ex->match_cap; r.stack_cap = ex->stack_cap; r.seen_cap = ex->seen_cap;
        int rc = expand_run_alloc(ctx, r);
        if (rc != FM_OK) return rc;
        ex->runs.push_back(r);
    }
    return FM_OK;
}

static void expand_bind_run(ExpandPair& P, const ExpandRun& r)
{
    P.stack = r.stack; P.stack_cap = r.stack_cap;
    P.seen = r.seen; P.seen_cap = r.seen_cap;
    P.found = r.found; P.found_cap = r.found_cap;
    P.m_index = r.m_index; P.m_pos = r.m_pos; P.m_ratio = r.m_ratio; P.match_cap = r.match_cap;
    P.result = r.result;
}

extern "C" int fm_expand_create(fm_ctx* ctx, const fm_expand_desc* d, fm_expand** out)
{
    if (!ctx) return fail(nullptr, FM_EINVAL, "fm_expand_create: ctx is NULL");
    if (!d || !out) return fail(ctx, FM_EINVAL, "fm_expand_create: NULL argument");
    *out = nullptr;
    if (!d->query || !d->target) return fail(ctx, FM_EINVAL, "fm_expand_create: NULL bank");
    if (d->query->kind != d->target->kind)
        return fail(ctx, FM_EINVAL, "fm_expand_create: query/target kind mismatch");
    const bool f32 = d->query->kind == FM_BANK_F32;
    if (f32 && !filter_usable(*d->target, *d->query))
        return fail(ctx, FM_EUNSUPPORTED, "fm_expand_create: float32 banks without usable fp16 filter planes (non-finite values or scales more than 2^40 apart)");
    if (d->query->dim != d->target->dim) return fail(ctx, FM_EINVAL, "fm_expand_create: dim mismatch");
    if (!d->query->selfdist) return fail(ctx, FM_EINVAL, "fm_expand_create: query bank has no self distances");
    const int64_t nq = d->query->n, nt = d->target->n;
    const int64_t ncells = (int64_t)d->rows * d->cols;
    if (d->width < 1 || d->height < 1 || d->cell_w < 1 || d->cell_h < 1 || d->rows < 1 || d->cols < 1 || d->radius < 0)
        return fail(ctx, FM_EINVAL, "fm_expand_create: bad grid parameters");
    if (d->metric < FM_METRIC_EUCLIDEAN || d->metric > FM_METRIC_CHEBYSHEV) return fail(ctx, FM_EINVAL, "fm_expand_create: unknown metric");
    if (d->rows > 65535 || d->cols > 65535 || d->width > 65535 || d->height > 65535)
        return fail(ctx, FM_EUNSUPPORTED, "fm_expand_create: image or grid too large for 16-bit cell keys");
    if ((nq > 0 && (!d->query_pos || !d->index_order)) || !d->index_start || !d->cell_off || (nt > 0 && !d->target_pos))
        return fail(ctx, FM_EINVAL, "fm_expand_create: NULL array");
    const int64_t nb = (int64_t)d->index_nbx * d->index_nby;
    if (d->index_nbx < 0 || d->index_nby < 0 || d->index_start[nb] != nq || !(d->index_bucket > 0.0))
        return fail(ctx, FM_EINVAL, "fm_expand_create: inconsistent position index");
    if (d->cell_off[0] != 0 || d->cell_off[ncells] != nt) return fail(ctx, FM_EINVAL, "fm_expand_create: cell_off must cover the target bank");
    for (int64_t c = 0; c < ncells; ++c)
        if (d->cell_off[c + 1] < d->cell_off[c]) return fail(ctx, FM_EINVAL, "fm_expand_create: cell_off not monotonic");
    for (int64_t i = 0; i < nq; ++i) {
        if (d->index_order[i] < 0 || d->index_order[i] >= nq) return fail(ctx, FM_EINVAL, "fm_expand_create: index_order out of range");
        const double x = d->query_pos[2 * i], y = d->query_pos[2 * i + 1];
        if (!(x >= 0.0) || !(y >= 0.0) || x / d->cell_w >= 65535.0 || y / d->cell_h >= 65535.0)
            return fail(ctx, FM_EUNSUPPORTED, "fm_expand_create: query position outside the 16-bit cell-key range");
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    fm_expand* ex = new (std::nothrow) fm_expand();
    if (!ex) return fail(ctx, FM_ENOMEM, "fm_expand_create: out of host memory");
    ex->nq = nq;
    ex->match_cap = d->match_cap > 0 ? d->match_cap : (4 * nq > 1024 ? 4 * nq : 1024);
    ex->stack_cap = d->stack_cap > 0 ? d->stack_cap : (64 * ncells > 65536 ? 64 * ncells : 65536);
    ex->seen_cap = pow2_at_least(16 * ncells > 65536 ? 16 * ncells : 65536);
    // the shared arrays in one allocation
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += al256(bytes > 0 ? bytes : 1); return o; };
    const size_t o_qpos = carve((size_t)nq * 16), o_order = carve((size_t)nq * 4), o_start = carve((size_t)(nb + 1) * 4);
    const size_t o_coff = carve((size_t)(ncells + 1) * 8), o_tpos = carve((size_t)nt * 16);
    hipError_t e = hipMalloc(&ex->blob, off);
    if (e != hipSuccess) { (void)hipGetLastError(); delete ex; return fail(ctx, FM_ENOMEM, std::string("fm_expand_create: hipMalloc: ") + hipGetErrorString(e)); }
    char* b = (char*)ex->blob;
    auto bail = [&](int code) { for (auto& r : ex->runs) expand_run_free(r); (void)hipFree(ex->blob); delete ex; return code; };
#define ETRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { (void)hipGetLastError(); \
        return bail(fail(ctx, FM_EDEVICE, std::string(#expr " failed: ") + hipGetErrorString(_e))); } } while (0)
    if (nq) ETRY(hipMemcpyAsync(b + o_qpos, d->query_pos, (size_t)nq * 16, hipMemcpyHostToDevice, ctx->stream));
    if (nq) ETRY(hipMemcpyAsync(b + o_order, d->index_order, (size_t)nq * 4, hipMemcpyHostToDevice, ctx->stream));
    ETRY(hipMemcpyAsync(b + o_start, d->index_start, (size_t)(nb + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    ETRY(hipMemcpyAsync(b + o_coff, d->cell_off, (size_t)(ncells + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    if (nt) ETRY(hipMemcpyAsync(b + o_tpos, d->target_pos, (size_t)nt * 16, hipMemcpyHostToDevice, ctx->stream));
    ETRY(hipStreamSynchronize(ctx->stream));
#undef ETRY
    ExpandPair& P = ex->dev;
    P.q_rows8 = d->query->rows8; P.q_norm = d->query->norm; P.q_selfdist = d->query->selfdist;
    P.q_pos = (const double*)(b + o_qpos);
    P.idx_order = (const int32_t*)(b + o_order); P.idx_start = (const int32_t*)(b + o_start);
    P.idx_bucket = d->index_bucket; P.idx_x0 = d->index_x0; P.idx_y0 = d->index_y0;
    P.idx_nbx = d->index_nbx; P.idx_nby = d->index_nby;
    P.metric = d->metric;
    P.t_rows8 = d->target->rows8; P.t_norm = d->target->norm;
    P.f32 = f32 ? 1 : 0;
    P.tie_guard = (!f32 && sqrt_tie_possible(*d->query, *d->target)) ? 1 : 0;
    P.rf = RoundF32{};
    if (f32) fill_round_f32(&P.rf, *d->query, *d->target);
    P.cell_off = (const int64_t*)(b + o_coff); P.t_pos = (const double*)(b + o_tpos);
    P.width = d->width; P.height = d->height; P.cell_w = d->cell_w; P.cell_h = d->cell_h;
    P.rows = d->rows; P.cols = d->cols; P.margin = d->margin; P.radius = d->radius;
    P.seeds = nullptr; P.n_seeds = 0; P.tau = 0.0;
    P.prof = 0;                        // (set per run from the context's expand_prof option)
    // the first run state exists from the start (a pair that cannot get one fails here, not at its first run)
    int rc = expand_ensure_run(ctx, ex, 0);
    if (rc != FM_OK) return bail(rc);
    expand_bind_run(P, ex->runs[0]);
    *out = ex;
    return FM_OK;
}

extern "C" int fm_expand_destroy(fm_ctx* ctx, fm_expand* ex)
{
    if (!ex) return FM_OK;
    if (ctx) { (void)hipSetDevice(ctx->device); (void)hipStreamSynchronize(ctx->stream); }
    for (auto& r : ex->runs) expand_run_free(r);
    if (ex->blob) (void)hipFree(ex->blob);
    delete ex;
    return FM_OK;
}

// Run slot of entry i of a launch: how many earlier entries name the same pair.
static int expand_slot_of(fm_expand* const* pairs, int i)
{
    int k = 0;
    for (int j = 0; j < i; ++j) k += pairs[j] == pairs[i] ? 1 : 0;
    return k;
}

extern "C" int fm_expand_run(fm_ctx* ctx, int32_t n, fm_expand* const* pairs, const double* const* seeds,
                             const int64_t* n_seeds, const double* tau, int64_t* n_matches,
                             int64_t* n_rounds, int64_t* n_pairs, int32_t* status)
{
    if (!ctx) return fail(nullptr, FM_EINVAL, "fm_expand_run: ctx is NULL");
    if (n < 0) return fail(ctx, FM_EINVAL, "fm_expand_run: n < 0");
    if (n == 0) return FM_OK;
    if (!pairs || !seeds || !n_seeds || !tau) return fail(ctx, FM_EINVAL, "fm_expand_run: NULL argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::vector<ExpandPair> host((size_t)n);
    std::vector<ExpandRun*> run((size_t)n);
    int rc;
    {
        // slots by occurrence (a map instead of expand_slot_of's quadratic scan: a launch may hold thousands of runs)
        std::map<fm_expand*, int> seen_pairs;
        for (int i = 0; i < n; ++i) {
            fm_expand* ex = pairs[i];
            if (!ex) return fail(ctx, FM_EINVAL, "fm_expand_run: NULL pair");
            if (n_seeds[i] < 0 || (n_seeds[i] > 0 && !seeds[i])) return fail(ctx, FM_EINVAL, "fm_expand_run: bad seeds");
            const int slot = seen_pairs[ex]++;
            if ((rc = expand_ensure_run(ctx, ex, (size_t)slot)) != FM_OK) return rc;
        }
        seen_pairs.clear();
        for (int i = 0; i < n; ++i) {                      // (pointers into runs[] are taken once the vectors stopped growing)
            fm_expand* ex = pairs[i];
            ExpandRun* r = &ex->runs[(size_t)seen_pairs[ex]++];
            run[(size_t)i] = r;
            if (n_seeds[i] > r->seeds_cap) {
                if (r->d_seeds) { HIP_TRY(ctx, hipFree(r->d_seeds)); r->d_seeds = nullptr; r->seeds_cap = 0; }
                const int64_t cap = n_seeds[i] + n_seeds[i] / 2 + 64;
                HIP_TRY(ctx, hipMalloc((void**)&r->d_seeds, (size_t)cap * 32));
                r->seeds_cap = cap;
            }
        }
    }
    rc = ws_ensure(ctx, &ctx->ws_in, &ctx->ws_in_bytes, (size_t)n * sizeof(ExpandPair) + 64);
    if (rc != FM_OK) return rc;
    CallScope cs(ctx);
    for (int i = 0; i < n; ++i) {
        fm_expand* ex = pairs[i];
        ExpandRun* r = run[(size_t)i];
        if (n_seeds[i]) HIP_TRY(ctx, hipMemcpyAsync(r->d_seeds, seeds[i], (size_t)n_seeds[i] * 32, hipMemcpyHostToDevice, ctx->stream));
        // (the two tables are neighbours in the run's allocation: one fill)
        HIP_TRY(ctx, hipMemsetAsync(r->seen, 0xff, (size_t)((char*)r->found - (char*)r->seen) + (size_t)r->found_cap * 16, ctx->stream));
        host[i] = ex->dev;
        expand_bind_run(host[i], *r);
        host[i].seeds = r->d_seeds;
        host[i].n_seeds = n_seeds[i];
        host[i].tau = tau[i];
        host[i].prof = ctx->tune.expand_prof;
    }
    // the int8 and the float32 pairs are two kernels: descriptors grouped by kind, one launch each
    std::vector<ExpandPair> grouped;
    grouped.reserve((size_t)n);
    for (int i = 0; i < n; ++i) if (!host[i].f32) grouped.push_back(host[i]);
    const int n_i8 = (int)grouped.size();
    for (int i = 0; i < n; ++i) if (host[i].f32) grouped.push_back(host[i]);
    HIP_TRY(ctx, hipMemcpyAsync(ctx->ws_in, grouped.data(), (size_t)n * sizeof(ExpandPair), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_k0, ctx->stream));
    if (n_i8 > 0) HIP_TRY(ctx, launch_expand(ctx->ws_in, n_i8, false, false, ctx->stream));
    if (n - n_i8 > 0) HIP_TRY(ctx, launch_expand((const char*)ctx->ws_in + (size_t)n_i8 * sizeof(ExpandPair), n - n_i8, true, false, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_k1, ctx->stream));
    ctx->kernel_timed = true;
    std::vector<long long> res((size_t)n * 4);
    for (int i = 0; i < n; ++i)
        HIP_TRY(ctx, hipMemcpyAsync(&res[(size_t)i * 4], run[(size_t)i]->result, 32, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    // Runs that ended on a capacity run again, from the start: a radius subset beyond the kernel's 2048 rows
    // (status 2, int8 banks) in the 4096-row variant of the kernel; a full pending stack, result list or
    // hash table (status 1, 4, 5: thresholds above 1 accept nearly every cross-checked pair and the
    // expansion heads for every (cell, query cell) combination) in a run state four times as large, at most
    // `expand_grow` times over (default 2; the status stands after that).  The other runs keep their results.
    std::vector<char> big((size_t)n, 0);
    for (int pass = 0; pass <= ctx->tune.expand_grow + 1; ++pass) {
        std::vector<int> redo;
        for (int i = 0; i < n; ++i) {
            const long long st = res[(size_t)i * 4 + 3];
            ExpandRun* r = run[(size_t)i];
            if (st == 2 && !host[i].f32 && ctx->tune.expand_big && !big[(size_t)i]) { big[(size_t)i] = 1; redo.push_back(i); continue; }
            if ((st == 1 || st == 4 || st == 5) && pass < ctx->tune.expand_grow + (big[(size_t)i] ? 1 : 0)) {
                const int64_t limit = (int64_t)1 << 28;
                if (st == 1) { if (r->stack_cap >= limit) continue; r->stack_cap *= 4; }
                if (st == 4) { if (r->match_cap >= limit) continue; r->match_cap *= 4; }
                if (st == 5) { if (r->seen_cap >= limit) continue; r->seen_cap *= 4; }
                if (expand_run_alloc(ctx, *r) != FM_OK) {                 // no memory for the larger state: the status stands
                    r->match_cap = pairs[i]->match_cap; r->stack_cap = pairs[i]->stack_cap; r->seen_cap = pairs[i]->seen_cap;
                    if ((rc = expand_run_alloc(ctx, *r)) != FM_OK) return rc;
                    continue;
                }
                expand_bind_run(host[i], *r);
                redo.push_back(i);
            }
        }
        if (redo.empty()) break;
        for (int v = 0; v < 2; ++v) {                   // the two int8 capacity variants (float32 pairs: the small one)
            grouped.clear();
            std::vector<int> idx;
            for (int i : redo) if ((big[(size_t)i] ? 1 : 0) == v && !host[i].f32) { grouped.push_back(host[i]); idx.push_back(i); }
            const int n8 = (int)idx.size();
            if (v == 0) for (int i : redo) if (host[i].f32) { grouped.push_back(host[i]); idx.push_back(i); }
            if (idx.empty()) continue;
            for (int i : idx)
                HIP_TRY(ctx, hipMemsetAsync(run[(size_t)i]->seen, 0xff, (size_t)((char*)run[(size_t)i]->found - (char*)run[(size_t)i]->seen) +
                                            (size_t)run[(size_t)i]->found_cap * 16, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(ctx->ws_in, grouped.data(), grouped.size() * sizeof(ExpandPair), hipMemcpyHostToDevice, ctx->stream));
            if (n8 > 0) HIP_TRY(ctx, launch_expand(ctx->ws_in, n8, false, v == 1, ctx->stream));
            if ((int)idx.size() > n8)
                HIP_TRY(ctx, launch_expand((const char*)ctx->ws_in + (size_t)n8 * sizeof(ExpandPair), (int)idx.size() - n8, true, false, ctx->stream));
            HIP_TRY(ctx, hipEventRecord(ctx->ev_k1, ctx->stream));
            for (int i : idx)
                HIP_TRY(ctx, hipMemcpyAsync(&res[(size_t)i * 4], run[(size_t)i]->result, 32, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));     // (ws_in is reused by the next variant)
        }
    }
    for (int i = 0; i < n; ++i) ctx->pending_pairs += res[(size_t)i * 4 + 2];
    if (ctx->tune.expand_prof) {
        long long pr[16];
        (void)hipMemcpy(pr, run[0]->result, sizeof(pr), hipMemcpyDeviceToHost);
        static const char* names[12] = {"pop:barrier", "radius", "sort", "x1_tail", "compact", "neigh+push+emit", "end", "pop:thread0",
                                        "x1:bfrag+barrier", "x1:gather", "x1:mfma", "x1:merge"};
        fprintf(stderr, "[fm_expand prof, run 0, %lld rounds] ", pr[1]);
        for (int k = 0; k < 12; ++k) fprintf(stderr, "%s %.2f us  ", names[k], pr[1] ? pr[4 + k] * 0.01 / (double)pr[1] : 0.0);
        fprintf(stderr, "\n");
    }
    rc = cs.finish();
    if (rc != FM_OK) return rc;
    for (int i = 0; i < n; ++i) {
        if (n_matches) n_matches[i] = res[(size_t)i * 4 + 0];
        if (n_rounds) n_rounds[i] = res[(size_t)i * 4 + 1];
        if (n_pairs) n_pairs[i] = res[(size_t)i * 4 + 2];
        if (status) status[i] = (int32_t)res[(size_t)i * 4 + 3];
    }
    return FM_OK;
}

extern "C" int fm_expand_fetch(fm_ctx* ctx, const fm_expand* ex, int64_t n, int32_t* index, double* positions, double* ratio)
{
    if (!ctx || !ex) return fail(ctx, FM_EINVAL, "fm_expand_fetch: NULL argument");
    const int32_t slot = 0;
    const fm_expand* one = ex;
    int32_t* ip = index; double* pp = positions; double* rp = ratio;
    return fm_expand_fetch_many(ctx, 1, &one, &slot, &n, index ? &ip : nullptr, positions ? &pp : nullptr, ratio ? &rp : nullptr);
}

// The results of several runs of one fm_expand_run in ONE pass: every copy is enqueued, then a single
// synchronisation (a synchronisation per run: 64 runs = 2 ms of a 32 ms call).
extern "C" int fm_expand_fetch_many(fm_ctx* ctx, int32_t n_ex, const fm_expand* const* ex, const int32_t* slot, const int64_t* n,
                                    int32_t* const* index, double* const* positions, double* const* ratio)
{
    if (!ctx) return fail(nullptr, FM_EINVAL, "fm_expand_fetch_many: ctx is NULL");
    if (n_ex < 0) return fail(ctx, FM_EINVAL, "fm_expand_fetch_many: n_ex < 0");
    if (n_ex == 0) return FM_OK;
    if (!ex || !n) return fail(ctx, FM_EINVAL, "fm_expand_fetch_many: NULL argument");
    size_t total = 0;
    std::vector<const ExpandRun*> run((size_t)n_ex);
    for (int i = 0; i < n_ex; ++i) {
        if (!ex[i]) return fail(ctx, FM_EINVAL, "fm_expand_fetch_many: NULL pair");
        const int s = slot ? slot[i] : expand_slot_of((fm_expand* const*)ex, i);
        if (s < 0 || (size_t)s >= ex[i]->runs.size()) return fail(ctx, FM_EINVAL, "fm_expand_fetch_many: the pair has no such run slot");
        run[(size_t)i] = &ex[i]->runs[(size_t)s];
        if (n[i] < 0 || n[i] > run[(size_t)i]->match_cap) return fail(ctx, FM_EINVAL, "fm_expand_fetch_many: n out of range");
        total += (size_t)n[i] * 44 + 192;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    struct StageGuard {      // drop the staged copies on every exit path (their targets die with the call)
        fm_ctx* c;
        ~StageGuard() { c->staged.clear(); c->h_stage_used = 0; }
    } guard{ctx};
    ctx->staged.clear();
    ctx->h_stage_used = 0;
    if (total > ctx->h_stage_bytes) {          // one staging area for all of it (d2h grows it only while nothing is staged)
        if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
        ctx->h_stage = nullptr;
        ctx->h_stage_bytes = 0;
        const size_t want = total + total / 4 + (1 << 20);
        if (hipHostMalloc((void**)&ctx->h_stage, want, hipHostMallocDefault) == hipSuccess) ctx->h_stage_bytes = want;
        else { (void)hipGetLastError(); ctx->h_stage = nullptr; }
    }
    for (int i = 0; i < n_ex; ++i) {
        if (n[i] == 0) continue;
        if (index && index[i]) HIP_TRY(ctx, d2h(ctx, index[i], run[(size_t)i]->m_index, (size_t)n[i] * 4));
        if (positions && positions[i]) HIP_TRY(ctx, d2h(ctx, positions[i], run[(size_t)i]->m_pos, (size_t)n[i] * 32));
        if (ratio && ratio[i]) HIP_TRY(ctx, d2h(ctx, ratio[i], run[(size_t)i]->m_ratio, (size_t)n[i] * 8));
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (const auto& c : ctx->staged) memcpy(c.dst, ctx->h_stage + c.off, c.bytes);
    return FM_OK;
}

// ---------------------------------------------------------------------------------------
// result gather over RCCL (comm.hip)
// ---------------------------------------------------------------------------------------
extern "C" int fm_comm_unique_id(void* id128)
{
    if (!id128) return fail(nullptr, FM_EINVAL, "fm_comm_unique_id: NULL buffer");
    std::string err;
    const int rc = comm_unique_id(id128, &err);
    return rc == FM_OK ? FM_OK : fail(nullptr, rc, "fm_comm_unique_id: " + err);
}

extern "C" int fm_comm_init(fm_ctx* ctx, int nranks, int rank, const void* id128)
{
    if (!ctx) return fail(nullptr, FM_EINVAL, "fm_comm_init: ctx is NULL");
    if (!id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(ctx, FM_EINVAL, "fm_comm_init: bad argument");
    if (ctx->comm) return fail(ctx, FM_EINVAL, "fm_comm_init: the context already has a communicator (fm_comm_destroy first)");
    std::string err;
    void* comm = nullptr;
    const int rc = comm_init(ctx->device, nranks, rank, id128, &comm, &err);
    if (rc != FM_OK) return fail(ctx, rc, "fm_comm_init: " + err);
    ctx->comm = comm;
    ctx->comm_ranks = nranks;
    return FM_OK;
}

extern "C" int fm_comm_destroy(fm_ctx* ctx)
{
    if (!ctx) return FM_OK;
    if (ctx->comm) {
        sync_all_streams(ctx);               // (a gather behind an async fill runs on a tail stream)
        comm_destroy(ctx->comm);
        ctx->comm = nullptr;
        ctx->comm_ranks = 0;
    }
    return FM_OK;
}

// Two-phase form: the counts first, then only as many rows per rank as the fullest rank holds (the padded
// form ships cap rows per rank whatever they hold).  Costs a host synchronisation between the phases.
extern "C" int fm_gather_matches_counted(fm_ctx* ctx, const int32_t* d_rows, const int64_t* d_count, int64_t cap,
                                         int32_t* d_all_rows, int64_t* d_all_counts, int64_t* rows_per_rank)
{
    if (!ctx) return fail(nullptr, FM_EINVAL, "fm_gather_matches_counted: ctx is NULL");
    if (!ctx->comm) return fail(ctx, FM_EINVAL, "fm_gather_matches_counted: no communicator (fm_comm_init)");
    if (cap < 0 || !d_count || !d_all_counts || !rows_per_rank || (cap > 0 && (!d_rows || !d_all_rows)))
        return fail(ctx, FM_EINVAL, "fm_gather_matches_counted: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::string err;
    hipStream_t gs = ctx->rows_stream ? ctx->rows_stream : ctx->stream;
    int rc = comm_gather(ctx->comm, nullptr, d_count, 0, nullptr, d_all_counts, gs, &err);
    if (rc != FM_OK) return fail(ctx, rc, "fm_gather_matches_counted: " + err);
    std::vector<int64_t> counts((size_t)ctx->comm_ranks);
    HIP_TRY(ctx, hipMemcpyAsync(counts.data(), d_all_counts, counts.size() * 8, hipMemcpyDeviceToHost, gs));
    HIP_TRY(ctx, hipStreamSynchronize(gs));
    int64_t m = 0;
    for (int64_t c : counts) m = c > m ? c : m;
    if (m > cap) m = cap;
    *rows_per_rank = m;
    if (m > 0) {
        rc = comm_gather(ctx->comm, d_rows, nullptr, m, d_all_rows, nullptr, gs, &err);
        if (rc != FM_OK) return fail(ctx, rc, "fm_gather_matches_counted: " + err);
        HIP_TRY(ctx, hipStreamSynchronize(gs));
    }
    return FM_OK;
}

extern "C" int fm_gather_matches(fm_ctx* ctx, const int32_t* d_rows, const int64_t* d_count, int64_t cap,
                                 int32_t* d_all_rows, int64_t* d_all_counts, int wait)
{
    if (!ctx) return fail(nullptr, FM_EINVAL, "fm_gather_matches: ctx is NULL");
    if (!ctx->comm) return fail(ctx, FM_EINVAL, "fm_gather_matches: no communicator (fm_comm_init)");
    if (cap < 0 || !d_count || !d_all_counts || (cap > 0 && (!d_rows || !d_all_rows)))
        return fail(ctx, FM_EINVAL, "fm_gather_matches: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::string err;
    // on the stream that filled d_rows: behind fm_match_accepted_dev_async that is the tail stream, so
    // the collective does not sit in front of the next pair's K1
    hipStream_t gs = ctx->rows_stream ? ctx->rows_stream : ctx->stream;
    const int rc = comm_gather(ctx->comm, d_rows, d_count, cap, d_all_rows, d_all_counts, gs, &err);
    if (rc != FM_OK) return fail(ctx, rc, "fm_gather_matches: " + err);
    if (wait) HIP_TRY(ctx, hipStreamSynchronize(gs));
    return FM_OK;
}
