// C-ABI of libfastmatch_hip.so (include/fastmatch_hip.h), part 2: the matcher entry points and the small kernels
// around K1 --
//   top-2 merge of the split partials for knnMatch(k = 2), Lowe ratio
//   cross-check: election (reverse-NN partials -> 64-bit scatter-min), decode + float64 ratio + threshold,
//   ordered compaction of the accepted matches (host arrays or 12-byte device rows)
//   exact repair of float32-root ties (sqrt_fix_kernel)
// and the pipelines built from them: synchronous calls, async calls on two streams, batches of pairs that
// share distance-kernel launches, sharded election keys, per-round launches (K4).
// Reference call sites are cited in the header next to each entry point.
#include "ctx_internal.h"

using namespace fm;

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
    // The float32 order can differ from the d2 order only if the root class of the best or of the 2nd best d2
    // has a second member (d2 - 1 or d2 + 1 shares its root): only then can a candidate outside this list, or
    // the other list entry, tie with it and win on its index.
    if (fix && b1 != ~0ull && (unsigned)(b1 >> 32) >= kSqrtTieMin) {
        const unsigned e0 = (unsigned)(b0 >> 32), e1 = (unsigned)(b1 >> 32);
        const bool paired1 = sqrt_ties_up(e1) || sqrt_ties_up(e1 - 1u);
        const bool paired0 = e0 >= kSqrtTieMin && (sqrt_ties_up(e0) || sqrt_ties_up(e0 - 1u));
        if (paired0 || paired1) fix[4 + atomicAdd(fix, 1u)] = (unsigned)i;
    }
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

// Self distances (cache.pyx:250-252, 271-273: [r[1].distance for r in bf_match(d, d, k = 2)]) from the split partials
// of the masked-diagonal top-1: row i's value is the float32 distance to its nearest OTHER row, as float64; +inf
// for a bank of one row.  Only the value is kept, so the float32-root ties of the integer route need no repair
// (the smallest root is the root of the smallest d2).  Up to kRRBatchMax banks of one shape per launch
// (blockIdx.y); bound_reset: the sweep is complete, its bound[] array goes back to "no bound".
__global__ void selfdist_from_knn_kernel(const float* __restrict__ dist2, int64_t n, double* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (double)dist2[2 * i + 1];
}

struct SelfMerge {
    const unsigned long long* partial[kRRBatchMax];
    double* out[kRRBatchMax];
    int*    bound_reset[kRRBatchMax];
};

__global__ void selfdist_merge_kernel(SelfMerge m, int nsplit, int ncols_alloc, int64_t n, int f32)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (m.bound_reset[b] && i < ncols_alloc) m.bound_reset[b][i] = INT32_MIN;
    if (i >= n) return;
    const unsigned long long* p = m.partial[b];
    unsigned long long k = ~0ull;
    for (int s = 0; s < nsplit; ++s) {
        const unsigned long long v = p[(size_t)s * ncols_alloc + i];
        k = v < k ? v : k;
    }
    m.out[b][i] = (k == ~0ull) ? (double)INFINITY : (double)key_dist(k, f32);
}

// The triangular self sweep (rowreduce.hip, TRI) leaves, per row, the best hi it has reached as an output row in
// bound[]: d2 = |row|^2 + 1 - bound; a word at or below kTriNoBoundHost met no real row (a bank of one row).
struct TriFinish {
    const int* bound[kRRBatchMax];
    const int32_t* norm[kRRBatchMax];
    double* out[kRRBatchMax];
    int64_t n[kRRBatchMax];
};

__global__ void selfdist_tri_finish_kernel(TriFinish m)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= m.n[b]) return;
    const int h = m.bound[b][i];
    m.out[b][i] = (h > kTriNoBoundHost) ? (double)sqrtf((float)(unsigned)(m.norm[b][i] + 1 - h)) : (double)INFINITY;
}

extern "C" int fm_self_dist_plan(int64_t n_pad, int32_t stages, int32_t* table, int64_t cap, int32_t* n_workgroups,
                                 int32_t* n_diag, int32_t* stages_used)
{
    if (n_pad < kStageRows || n_pad % kStageRows != 0 || n_pad > kTriMaxRows || stages < 0 || cap < 0 || (cap > 0 && !table))
        return fail(nullptr, FM_EINVAL, "fm_self_dist_plan: n_pad must be a positive multiple of 128 up to 2^26, stages >= 0");
    try {
        // (the count-only call -- cap 0 -- builds no table: 16 bytes per workgroup, a million of them for a 3M-row bank)
        std::vector<int> tb;
        const TriPlan pl = plan_tri(n_pad, stages, cap > 0 ? &tb : nullptr);
        if (n_workgroups) *n_workgroups = pl.npieces;
        if (n_diag) *n_diag = pl.ndiag;
        if (stages_used) *stages_used = pl.stages;
        const int64_t m = cap < pl.npieces ? cap : pl.npieces;
        for (int64_t i = 0; i < 4 * m; ++i) table[i] = tb[(size_t)i];
    } catch (const std::bad_alloc&) {
        return fail(nullptr, FM_ENOMEM, "fm_self_dist_plan: out of host memory");
    }
    return FM_OK;
}

// The triangular sweep's plan of a bank size: piece length and workgroup counts by arithmetic (plan_tri, api_grid.hip); the
// workgroups themselves follow from (chunks, stages, piece length) in the kernel (tri_entry, rowreduce.hip), so a plan owns
// no device memory (r05, last) and the context keeps plans only to spare the arithmetic of a size it has seen.
static int tri_plan_for(fm_ctx* ctx, int64_t n_pad, TriPlan* out)
{
    const std::pair<int64_t, int> key(n_pad, ctx->tune.tri_stages * 2048 + ctx->tune.bound_every);
    auto it = ctx->tri_plans.find(key);
    if (it != ctx->tri_plans.end()) { *out = it->second; return FM_OK; }
    if (ctx->tri_plans.size() >= 4096) ctx->tri_plans.clear();
    TriPlan pl = plan_tri(n_pad, ctx->tune.tri_stages, nullptr);
    pl.bound_every = 1;
    for (int b = 2; b <= 1024; b <<= 1) if (ctx->tune.bound_every == b) pl.bound_every = b;
    ctx->tri_plans[key] = pl;
    *out = pl;
    return FM_OK;
}

// ---------------------------------------------------------------------------------------
// float32 route: K5 alone, or the fp16 filter (K8) with K5 as its conditional fallback
// ---------------------------------------------------------------------------------------
// Leaves the packed keys in ws_partial in `pl`'s layout (K5's plan) either way.
static int rowreduce_f32_route(fm_ctx* ctx, const fm_bank* cols, const fm_bank* red, int ktop, RowReducePlan* pl_out, bool self = false)
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
        HIP_TRY(ctx, launch_rowreduce_f32(*cols, *red, ktop, pl, (unsigned long long*)ctx->ws_partial, nullptr, ctx->stream, self));
        HIP_TRY(ctx, hipEventRecord(ctx->ev_k1, ctx->stream));
        return FM_OK;
    }
    const FilterPlan fp = plan_filter(cols->n_pad, red->n_pad, ctx->tune);
    const size_t bnd = (fp.bound_bytes() + 255) & ~(size_t)255;
    int rc = ws_ensure(ctx, &ctx->ws_partial, &ctx->ws_partial_bytes, part + fp.slots_bytes() + bnd + fp.aux_bytes() + 64);
    if (rc != FM_OK) return rc;
    unsigned long long* d_part = (unsigned long long*)ctx->ws_partial;
    unsigned long long* d_slots = (unsigned long long*)((char*)ctx->ws_partial + part);
    int* d_bound = (int*)((char*)d_slots + fp.slots_bytes());
    float* d_aux = (float*)((char*)d_bound + bnd);
    HIP_TRY(ctx, hipMemsetAsync(d_part, 0xff, pl.partial_bytes(ktop), ctx->stream));
    HIP_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)d_bound, filter_empty_bound(), (size_t)fp.ncols_alloc * 2, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_counters, 0, 8, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_k0, ctx->stream));
    HIP_TRY(ctx, launch_filter(*cols, *red, ktop, fp, d_slots, d_bound, ctx->d_counters, d_part, ctx->stream, self, d_aux));
    HIP_TRY(ctx, launch_rowreduce_f32(*cols, *red, ktop, pl, d_part, ctx->d_counters, ctx->stream, self));
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
// K7's delegated cross-check: one expansion round's X1 on the whole GPU
// ---------------------------------------------------------------------------------------
// The rows rows[0 .. nq) of bank q gathered into a bank image of their own (rows, norms, aux words in the layout
// bank_prep_kernel writes; slots past nq are padding rows).  One 256-thread block per 32-row tile.
__global__ __launch_bounds__(256)
void gather_rows_kernel(const int32_t* __restrict__ rows, int64_t nq, const int8_t* __restrict__ src8, const int32_t* __restrict__ srcnorm,
                        int8_t* __restrict__ dst8, int32_t* __restrict__ dstnorm, int32_t* __restrict__ dstaux,
                        unsigned long long* __restrict__ qbest, int* __restrict__ bound, int64_t nbound)
{
    const int tid = threadIdx.x, r = tid >> 3, c = tid & 7;
    const int64_t tile = blockIdx.x, slot = tile * kTileRows + r;
    // (the two fills the round's other kernels want, here instead of two fill launches of their own: the election's table
    // and K1's shared bounds)
    if (c == 1 && slot < nq) qbest[slot] = ~0ull;
    if (bound)
        for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < nbound; i += (int64_t)gridDim.x * 256) bound[i] = (int)0x80000000;
    uint4 w = make_uint4(0, 0, 0, 0);
    int nm = 0;
    if (slot < nq) {
        const int64_t qi = rows[slot];
        w = *(const uint4*)(src8 + qi * kDim + 16 * c);
        nm = srcnorm[qi];
    }
    *(uint4*)(dst8 + slot * kDim + 16 * c) = w;
    if (c == 0) {
        const int sub = r >> 4, rr = r & 15, id = 4 * sub + (rr & 3);
        int32_t* a = dstaux + tile * kAuxPerTile + 32 * sub;
        if (slot < nq) { dstnorm[slot] = nm; a[rr] = -(nm >> 1); a[16 + rr] = ((1 - (nm & 1)) << 4) | (15 - id); }
        else           { dstnorm[slot] = 0;  a[rr] = kPadCinit;  a[16 + rr] = 15 - id; }
    }
}

// The same for a float32 bank: float32 rows, fp16 filter rows, scaled norms and accumulator inits, in the layout
// bank_copy_f32_kernel / bank_prep_f16_kernel write (padding slots: zero rows, init -3.4e38).  16 rows per 256-thread block.
__global__ __launch_bounds__(256)
void gather_rows_f32_kernel(const int32_t* __restrict__ rows, int64_t nq, const float* __restrict__ srcf, const uint16_t* __restrict__ srch,
                            const float* __restrict__ srcnorm, const float* __restrict__ srcaux, float* __restrict__ dstf,
                            uint16_t* __restrict__ dsth, float* __restrict__ dstnorm, float* __restrict__ dstaux,
                            unsigned long long* __restrict__ qbest)
{
    const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
    const int64_t slot = (int64_t)blockIdx.x * 16 + r;
    float4 f0 = make_float4(0.f, 0.f, 0.f, 0.f), f1 = f0;
    uint4 h = make_uint4(0, 0, 0, 0);
    float nm = 0.f, ax = -3.4e38f;
    if (slot < nq) {
        const int64_t qi = rows[slot];
        f0 = *(const float4*)(srcf + qi * kDim + 8 * c);
        f1 = *(const float4*)(srcf + qi * kDim + 8 * c + 4);
        h = *(const uint4*)(srch + qi * kDim + 8 * c);
        nm = srcnorm[qi];
        ax = srcaux[qi];
        if (c == 1) qbest[slot] = ~0ull;
    }
    *(float4*)(dstf + slot * kDim + 8 * c) = f0;
    *(float4*)(dstf + slot * kDim + 8 * c + 4) = f1;
    *(uint4*)(dsth + slot * kDim + 8 * c) = h;
    if (c == 0) { dstnorm[slot] = nm; dstaux[slot] = ax; }
}

// Cross-checked 1-NN of the query subset d_rows[0 .. nq) of bank q against the train rows [t0, t0 + nt) of bank t, the
// way fm_xcheck1 does it on gathered banks (reverse NN by K1 with the train rows as output rows, election by
// scatter-min), into d_qbest[slot] = (float32 distance bits << 32 | local train row), ~0 = unmatched.  Enqueued on the
// context's stream; the workspaces are the context's (ws_out: the gathered bank, ws_partial: K1's partials).  For
// integer-route pairs outside the float32-root tie range (the caller checks).
int fm::round_xcheck_dense(fm_ctx* ctx, const Bank& q, const int32_t* d_rows, int64_t nq, const Bank& t, int64_t t0, int64_t nt,
                           unsigned long long* d_qbest)
{
    if (nq <= 0 || nt <= 0) return FM_OK;
    const int64_t nq_pad = ((nq + kStageRows - 1) / kStageRows) * kStageRows;
    size_t off = 0;
    auto carve = [&](size_t b) { size_t o = off; off += (b + 255) & ~(size_t)255; return o; };
    if (q.kind == FM_BANK_F32) {
        // float32 banks: the gathered rows in the float32 layout, then the float32 route's own cross-check (fp16 filter +
        // exact rescoring, or all pairs for small rounds: rowreduce_f32_route) with the cell's rows as output rows
        const size_t o_f = carve((size_t)nq_pad * kDim * 4), o_h = carve((size_t)nq_pad * kDim * 2), o_n = carve((size_t)nq_pad * 4),
                     o_a = carve((size_t)nq_pad * 4);
        int rc = ws_ensure(ctx, &ctx->ws_out, &ctx->ws_out_bytes, off + 64);
        if (rc != FM_OK) return rc;
        char* b = (char*)ctx->ws_out;
        fm_bank gq, tv;
        gq.kind = FM_BANK_F32; gq.n = nq; gq.dim = q.dim; gq.n_pad = nq_pad; gq.cap_pad = nq_pad;
        gq.rowsf = (float*)(b + o_f); gq.rowsh = (uint16_t*)(b + o_h); gq.normf = (float*)(b + o_n); gq.auxf = (float*)(b + o_a);
        gq.nm_max = q.nm_max; gq.kscale = q.kscale; gq.filt_ok = q.filt_ok;
        hipLaunchKernelGGL(gather_rows_f32_kernel, dim3((unsigned)(nq_pad / 16)), dim3(256), 0, ctx->stream, d_rows, nq,
                           (const float*)q.rowsf, (const uint16_t*)q.rowsh, (const float*)q.normf, (const float*)q.auxf,
                           gq.rowsf, gq.rowsh, gq.normf, gq.auxf, d_qbest);
        HIP_TRY(ctx, hipGetLastError());
        tv.kind = FM_BANK_F32; tv.n = nt; tv.dim = t.dim;
        tv.n_pad = ((nt + kStageRows - 1) / kStageRows) * kStageRows;
        const int64_t room = (t.cap_pad > 0 ? t.cap_pad : t.n_pad) - t0;
        if (tv.n_pad > room) tv.n_pad = room;
        tv.cap_pad = tv.n_pad;
        tv.rowsf = t.rowsf + (size_t)t0 * kDim; tv.rowsh = t.rowsh ? t.rowsh + (size_t)t0 * kDim : nullptr;
        tv.normf = t.normf ? t.normf + t0 : nullptr; tv.auxf = t.auxf ? t.auxf + t0 : nullptr;
        tv.nm_max = t.nm_max; tv.kscale = t.kscale; tv.filt_ok = t.filt_ok;
        RowReducePlan pl;
        if ((rc = rowreduce_f32_route(ctx, &tv, &gq, 1, &pl)) != FM_OK) return rc;
        const int64_t sthreads = nt * 4;
        hipLaunchKernelGGL(xcheck_scatter_kernel, dim3((unsigned)((sthreads + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const unsigned long long*)ctx->ws_partial, pl.nsplit, pl.ncols_alloc, nt, d_qbest, 0u, 1, (int*)nullptr, (unsigned*)nullptr);
        HIP_TRY(ctx, hipGetLastError());
        return FM_OK;
    }
    const size_t o_rows = carve((size_t)nq_pad * kDim), o_norm = carve((size_t)nq_pad * 4), o_aux = carve((size_t)(nq_pad / kTileRows) * kAuxPerTile * 4);
    int rc = ws_ensure(ctx, &ctx->ws_out, &ctx->ws_out_bytes, off + 64);
    if (rc != FM_OK) return rc;
    char* b = (char*)ctx->ws_out;
    Bank gq;
    gq.kind = FM_BANK_I8; gq.n = nq; gq.dim = q.dim; gq.n_pad = nq_pad; gq.cap_pad = nq_pad;
    gq.rows8 = (int8_t*)(b + o_rows); gq.norm = (int32_t*)(b + o_norm); gq.aux = (int32_t*)(b + o_aux);
    // the cell's rows as a bank of their own: a view into the target bank (rows past nt are other cells' rows or padding;
    // what K1 computes for them is never looked at)
    Bank tv;
    tv.kind = FM_BANK_I8; tv.n = nt; tv.dim = t.dim;
    tv.n_pad = ((nt + kStageRows - 1) / kStageRows) * kStageRows;
    const int64_t room = (t.cap_pad > 0 ? t.cap_pad : t.n_pad) - t0;
    if (tv.n_pad > room) tv.n_pad = room;
    tv.cap_pad = tv.n_pad;
    tv.rows8 = t.rows8 + (size_t)t0 * kDim; tv.norm = t.norm + t0; tv.aux = nullptr;
    const RowReducePlan pl = plan_rowreduce(tv.n_pad, gq.n_pad, ctx->tune);
    const size_t pbytes = (pl.partial_bytes(1) + 255) & ~(size_t)255;
    if ((rc = ws_ensure(ctx, &ctx->ws_partial, &ctx->ws_partial_bytes, pbytes + pl.bound_bytes() + 64)) != FM_OK) return rc;
    int* d_bound = ((ctx->tune.coop != 0) && pl.nsplit > 1) ? (int*)((char*)ctx->ws_partial + pbytes) : nullptr;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)(nq_pad / kTileRows)), dim3(256), 0, ctx->stream, d_rows, nq,
                       (const int8_t*)q.rows8, (const int32_t*)q.norm, gq.rows8, gq.norm, gq.aux, d_qbest, d_bound, (int64_t)pl.ncols_alloc);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, launch_rowreduce(tv, gq, 1, pl, (unsigned long long*)ctx->ws_partial, d_bound, (ctx->tune.glds != 0), ctx->stream));
    const int64_t sthreads = nt * 4;
    hipLaunchKernelGGL(xcheck_scatter_kernel, dim3((unsigned)((sthreads + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const unsigned long long*)ctx->ws_partial, pl.nsplit, pl.ncols_alloc, nt, d_qbest, 0u, 0, (int*)nullptr, (unsigned*)nullptr);
    HIP_TRY(ctx, hipGetLastError());
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
    ctx->pending_bytes += bank_bytes(q) + bank_bytes(t);
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

// knnMatch(dt1, dt2, k) for any k the signature of the reference's bf_match / flann_match admits (matchutil.py:39-43, 46-67).
extern "C" int fm_knn(fm_ctx* ctx, const fm_bank* q, const fm_bank* t, int32_t k, int32_t* idx, float* dist)
{
    int rc = check_pair(ctx, q, t, "fm_knn");
    if (rc != FM_OK) return rc;
    if (k < 1) return fail(ctx, FM_EINVAL, "fm_knn: k must be at least 1");
    if (k > 8) return fail(ctx, FM_EUNSUPPORTED, "fm_knn: k above 8 is not built (cv2.BFMatcher.knnMatch takes any k; the reference calls it with 1 and 2)");
    const int64_t nq = q->n;
    if (nq > 0 && (!idx || !dist)) return fail(ctx, FM_EINVAL, "fm_knn: output pointer is NULL");
    if (nq == 0) return FM_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // idx [nq][max(k, 2)] | dist [nq][max(k, 2)] | (k = 1) first columns idx1 [nq], dist1 [nq]
    const size_t out_bytes = (((size_t)nq * (size_t)(k < 2 ? 2 : k) * 4) + 255) & ~(size_t)255;
    const size_t col_bytes = (((size_t)nq * 4) + 255) & ~(size_t)255;
    if ((rc = ws_ensure(ctx, &ctx->ws_out, &ctx->ws_out_bytes, 2 * out_bytes + 2 * col_bytes + 64)) != FM_OK) return rc;
    int32_t* d_idx = (int32_t*)ctx->ws_out;
    float* d_dist = (float*)((char*)ctx->ws_out + out_bytes);
    int32_t* d_idx1 = (int32_t*)((char*)ctx->ws_out + 2 * out_bytes);
    float* d_dist1 = (float*)((char*)ctx->ws_out + 2 * out_bytes + col_bytes);
    CallScope cs(ctx);
    if (k <= 2) {
        // the matrix-core path; k = 1 is the first column of the 2-NN lists
        if ((rc = knn2_device(ctx, q, t, d_idx, d_dist)) != FM_OK) return rc;
        if (k == 2) {
            HIP_TRY(ctx, d2h(ctx, idx, d_idx, (size_t)nq * 8));
            HIP_TRY(ctx, d2h(ctx, dist, d_dist, (size_t)nq * 8));
        } else {
            HIP_TRY(ctx, hipMemcpy2DAsync(d_idx1, 4, d_idx, 8, 4, (size_t)nq, hipMemcpyDeviceToDevice, ctx->stream));
            HIP_TRY(ctx, hipMemcpy2DAsync(d_dist1, 4, d_dist, 8, 4, (size_t)nq, hipMemcpyDeviceToDevice, ctx->stream));
            HIP_TRY(ctx, d2h(ctx, idx, d_idx1, (size_t)nq * 4));
            HIP_TRY(ctx, d2h(ctx, dist, d_dist1, (size_t)nq * 4));
        }
        return cs.finish();
    }
    if ((rc = ws_ensure(ctx, &ctx->ws_partial, &ctx->ws_partial_bytes, knnk_partial_bytes(nq, t->n, k) + 64)) != FM_OK) return rc;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_k0, ctx->stream));
    HIP_TRY(ctx, launch_knnk(*q, *t, k, (unsigned long long*)ctx->ws_partial, d_idx, d_dist, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_k1, ctx->stream));
    ctx->kernel_timed = true;
    ctx->pending_pairs += nq * t->n;
    ctx->pending_bytes += bank_bytes(q) + bank_bytes(t);
    HIP_TRY(ctx, d2h(ctx, idx, d_idx, (size_t)nq * k * 4));
    HIP_TRY(ctx, d2h(ctx, dist, d_dist, (size_t)nq * k * 4));
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

static int take_timer(fm_ctx* ctx, fm_ctx::PendingTimer* tm);

// Self distances of n banks on the context's stream, each into d_out[i] (device, [banks[i]->n] float64).
// Integer-valued banks: the triangular sweep (launch_rowreduce_tri: from 32768 padded rows on, and for every run of two or
// more integer banks, of any sizes, up to "batch_group" banks per launch pair) or the masked-diagonal top-1 sweep (K1's top-1
// kernel; launch_rowreduce_self; consecutive banks of one padded size through ONE launch of rowreduce_batch_kernel);
// float32 banks: the float32 route with the diagonal masked (K8 filter + exact rescoring, or K5), one by one.
// Banks with non-finite values keep the literal form (2-NN, second column).  Enqueue only; the split partials
// live in ws_partial, which every user touches on ctx->stream only.  timed: record ev_k0 / ev_k1 around the
// (last) distance-kernel launch.
static int selfdist_device(fm_ctx* ctx, int n, const fm_bank* const* banks, double* const* d_out, bool timed)
{
    const int group_max = ctx->tune.batch_group;
    int i = 0;
    while (i < n) {
        const fm_bank* b = banks[i];
        if (b->n == 0) { ++i; continue; }
        int rc;
        if (b->kind == FM_BANK_F32) {
            SelfMerge m{};
            RowReducePlan pl;
            if (!b->filt_ok) {
                // non-finite values: no shortcut is claimed for them
                if ((rc = ws_ensure(ctx, &ctx->ws_out, &ctx->ws_out_bytes, (size_t)b->n * 16 + 64)) != FM_OK) return rc;
                int32_t* d_idx = (int32_t*)ctx->ws_out;
                float* d_dist = (float*)((char*)ctx->ws_out + (size_t)b->n * 8);
                if ((rc = knn2_device(ctx, b, b, d_idx, d_dist)) != FM_OK) return rc;
                hipLaunchKernelGGL(selfdist_from_knn_kernel, dim3((unsigned)((b->n + 255) / 256)), dim3(256), 0, ctx->stream,
                                   (const float*)d_dist, b->n, d_out[i]);
                HIP_TRY(ctx, hipGetLastError());
                ++i;
                continue;
            }
            // r06: every distance once -- the triangular sweep around the fp16 filter (filter_f16.hip, TRI) from 65536 padded
            // rows on ("self_tri" 1; 2 = always, 0 = never), else the masked full sweep through the filter / K5.  (Measured,
            // profiles/r06e_f32_selfdist_tri.log: x 1.00 of the masked sweep's time at 50k rows, 0.94 at 65k, 0.83 at 100k,
            // 0.76 at 200k -- below ~60k rows the sweep has fewer pieces than the chip holds workgroups, every row's visits
            // happen at once against a stale bound, and the lists overflow into rescans.)
            const bool tri = ctx->tune.f32_filter != 0 && ctx->tune.self_tri != 0 && b->n_pad <= kTriMaxRows &&
                             (ctx->tune.self_tri == 2 || b->n_pad >= 65536) && filter_usable(*b, *b);
            if (tri) {
                pl = plan_rowreduce_f32(b->n_pad, b->n_pad, ctx->tune.nsplit);
                TriPlan tp;
                if ((rc = tri_plan_for(ctx, b->n_pad, &tp)) != FM_OK) return rc;
                const size_t part = (pl.partial_bytes(1) + 255) & ~(size_t)255;
                const size_t bnd = ((size_t)tp.ncols_alloc * 4 + 255) & ~(size_t)255;
                if ((rc = ws_ensure(ctx, &ctx->ws_partial, &ctx->ws_partial_bytes, part + bnd + filter_tri_bytes(tp.ncols_alloc) + 64)) != FM_OK) return rc;
                unsigned long long* d_part = (unsigned long long*)ctx->ws_partial;
                int* d_bound = (int*)((char*)ctx->ws_partial + part);
                void* d_tri = (char*)d_bound + bnd;
                HIP_TRY(ctx, hipMemsetAsync(d_part, 0xff, pl.partial_bytes(1), ctx->stream));
                HIP_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)d_bound, filter_empty_bound(), (size_t)tp.ncols_alloc, ctx->stream));
                HIP_TRY(ctx, hipMemsetAsync(ctx->d_counters, 0, 8, ctx->stream));
                HIP_TRY(ctx, hipEventRecord(ctx->ev_k0, ctx->stream));
                HIP_TRY(ctx, launch_filter_tri(*b, tp, ctx->tune.f32_bound_every, d_tri, d_bound, ctx->d_counters, d_part, ctx->stream));
                HIP_TRY(ctx, launch_rowreduce_f32(*b, *b, 1, pl, d_part, ctx->d_counters, ctx->stream, true));
                HIP_TRY(ctx, hipEventRecord(ctx->ev_k1, ctx->stream));
                ctx->filter_launches += 1;
            } else if ((rc = rowreduce_f32_route(ctx, b, b, 1, &pl, true)) != FM_OK) return rc;
            ctx->kernel_timed = true;
            ctx->pending_pairs += b->n * b->n;
            ctx->pending_bytes += bank_bytes(b);
            m.partial[0] = (const unsigned long long*)ctx->ws_partial;
            m.out[0] = d_out[i];
            hipLaunchKernelGGL(selfdist_merge_kernel, dim3((unsigned)((b->n + 255) / 256), 1), dim3(256), 0, ctx->stream,
                               m, pl.nsplit, pl.ncols_alloc, b->n, 1);
            HIP_TRY(ctx, hipGetLastError());
            ++i;
            continue;
        }
        // Every distance once: the triangular sweep, up to "batch_group" banks per launch, every bank under the plan of its
        // own size.  Rule ("self_tri" 1): a bank of 32768 padded rows or more, or ANY run of two or more integer banks -- a
        // dataset of small images (r05, last: 64 banks of ~12.5k rows took 65 us each one by one through the full sweep and
        // take 23 us each in batched triangular launches; ~3k rows: 35 -> 9.5 us).
        int g = 1;
        const bool tri_fits = b->n_pad <= kTriMaxRows;
        if (ctx->tune.glds != 0 && ctx->tune.self_tri != 0 && tri_fits)
            while (i + g < n && g < group_max && g < kRRBatchMax && banks[i + g]->kind == FM_BANK_I8 && banks[i + g]->n > 0 &&
                   banks[i + g]->n_pad <= kTriMaxRows) ++g;
        if (ctx->tune.glds != 0 && tri_fits && (ctx->tune.self_tri == 2 || (ctx->tune.self_tri == 1 && (b->n_pad >= 32768 || g >= 2)))) {
            TriPlan tps[kRRBatchMax];
            size_t boff[kRRBatchMax + 1];
            boff[0] = 0;
            for (int j = 0; j < g; ++j) {
                if ((rc = tri_plan_for(ctx, banks[i + j]->n_pad, &tps[j])) != FM_OK) return rc;
                boff[j + 1] = boff[j] + (((size_t)tps[j].ncols_alloc * 4 + 255) & ~(size_t)255);
            }
            if ((rc = ws_ensure(ctx, &ctx->ws_partial, &ctx->ws_partial_bytes, boff[g])) != FM_OK) return rc;
            const Bank* bk[kRRBatchMax];
            int* bnd[kRRBatchMax];
            TriFinish fin{};
            int64_t nmax = 0;
            for (int j = 0; j < g; ++j) {
                bk[j] = banks[i + j];
                bnd[j] = (int*)((char*)ctx->ws_partial + boff[j]);
                fin.bound[j] = bnd[j]; fin.norm[j] = bk[j]->norm; fin.out[j] = d_out[i + j]; fin.n[j] = bk[j]->n;
                nmax = bk[j]->n > nmax ? bk[j]->n : nmax;
                // (each distance is computed once; the pairs a caller asked for are still n x n)
                ctx->pending_pairs += bk[j]->n * bk[j]->n;
                ctx->pending_bytes += bank_bytes(bk[j]);
            }
            HIP_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)bnd[0], (int)0x80000000, boff[g] / 4, ctx->stream));
            if (timed) HIP_TRY(ctx, hipEventRecord(ctx->ev_k0, ctx->stream));
            HIP_TRY(ctx, launch_rowreduce_tri(g, bk, tps, bnd, ctx->tune.prio != 0, ctx->stream));
            if (timed) HIP_TRY(ctx, hipEventRecord(ctx->ev_k1, ctx->stream));
            ctx->kernel_timed = timed;
            hipLaunchKernelGGL(selfdist_tri_finish_kernel, dim3((unsigned)((nmax + 255) / 256), (unsigned)g), dim3(256), 0, ctx->stream, fin);
            HIP_TRY(ctx, hipGetLastError());
            i += g;
            continue;
        }
        const RowReducePlan pl = plan_rowreduce_self(b->n_pad, ctx->tune);
        g = 1;
        if (pl.nw == 8 && (ctx->tune.glds != 0) && pl.nbuf != 2)
            while (i + g < n && g < group_max && banks[i + g]->kind == FM_BANK_I8 && banks[i + g]->n > 0 && banks[i + g]->n_pad == b->n_pad) ++g;
        const bool coop = (ctx->tune.coop != 0) && pl.nsplit > 1;
        const size_t pbytes = (pl.partial_bytes(1) + 255) & ~(size_t)255, bbytes = ((size_t)pl.ncols_alloc * 4 + 255) & ~(size_t)255;
        if ((rc = ws_ensure(ctx, &ctx->ws_partial, &ctx->ws_partial_bytes, (size_t)g * (pbytes + bbytes))) != FM_OK) return rc;
        SelfMerge m{};
        const Bank* bk[kRRBatchMax];
        unsigned long long* part[kRRBatchMax];
        int* bnd[kRRBatchMax];
        int64_t nmax = 0;
        for (int j = 0; j < g; ++j) {
            bk[j] = banks[i + j];
            // (partials first, then the g bound arrays back to back: one fill re-arms them all)
            part[j] = (unsigned long long*)((char*)ctx->ws_partial + (size_t)j * pbytes);
            bnd[j] = coop ? (int*)((char*)ctx->ws_partial + (size_t)g * pbytes + (size_t)j * bbytes) : nullptr;
            m.partial[j] = part[j];
            m.out[j] = d_out[i + j];
            nmax = bk[j]->n > nmax ? bk[j]->n : nmax;
            ctx->pending_pairs += bk[j]->n * bk[j]->n;
            ctx->pending_bytes += bank_bytes(bk[j]);
        }
        if (coop && !ablate_keep_bounds())
            HIP_TRY(ctx, hipMemsetD32Async((hipDeviceptr_t)bnd[0], (int)0x80000000, (size_t)g * (bbytes / 4), ctx->stream));
        if (timed) HIP_TRY(ctx, hipEventRecord(ctx->ev_k0, ctx->stream));
        if (g > 1) {
            RowReducePlan pls[kRRBatchMax];
            for (int j = 0; j < g; ++j) pls[j] = pl;
            HIP_TRY(ctx, launch_rowreduce_batch(g, bk, bk, pls, part, bnd, ctx->stream, true));
        }
        else       HIP_TRY(ctx, launch_rowreduce_self(*bk[0], pl, part[0], bnd[0], (ctx->tune.glds != 0), ctx->stream));
        if (timed) HIP_TRY(ctx, hipEventRecord(ctx->ev_k1, ctx->stream));
        ctx->kernel_timed = timed;
        // (banks of one padded size may hold different row counts: the merge covers the largest, rows beyond a
        // bank's own n are never written -- its out[] has n entries -- see the guard below)
        for (int j = 0; j < g; ++j) {
            SelfMerge one{};
            one.partial[0] = m.partial[j]; one.out[0] = m.out[j];
            if (bk[j]->n != nmax) {
                hipLaunchKernelGGL(selfdist_merge_kernel, dim3((unsigned)((bk[j]->n + 255) / 256), 1), dim3(256), 0, ctx->stream,
                                   one, pl.nsplit, pl.ncols_alloc, bk[j]->n, 0);
                m.partial[j] = nullptr;
            }
        }
        // the banks that share nmax go through one launch
        SelfMerge same{};
        int ns = 0;
        for (int j = 0; j < g; ++j) if (m.partial[j]) { same.partial[ns] = m.partial[j]; same.out[ns] = m.out[j]; ++ns; }
        if (ns > 0)
            hipLaunchKernelGGL(selfdist_merge_kernel, dim3((unsigned)((nmax + 255) / 256), (unsigned)ns), dim3(256), 0, ctx->stream,
                               same, pl.nsplit, pl.ncols_alloc, nmax, 0);
        HIP_TRY(ctx, hipGetLastError());
        i += g;
    }
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
    // (ws_in: ws_out is the 2-NN scratch of the non-finite float32 case)
    if ((rc = ws_ensure(ctx, &ctx->ws_in, &ctx->ws_in_bytes, (size_t)n * 8 + 64)) != FM_OK) return rc;
    double* d_sd = (double*)ctx->ws_in;
    CallScope cs(ctx);
    if ((rc = selfdist_device(ctx, 1, &bank, &d_sd, true)) != FM_OK) return rc;
    HIP_TRY(ctx, d2h(ctx, selfdist, d_sd, (size_t)n * 8));
    return cs.finish();
}

extern "C" int fm_self_dist_batch(fm_ctx* ctx, int32_t n, fm_bank* const* banks, double* const* out)
{
    if (!ctx) return fail(nullptr, FM_EINVAL, "fm_self_dist_batch: ctx is NULL");
    if (n < 0) return fail(ctx, FM_EINVAL, "fm_self_dist_batch: n < 0");
    if (n == 0) return FM_OK;
    if (!banks) return fail(ctx, FM_EINVAL, "fm_self_dist_batch: banks is NULL");
    for (int i = 0; i < n; ++i) {
        if (!banks[i]) return fail(ctx, FM_EINVAL, "fm_self_dist_batch: bank is NULL");
        for (int j = 0; j < i; ++j) if (banks[j] == banks[i]) return fail(ctx, FM_EINVAL, "fm_self_dist_batch: a bank is listed twice");
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::vector<double*> d_out((size_t)n, nullptr);
    std::vector<const fm_bank*> cb((size_t)n, nullptr);
    bool any_out = false;
    for (int i = 0; i < n; ++i) {
        fm_bank* b = banks[i];
        // (the first attachment allocates; a refilled bank keeps its array: fm_bank_refill_u8_async keeps the capacity)
        if (!b->selfdist) HIP_TRY(ctx, hipMalloc((void**)&b->selfdist, (size_t)(b->cap_pad > 0 ? b->cap_pad : 1) * 8));
        d_out[(size_t)i] = b->selfdist;
        cb[(size_t)i] = b;
        any_out = any_out || (out && out[i] && b->n > 0);
    }
    int rc;
    if (!any_out) {
        // enqueue only: accounted at the next fm_sync like the other asynchronous calls
        fm_ctx::PendingTimer tm;
        if ((rc = take_timer(ctx, &tm)) != FM_OK) return rc;
        const int64_t before = ctx->pending_pairs, before_b = ctx->pending_bytes;
        ctx->pending_pairs = 0;
        ctx->pending_bytes = 0;
        HIP_TRY(ctx, hipEventRecord(tm.k0, ctx->stream));
        rc = selfdist_device(ctx, n, cb.data(), d_out.data(), false);
        HIP_TRY(ctx, hipEventRecord(tm.k1, ctx->stream));
        HIP_TRY(ctx, hipEventRecord(tm.c1, ctx->stream));
        tm.timed = rc == FM_OK && ctx->pending_pairs > 0;
        tm.call_timed = true;
        tm.pairs = ctx->pending_pairs;
        tm.bytes = ctx->pending_bytes;
        ctx->pending_pairs = before;
        ctx->pending_bytes = before_b;
        ctx->pending.push_back(tm);
        return rc;
    }
    CallScope cs(ctx);
    if ((rc = selfdist_device(ctx, n, cb.data(), d_out.data(), true)) != FM_OK) return rc;
    for (int i = 0; i < n; ++i)
        if (out[i] && banks[i]->n > 0) HIP_TRY(ctx, d2h(ctx, out[i], d_out[(size_t)i], (size_t)banks[i]->n * 8));
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
        tm.bytes = bank_bytes(q) + bank_bytes(t);
        fm_ctx::AsyncSlot& sl = ctx->aslot[ctx->aslot_next];
        ctx->aslot_next ^= 1;
        const SlotLayout L = slot_layout(nq, nt, pl);
        if ((rc = slot_prepare(ctx, sl, L, nq, pl, ctx->stream)) != FM_OK) return rc;
        const bool coop = (ctx->tune.coop != 0) && pl.nsplit > 1;
        // Every event record is a packet the K1 launches of consecutive calls queue behind; the
        // start-of-kernel event is therefore taken for every async_time_every-th call only (those
        // calls are the ones fm_get_stats accounts as timed K1 launches; option async_time_every, default 4).
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
        ctx->pending_bytes += bank_bytes(q) + bank_bytes(t);
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
        ctx->pending_bytes += bank_bytes(q) + bank_bytes(t);
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
        // (r05, last: the pairs of a run need not share their padded sizes any more -- every pair brings its own plan, the
        // launch its own block ranges -- only the kernel shape: 4 blocks per wave, 8 waves, three stage buffers)
        // (a SMALL pair -- train bank below 32768 rows -- is planned with 4-wave workgroups when it runs alone, to fill the chip;
        // in a batched launch the other pairs do that, so it is planned again in the batched kernel's shape: 64 pairs of
        // ~12.5k x 12.5k rows 79 -> 44 us per pair, ~3k x 3k 49 -> 28, scripts/gpu_small_pairs.py)
        auto plan_fits = [&](int k, RowReducePlan* out) {
            if (ctx->tune.glds == 0) return false;
            RowReducePlan p = plan_rowreduce(t[k]->n_pad, q[k]->n_pad, ctx->tune);
            if (!(p.nb == 4 && p.nw == 8 && p.nbuf != 2) && ctx->tune.nb == 0 && ctx->tune.nw == 0) {
                Tuning shaped = ctx->tune;
                shaped.nb = 4; shaped.nw = 8;
                p = plan_rowreduce(t[k]->n_pad, q[k]->n_pad, shaped);
            }
            if (out) *out = p;
            return p.nb == 4 && p.nw == 8 && p.nbuf != 2;
        };
        int run = 1;
        if (batchable(i) && plan_fits(i, nullptr))
            while (i + run < n && batchable(i + run) && plan_fits(i + run, nullptr)) ++run;
        int g = run;
        if (run > group_max + tail_n) g = group_max;
        else if (run > 4 && tail_n > 0) g = run - tail_n < group_max ? run - tail_n : group_max;
        if (g > group_max) g = group_max;
        if (g < 1) g = 1;
        RowReducePlan pls[kRRBatchMax];
        for (int j = 0; j < g && g > 1; ++j) (void)plan_fits(i + j, &pls[j]);
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
        void* al[kRRBatchMax][5];
        int slot_of[kRRBatchMax];
        SlotLayout L[kRRBatchMax];
        const Bank* cols[kRRBatchMax]; const Bank* red[kRRBatchMax];
        unsigned long long* part[kRRBatchMax]; int* bnd[kRRBatchMax];
        fm_ctx::PendingTimer tm;
        if ((rc = take_timer(ctx, &tm)) != FM_OK) return rc;
        tm.timed = true; tm.call_timed = true; tm.pairs = 0; tm.bytes = 0;
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
            const RowReducePlan& pl = pls[j];
            const bool coop = (ctx->tune.coop != 0) && pl.nsplit > 1;
            L[j] = slot_layout(q[k]->n, t[k]->n, pl);
            if ((rc = slot_prepare(ctx, sl, L[j], q[k]->n, pl, ctx->stream)) != FM_OK) { ctx->timer_pool.push_back(tm); return rc; }
            cols[j] = t[k]; red[j] = q[k];            // reverse NN: output rows = train rows, reduced over the query rows
            part[j] = (unsigned long long*)sl.ws;
            bnd[j] = coop ? (int*)((char*)sl.ws + L[j].pbytes) : nullptr;
            tm.pairs += q[k]->n * t[k]->n;
            tm.bytes += bank_bytes(q[k]) + bank_bytes(t[k]);
        }
        HIP_TRY(ctx, hipEventRecord(tm.k0, ctx->stream));
        HIP_TRY(ctx, launch_rowreduce_batch(g, cols, red, pls, part, bnd, ctx->stream));
        HIP_TRY(ctx, hipEventRecord(tm.k1, ctx->stream));
        for (int j = 0; j < g; ++j) {
            const int k = i + j;
            hipStream_t ts = ctx->tails[j % fm_ctx::kTails];
            fm_ctx::AsyncSlot& sl = ctx->bslot[(size_t)slot_of[j]];
            HIP_TRY(ctx, hipStreamWaitEvent(ts, tm.k1, 0));
            if ((rc = enqueue_tail(ctx, ts, sl, L[j], q[k], t[k], q[k]->n, t[k]->n, pls[j], tau, cap, al[j][0], al[j][1], al[j][2], al[j][3],
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
void fm::fill_round_f32(RoundF32* r, const Bank& q, const Bank& t)
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
    int64_t pairs = 0, rows_read = 0;
    for (int64_t b = 0; b < n_rounds; ++b) {
        const int64_t nq = q_off[b + 1] - q_off[b], nt = t_off[b + 1] - t_off[b];
        if (nq < 0 || nt < 0 || t_off[b] < 0 || t_off[b + 1] > t->n)
            return fail(ctx, FM_EINVAL, "fm_xcheck1_batched: bad round offsets");
        if (nq > round_qcap())
            return fail(ctx, FM_EUNSUPPORTED, "fm_xcheck1_batched: a round has more than 4096 query rows; use fm_xcheck1 on gathered banks");
        pairs += nq * nt;
        rows_read += nq + nt;
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
    ctx->pending_bytes += rows_read * (f32 ? 512 : 128);
    HIP_TRY(ctx, d2h(ctx, tidx, ob + o_tidx, (size_t)tot * 4));
    HIP_TRY(ctx, d2h(ctx, dist, ob + o_dist, (size_t)tot * 4));
    if (ratio) HIP_TRY(ctx, d2h(ctx, ratio, ob + o_ratio, (size_t)tot * 8));
    return cs.finish();
}
