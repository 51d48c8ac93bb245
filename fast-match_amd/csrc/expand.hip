// K7 -- device-resident expansion loop: fastmatch.pyx:56-103 (do_iter + get_neighbors) and
// :145-169 (match_position) for a pre-extracted image pair, one persistent workgroup per
// pair, no host round trip between rounds.
//
// The loop is order dependent (neighbours are PREPENDED: depth first; the query subset of a
// (cell, query-cell) key is centred on the first seed that reaches it -- SURVEY.md fact 9),
// so one workgroup replays it sequentially in exactly the reference's order; parallelism is
// inside a round (radius query, sort, MFMA cross-check, hash probes) and across independent
// pairs (one workgroup each).  Per round:
//   1. pop the next (query_pos, target_pos) -- pending stack first, then the seed list;
//      skip it if its key (col,row,qcol,qrow) was already matched        fastmatch.pyx:68-72
//   2. radius query on the query keypoints around the truncated position, sorted by
//      (dx^2+dy^2 in float64, index), boundary inclusive                 cache.pyx:173-188
//   3. cross-checked 1-NN of that subset against the cell's descriptors  fastmatch.pyx:161-162
//      (x1_round: int8 MFMA), ratio = float64(dist)/selfdist             fastmatch.pyx:165
//   4. accepted = ratio < tau; for every accepted match the 4-neighbour cell on the side its
//      target point lies in (Grid_Cache.get_neighbor, cache.pyx:72-92); pushed so that the
//      first accepted match's neighbour is visited next                  fastmatch.pyx:75-77
//   5. accepted matches not seen before under (ratio, int-truncated positions) are appended
//      to the output in order                                            fastmatch.pyx:82-86
// Two reductions that cannot change the outcome keep the stack short: a neighbour whose key
// is already matched when it is pushed, or equals the key of an earlier neighbour of the same
// round, would be skipped when popped anyway, so it is not pushed.
#include "round_body.h"
#include <type_traits>
#include "expand_pair.h"

namespace fm {

constexpr int kExpThreads = 512;          // one workgroup of 8 waves per image pair (two waves per SIMD: the round is
                                          // a chain of short latency-bound steps, a second wave hides part of each)
constexpr int kExpWaves = kExpThreads / 64;
constexpr int kExpCand = 2048;            // radius-subset capacity per round (the kernel every pair starts with)
constexpr int kExpCandBig = 4096;         // ... of the kernel a pair is re-run with when a round exceeds it (int8 banks)

// Capacities of one kernel variant.  LDS: gather stage | keys/qbest u64[CAND] | cand i32[CAND] |
// nkey u64[CAND] | tix i32[CAND] | hist | tbest | the cell's first 128 train rows (x1_stage_cell; behind the cross-check the
// same 16 KiB hold the duplicate tables of steps 4/5, dup_insert).  The big variant pays for its 4096-row arrays with a
// 256-row gather stage (two gather steps for a typical round) and keeps almost no match positions in LDS.
template <int CAND>
struct ExpCfg {
    static constexpr int kCand = CAND;
    static constexpr int kSlotBits = CAND <= 2048 ? 11 : 12;                 // tix[] = slot | t_local << kSlotBits
    static constexpr int kSR = CAND <= 2048 ? 512 : 256;                     // query rows gathered per staging step
    static constexpr int kStageBytes = kSR * kDim + kSR / 32 * 256;
    static constexpr int kPosCap = (kStageBytes - CAND * 8) / 32;            // accepted matches whose positions are kept in LDS
    static constexpr int kLdsBytes = kStageBytes + CAND * (8 + 8 + 4 + 4) + (2 * 1024 + 16) * 4 + kTbestWords * 8 + kCellStageBytes;
    // float32 route: candidate list of x1_round_f32 aliases the sort scratch (nkey + tix + hist), free during step 3
    static constexpr int kClistCap = (CAND * (8 + 4) + 2 * 1024 * 4) / 4;
    static_assert((1 << kSlotBits) >= CAND, "slot bits");
    static_assert(kPosCap >= 0 && CAND * 8 <= kStageBytes, "rk[] must fit the stage buffer");
    static_assert(kLdsBytes <= 160 * 1024, "LDS of one CU");
};

enum { kExpOk = 0, kExpStackFull = 1, kExpCandFull = 2, kExpOutOfBounds = 3, kExpMatchFull = 4, kExpTableFull = 5,
       kExpListFull = 6, kExpNeedCell = 7, kExpNeedXcheck = 8, kExpLogFull = 9 };
static_assert(kRF_StageBytes <= ExpCfg<kExpCand>::kStageBytes, "the float32 round's gather image must fit the stage buffer");

__device__ __forceinline__ unsigned long long mix64(unsigned long long x)
{
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}

__device__ __forceinline__ unsigned long long pack4x16(int a, int b, int c, int d)
{
    return ((unsigned long long)(unsigned short)a << 48) | ((unsigned long long)(unsigned short)b << 32) |
           ((unsigned long long)(unsigned short)c << 16) | (unsigned long long)(unsigned short)d;
}

__device__ __forceinline__ bool set_contains(gptr<const unsigned long long> tab, long long cap, unsigned long long key)
{
    long long p = (long long)(mix64(key) & (unsigned long long)(cap - 1));
    for (long long n = 0; n < cap; ++n) {
        const unsigned long long v = tab[p];
        if (v == key) return true;
        if (v == ~0ull) return false;
        p = (p + 1) & (cap - 1);
    }
    return false;
}

// Single-writer insert (thread 0 only).  Returns false when the table is full.
__device__ __forceinline__ bool set_insert(gptr<unsigned long long> tab, long long cap, unsigned long long key)
{
    long long p = (long long)(mix64(key) & (unsigned long long)(cap - 1));
    for (long long n = 0; n < cap; ++n) {
        const unsigned long long v = tab[p];
        if (v == key) return true;
        if (v == ~0ull) { tab[p] = key; return true; }
        p = (p + 1) & (cap - 1);
    }
    return false;
}

__device__ __forceinline__ bool found_contains(gptr<const unsigned long long> tab, long long cap,
                                               unsigned long long k0, unsigned long long k1)
{
    long long p = (long long)(mix64(k0 ^ mix64(k1)) & (unsigned long long)(cap - 1));
    for (long long n = 0; n < cap; ++n) {
        const unsigned long long v0 = tab[2 * p];
        if (v0 == ~0ull) return false;
        if (v0 == k0 && tab[2 * p + 1] == k1) return true;
        p = (p + 1) & (cap - 1);
    }
    return false;
}

// Both probes of an accepted match in one go: is the neighbour key `nk` in the seen set (and if
// not, which empty slot ended its probe), and is (k0, k1) in the found set.  The first loads of
// the two probes are issued together (two dependent memory round trips would otherwise follow
// each other); collisions continue one probe at a time.
__device__ __forceinline__ void probe_both(gptr<const unsigned long long> seen, long long seen_cap, unsigned long long nk,
                                           gptr<const unsigned long long> found, long long found_cap,
                                           unsigned long long k0, unsigned long long k1,
                                           bool* in_seen, long long* seen_slot, bool* in_found)
{
    long long p1 = (long long)(mix64(nk) & (unsigned long long)(seen_cap - 1));
    long long p2 = (long long)(mix64(k0 ^ mix64(k1)) & (unsigned long long)(found_cap - 1));
    const bool want_seen = nk != ~0ull;
    unsigned long long v1 = want_seen ? seen[p1] : ~0ull;
    unsigned long long w0 = found[2 * p2], w1 = found[2 * p2 + 1];
    bool hit = false;
    if (want_seen) {
        for (long long n = 0; n < seen_cap; ++n) {
            if (v1 == nk) { hit = true; break; }
            if (v1 == ~0ull) break;
            p1 = (p1 + 1) & (seen_cap - 1);
            v1 = seen[p1];
        }
    }
    *in_seen = hit;
    *seen_slot = p1;
    bool fhit = false;
    for (long long n = 0; n < found_cap; ++n) {
        if (w0 == ~0ull) break;
        if (w0 == k0 && w1 == k1) { fhit = true; break; }
        p2 = (p2 + 1) & (found_cap - 1);
        w0 = found[2 * p2]; w1 = found[2 * p2 + 1];
    }
    *in_found = fhit;
}

// Concurrent insert of keys known to be absent and mutually distinct (claim an empty slot).
__device__ __forceinline__ bool found_insert(gptr<unsigned long long> tab, long long cap,
                                             unsigned long long k0, unsigned long long k1)
{
    long long p = (long long)(mix64(k0 ^ mix64(k1)) & (unsigned long long)(cap - 1));
    for (long long n = 0; n < cap; ++n) {
        unsigned long long expect = ~0ull;
        if (__hip_atomic_compare_exchange_strong(&tab[2 * p], &expect, k0, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { tab[2 * p + 1] = k1; return true; }
        p = (p + 1) & (cap - 1);
    }
    return false;
}

// Duplicates inside one round's list of accepted matches, by a hash table in LDS: own[] = the entry that claimed a
// slot (-1 = empty), mn[] = the lowest entry index of that slot's key.  All entries insert at once; entry k is a
// duplicate of an earlier one iff mn[slot] != k after the barrier.  Equal keys probe the same sequence, so they meet
// in the first slot one of them claimed, whatever the order of the claims: the result does not depend on the race.
// No more than kDupMax entries (half the slots) may insert.
constexpr int kDupSlots = 1024, kDupMax = 512;
static_assert(4 * kDupSlots * 4 <= kCellStageBytes, "the four tables live in the train-cell stage, which is idle behind the cross-check");
template <class SAME>
__device__ __forceinline__ int dup_insert(int* own, int* mn, unsigned long long h, int k, SAME same_key)
{
    int p = (int)(h & (unsigned long long)(kDupSlots - 1));
    for (;;) {
        const int old = atomicCAS(&own[p], -1, k);
        if (old == -1 || same_key(old)) break;
        p = (p + 1) & (kDupSlots - 1);
    }
    atomicMin(&mn[p], k);
    return p;
}

// Single-writer insert-if-absent in one probe sequence: 1 = inserted, 0 = was present, -1 = full.
__device__ __forceinline__ int set_insert_new(gptr<unsigned long long> tab, long long cap, unsigned long long key)
{
    long long p = (long long)(mix64(key) & (unsigned long long)(cap - 1));
    for (long long n = 0; n < cap; ++n) {
        const unsigned long long v = tab[p];
        if (v == key) return 0;
        if (v == ~0ull) { tab[p] = key; return 1; }
        p = (p + 1) & (cap - 1);
    }
    return -1;
}

// Grid_Cache geometry (cache.pyx:95-99, 116-121, 72-92) in the reference's arithmetic:
// float64 division / multiplication, int() truncation toward zero.
__device__ __forceinline__ int blk(double v, int cell) { return (int)(v / (double)cell); }
__device__ __forceinline__ int center_coord(int i, int cell, int limit)
{
    const int c = (int)(((double)i + 0.5) * (double)cell);
    return c < limit - 1 ? c : limit - 1;
}

// Sort n <= CAND (key, idx) pairs held in LDS ascending by (key, idx); pairs are unique.
// keys are the bit patterns of squared distances in [0, r2]: inside a disc they are spread
// uniformly, so a counting sort over kSortBuckets linear buckets leaves ~n / kSortBuckets
// elements per bucket and the exact order inside a bucket is fixed by a handful of
// comparisons.  bucket(d2) is monotone in d2, so bucket order + in-bucket order = total order.
// Scratch: k2 (u64[CAND]), i2 (int[CAND]), hist (int[2 * kSortBuckets + 8]).
constexpr int kSortBuckets = 1024;
__device__ __forceinline__ int block_exclusive_scan_nosync(int v, int* my_offset, int* wave_tot);

// Clears the counting-sort buckets; a barrier must lie between this and block_sort_pairs.
__device__ __forceinline__ void block_sort_clear(int* hist)
{
    for (int b = threadIdx.x; b < kSortBuckets; b += kExpThreads) { hist[b] = 0; hist[kSortBuckets + 4 + b] = 0; }
}

// lo: the keys lie in [lo, r2] (a chunk of a huge round: the buckets then spread over that range only).
template <int CAND>
__device__ __forceinline__ void block_sort_pairs(unsigned long long* keys, int* idx, unsigned long long* k2,
                                                 int* i2, int* hist, int n, double r2, double lo = 0.0)
{
    const int tid = threadIdx.x;
    int* start = hist;                       // [kSortBuckets + 1] after the scan
    int* cursor = hist + kSortBuckets + 4;   // [kSortBuckets]
    // (the caller has cleared start[] and cursor[] -- block_sort_clear -- in front of a barrier it needs anyway)
    const double scale = r2 > lo ? (double)kSortBuckets / (r2 - lo) : 0.0;
    int myb[CAND / kExpThreads];
#pragma unroll
    for (int s = 0; s < CAND / kExpThreads; ++s) {
        const int i = s * kExpThreads + tid;
        myb[s] = 0;
        if (i < n) {
            const double d2 = __longlong_as_double((long long)keys[i]);
            int b = (int)((d2 - lo) * scale);           // (monotone in d2; a value just below lo truncates to bucket 0)
            b = b < kSortBuckets - 1 ? b : kSortBuckets - 1;
            myb[s] = b;
            atomicAdd(&start[b], 1);
        }
    }
    lds_barrier();
    // exclusive scan of the bucket counts: kSortBuckets / kExpThreads buckets per thread
    {
        constexpr int kPer = kSortBuckets / kExpThreads;
        int c[kPer], s = 0;
#pragma unroll
        for (int q = 0; q < kPer; ++q) { c[q] = start[tid * kPer + q]; s += c[q]; }
        int off;
        block_exclusive_scan_nosync(s, &off, hist + 2 * kSortBuckets + 8 /* tail: wave totals */);
#pragma unroll
        for (int q = 0; q < kPer; ++q) { start[tid * kPer + q] = off; off += c[q]; }
        if (tid == kExpThreads - 1) start[kSortBuckets] = off;
    }
    lds_barrier();
#pragma unroll
    for (int s = 0; s < CAND / kExpThreads; ++s) {
        const int i = s * kExpThreads + tid;
        if (i < n) {
            const int p = start[myb[s]] + atomicAdd(&cursor[myb[s]], 1);
            k2[p] = keys[i];
            i2[p] = idx[i] | 0;          // (bucket order; order inside a bucket is arbitrary here)
        }
    }
    lds_barrier();
    // final position = bucket start + number of smaller pairs inside the bucket
#pragma unroll
    for (int s = 0; s < CAND / kExpThreads; ++s) {
        const int p = s * kExpThreads + tid;
        if (p < n) {
            const unsigned long long k = k2[p];
            const int v = i2[p];
            double d2 = __longlong_as_double((long long)k);
            int b = (int)((d2 - lo) * scale);
            b = b < kSortBuckets - 1 ? b : kSortBuckets - 1;
            const int s0 = start[b], s1 = start[b + 1];
            int rank = s0;
            for (int j = s0; j < s1; ++j) {
                const unsigned long long kj = k2[j];
                rank += (kj < k || (kj == k && i2[j] < v)) ? 1 : 0;
            }
            keys[rank] = k;
            idx[rank] = v;
        }
    }
    lds_barrier();
}

// Exclusive scan of per-thread counts over the workgroup; returns the total.
__device__ __forceinline__ int block_exclusive_scan(int v, int* my_offset, int* wave_tot /*LDS[kExpWaves]*/);
// The same without the leading barrier: for a caller that has just passed one and whose wave_tot[] nobody reads any more.
__device__ __forceinline__ int block_exclusive_scan_nosync(int v, int* my_offset, int* wave_tot)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wave_tot[wave] = inc;
    lds_barrier();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kExpWaves; ++w) { if (w < wave) base += wave_tot[w]; tot += wave_tot[w]; }
    *my_offset = base + inc - v;
    return tot;
}
__device__ __forceinline__ int block_exclusive_scan(int v, int* my_offset, int* wave_tot /*LDS[kExpWaves]*/)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    lds_barrier();
    if (lane == 63) wave_tot[wave] = inc;
    lds_barrier();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kExpWaves; ++w) { if (w < wave) base += wave_tot[w]; tot += wave_tot[w]; }
    *my_offset = base + inc - v;
    return tot;
}

// Exclusive ranks of up to two boolean flags per thread over the workgroup, and their
// totals (a in the low half-word, b in the high one).  Ballots + population counts instead of a
// shuffle scan, and ONE barrier: consecutive calls alternate between two LDS buffers, so the
// barrier of call k also separates the reads of call k - 1 from the writes of call k + 1.
__device__ __forceinline__ int block_rank_flags(bool a, bool b, int* rank_a, int* rank_b, int (*wave_cnt)[kExpWaves], int& toggle)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long ma = __builtin_amdgcn_ballot_w64(a), mb = __builtin_amdgcn_ballot_w64(b);
    const unsigned long long below = (1ull << lane) - 1ull;
    int* buf = wave_cnt[toggle];
    toggle ^= 1;
    if (lane == 0) buf[wave] = __popcll(ma) | (__popcll(mb) << 16);
    lds_barrier();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kExpWaves; ++w) { const int c = buf[w]; if (w < wave) base += c; tot += c; }
    *rank_a = (base & 0xffff) + __popcll(ma & below);
    *rank_b = (base >> 16) + __popcll(mb & below);
    return tot;
}

// F32: the pairs of the launch hold float32 banks (float32 round) -- a kernel of its own, so that
// the int8 kernel does not carry the float32 round's registers (inlined together they spill).
// CAND: capacity variant (ExpCfg); the big one exists for int8 banks only.
// HUGE (int8 banks without the float32-root guard, float32 banks): a round whose radius subset exceeds CAND rows -- denser keypoints than a
// uniform image has, or a larger `radius` option; the reference has no limit (cache.pyx:173-188) -- is processed in
// chunks of at most CAND rows instead of ending the run: the subset's histogram over the sort's buckets cuts it into
// ranges of the sort key, every range is selected by a radius query of its own, sorted (ranges are disjoint and
// ascending, so chunk order + order inside the chunk = the order of the whole subset) and matched against the cell
// with the per-train-row minimum carried from chunk to chunk in global memory (x1_round_wsplit, MERGE); the election
// and steps 4 / 5 then read the whole subset's tables from global memory.  Rounds that fit run the code of the
// other variants unchanged.  Still given up (status 2 -> host loop): more than kHugeChunks chunks.  (r05: a single
// bucket of more than CAND keypoints -- thousands at one distance -- is split by rank, huge_rank_pivots; pairs under the
// float32-root guard have their ties repaired at the election; more than CAND ACCEPTED matches in one round are taken in
// blocks of the slot range, steps 4 / 5 below.)
constexpr int kHugeChunks = 640;

// first pivot c in [1, nch) whose value, with the trial bit or'ed in, exceeds the entry (nch: none); the pivots are ordered
__device__ __forceinline__ int rank_first_above(const unsigned long long* vk, const unsigned* vi, int nch, unsigned long long k, unsigned qi,
                                                unsigned long long kbit, unsigned ibit)
{
    int lo = 1, hi = nch;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const unsigned long long tk = vk[mid] | kbit;
        const unsigned ti = vi[mid] | ibit;
        if (k < tk || (k == tk && qi < ti)) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// Chunks of a huge round by RANK (expand_kernel, HUGE; r05): pivot c = the (key bits, keypoint index) pair of rank c * cand among
// the nq entries (h_keys[i], h_idx[i]), for c in [1, nch); chrow[c] = c * cand.  Built bit by bit from the top -- 63 key bits,
// then 31 index bits: a bit stays if no more than c * cand entries lie below the value with it -- one counting pass over the
// list per bit for all pivots at once (they stay ordered, so an entry finds the first pivot above it by bisection and the
// counts are a running sum).  ~95 passes: hundreds of microseconds for a round no image of keypoints produces.  A function of
// its own (not inlined): the cold path keeps its registers out of the kernel's allocation.  vk / vi / vcnt: LDS, nch + 1 words each.
__device__ __attribute__((noinline)) void huge_rank_pivots(unsigned long long* vk, unsigned* vi, int* vcnt, int* chrow, int nq, int nch, int cand,
                                                           const unsigned long long* h_keys, const int* h_idx)
{
    const int tid = threadIdx.x;
    for (int c = tid; c <= nch; c += kExpThreads) { vk[c] = 0ull; vi[c] = 0u; chrow[c] = c < nch ? c * cand : nq; }
    for (int bit = 93; bit >= 0; --bit) {
        const unsigned long long kbit = bit >= 31 ? 1ull << (bit - 31) : 0ull;
        const unsigned ibit = bit >= 31 ? 0u : 1u << bit;
        for (int c = tid; c <= nch; c += kExpThreads) vcnt[c] = 0;
        __syncthreads();
        for (int i = tid; i < nq; i += kExpThreads) {
            const unsigned long long k = __hip_atomic_load(h_keys + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned qi = (unsigned)__hip_atomic_load(h_idx + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicAdd(&vcnt[rank_first_above(vk, vi, nch, k, qi, kbit, ibit)], 1);
        }
        __syncthreads();
        for (int c = 1 + tid; c < nch; c += kExpThreads) {
            int below = 0;
            for (int j = 1; j <= c; ++j) below += vcnt[j];
            if (below <= c * cand) { vk[c] |= kbit; vi[c] |= ibit; }
        }
        __syncthreads();
    }
}
// LAZY (r04; the reference's own mode, cache.pyx:102-106, 124-138: a cell's features are computed -- SIFT on the crop -- when the
// loop first reaches the cell): cells carry a `ready` flag and their own (first row, row count) in a target bank that grows as
// the host adds cells.  A round that needs a cell which is not there SAVES the loop's state (counters, pending-stack height, seed
// cursor, the popped entry -- its key is already in the seen set) and ends the launch with status kExpNeedCell + the cell id; the host
// computes the cell (fm_bank_append_u8 + fm_expand_set_cell) and launches again with `resume`, which restores the state and takes the
// saved entry instead of popping one.  Everything between two such stops runs at the device loop's pace.
template <bool F32, int CAND = kExpCand, bool HUGE = false, bool LAZY = false>
__global__ __launch_bounds__(kExpThreads)
void expand_kernel(const ExpandPair* __restrict__ pairs)
{
    static_assert(!LAZY || (HUGE && CAND == kExpCand), "the lazy-target variants are built on the chunked kernels");
    using C = ExpCfg<CAND>;
    static_assert(!F32 || CAND == kExpCand, "the float32 round needs the 512-row stage buffer");
    static_assert(!HUGE || CAND == kExpCand, "the chunked round exists for the first capacity variant");
    // dynamic LDS (C::kLdsBytes): the gather stage, the sort keys / qbest table, the
    // candidate rows, and two scratch arrays
    extern __shared__ __attribute__((aligned(16))) char dyn_lds[];
    char* smem = dyn_lds;                                                         // C::kStageBytes
    unsigned long long* keys = (unsigned long long*)(dyn_lds + C::kStageBytes);   // sort keys, then qbest
    int* cand = (int*)(keys + CAND);                                              // candidate / sorted query rows
    unsigned long long* nkey = (unsigned long long*)(cand + CAND);                // ratio bits of accepted matches
    int* tix  = (int*)(nkey + CAND);                                              // sort scratch, then accepted list
    int* hist = tix + CAND;                                                       // counting-sort buckets
    // (nkey, tix, hist are contiguous: the float32 round's candidate list aliases them during step 3)
    __shared__ double cur[4];                         // query_pos, target_pos of the round
    __shared__ int sh_i[8];
    __shared__ long long sh_top;
    __shared__ int wave_cnt[2][kExpWaves];
    int rank_toggle = 0;                              // (uniform)
    __shared__ int sh_rf[kExpWaves];                          // per-wave candidate counts of the float32 round
    // The entry a round pushes on TOP of the stack is, nine times out of ten, the next one popped.
    // The pushing thread leaves a copy here together with its key and the empty slot that ended the
    // key's seen-set probe, so the pop needs neither the stack read nor a probe of its own.
    __shared__ long long sh_seed;
    __shared__ unsigned long long sh_key;
    __shared__ int sh_first[kExpWaves];
    __shared__ double nxt_e[4];
    __shared__ unsigned long long nxt_key;
    __shared__ long long nxt_slot;
    __shared__ int nxt_valid;
    if (threadIdx.x == 0) nxt_valid = 0;

    // The pair descriptor is read from memory, so its pointers reach the compiler as GENERIC ones and
    // every access through them would be a FLAT instruction (see gptr in tile_ops.h).  P is a view of
    // the descriptor whose pointers are typed as global memory.
    const ExpandPair& M = pairs[blockIdx.x];
    struct View {
        gptr<const int8_t> q_rows8; gptr<const int32_t> q_norm; gptr<const double> q_selfdist, q_pos, q_pos_ord;
        gptr<const int32_t> idx_order, idx_start;
        double idx_bucket, idx_x0, idx_y0; int idx_nbx, idx_nby, metric;
        gptr<const int8_t> t_rows8; gptr<const int32_t> t_norm; gptr<const int64_t> cell_off; gptr<const double> t_pos;
        int width, height, cell_w, cell_h, rows, cols, margin, radius, f32, tie_guard;
        gptr<const double> seeds; int64_t n_seeds; double tau;
        gptr<double> stack; int64_t stack_cap;
        gptr<unsigned long long> seen; int64_t seen_cap;
        gptr<unsigned long long> found; int64_t found_cap;
        gptr<int32_t> m_index; gptr<double> m_pos, m_ratio; int64_t match_cap;
        gptr<long long> result; int prof;
        gptr<int32_t> h_cand, h_ucand; gptr<unsigned long long> h_qbest, h_tbest, h_pkey;
        gptr<const int64_t> cell_start; gptr<const int32_t> cell_cnt, cell_ready; gptr<long long> resume_state; int resume;
        long long delegate_min;
        gptr<long long> lg_round; gptr<int32_t> lg_q, lg_t; gptr<double> lg_ratio; long long lg_round_cap, lg_entry_cap;
    } P;
    P.q_rows8 = (gptr<const int8_t>)M.q_rows8; P.q_norm = (gptr<const int32_t>)M.q_norm;
    P.q_selfdist = (gptr<const double>)M.q_selfdist; P.q_pos = (gptr<const double>)M.q_pos; P.q_pos_ord = (gptr<const double>)M.q_pos_ord;
    P.idx_order = (gptr<const int32_t>)M.idx_order; P.idx_start = (gptr<const int32_t>)M.idx_start;
    P.idx_bucket = M.idx_bucket; P.idx_x0 = M.idx_x0; P.idx_y0 = M.idx_y0; P.idx_nbx = M.idx_nbx; P.idx_nby = M.idx_nby; P.metric = M.metric;
    P.t_rows8 = (gptr<const int8_t>)M.t_rows8; P.t_norm = (gptr<const int32_t>)M.t_norm;
    P.cell_off = (gptr<const int64_t>)M.cell_off; P.t_pos = (gptr<const double>)M.t_pos;
    P.width = M.width; P.height = M.height; P.cell_w = M.cell_w; P.cell_h = M.cell_h;
    P.rows = M.rows; P.cols = M.cols; P.margin = M.margin; P.radius = M.radius; P.f32 = M.f32; P.tie_guard = M.tie_guard;
    P.seeds = (gptr<const double>)M.seeds; P.n_seeds = M.n_seeds; P.tau = M.tau;
    P.stack = (gptr<double>)M.stack; P.stack_cap = M.stack_cap;
    P.seen = (gptr<unsigned long long>)M.seen; P.seen_cap = M.seen_cap;
    P.found = (gptr<unsigned long long>)M.found; P.found_cap = M.found_cap;
    P.m_index = (gptr<int32_t>)M.m_index; P.m_pos = (gptr<double>)M.m_pos; P.m_ratio = (gptr<double>)M.m_ratio;
    P.match_cap = M.match_cap; P.result = (gptr<long long>)M.result; P.prof = M.prof;
    P.h_cand = (gptr<int32_t>)M.h_cand; P.h_ucand = (gptr<int32_t>)M.h_ucand; P.h_qbest = (gptr<unsigned long long>)M.h_qbest; P.h_tbest = (gptr<unsigned long long>)M.h_tbest; P.h_pkey = (gptr<unsigned long long>)M.h_pkey;
    P.cell_start = (gptr<const int64_t>)M.cell_start; P.cell_cnt = (gptr<const int32_t>)M.cell_cnt; P.cell_ready = (gptr<const int32_t>)M.cell_ready;
    P.resume_state = (gptr<long long>)M.resume_state; P.resume = M.resume; P.delegate_min = M.delegate_min;
    P.lg_round = (gptr<long long>)M.lg_round; P.lg_q = (gptr<int32_t>)M.lg_q; P.lg_t = (gptr<int32_t>)M.lg_t; P.lg_ratio = (gptr<double>)M.lg_ratio;
    P.lg_round_cap = M.lg_round_cap; P.lg_entry_cap = M.lg_entry_cap;
    const RoundF32G RF(M.rf);
    int tid = threadIdx.x;

    long long top = 0;            // stack height: owned by thread 0, published in sh_top each round
    long long seed_i = 0;
    long long n_matches = 0, n_rounds = 0, n_pairs = 0;
    long long seen_n = 0;
    int status = kExpOk;
    // phase timers (expand_prof): thread 0's alone, kept in LDS so that they hold no registers; pt[12] = the last stamp
    __shared__ long long pt[13];
    long long& tstamp = pt[12];
    if (tid == 0) { for (int k = 0; k < 12; ++k) pt[k] = 0; tstamp = P.prof ? wall_clock64() : 0; }
#define EXP_STAMP(k) do { if (P.prof && tid == 0) { const long long _n = wall_clock64(); pt[k] += _n - tstamp; tstamp = _n; } } while (0)
    // FM_PARK_PROF builds only (make FLAGS_expand=-DFM_PARK_PROF; scripts/README.md): where a round beyond the LDS tables spends
    // its time -- thread 0's wall clock per phase of such rounds, summed over the launches of a run into result[8 ..]
    // (0 radius walk, 1 list + histogram + chunk bounds, 2 partition, 3 chunk sorts (+ cross-check when not delegated),
    // 4 step 4, 5 step 5, 6 everything in rounds that fit, 7 number of such rounds)
#ifdef FM_PARK_PROF
    __shared__ long long hp[8];
    __shared__ long long hstamp;
    if (tid == 0) { for (int k = 0; k < 8; ++k) hp[k] = 0; hstamp = wall_clock64(); }
#define HUGE_STAMP(k) do { if (tid == 0) { const long long _n = wall_clock64(); hp[k] += _n - hstamp; hstamp = _n; } } while (0)
#else
#define HUGE_STAMP(k) do { } while (0)
#endif

    // per-round log: entries written so far (thread 0 updates it at the end of a round, everyone reads it behind barriers)
    __shared__ long long sh_nlog;
    __shared__ int sh_anyx;       // logging runs: some pair of the round in progress passed the cross-check
    if (tid == 0) { sh_nlog = 0; sh_anyx = 0; }
    // the record of the round in progress (thread 0): its entry, the cell it fetched, its accepted matches.
    // n_acc: -1 = a cell without features, -2 = no cross-checked pair at all (an empty radius subset included) -- in both
    // cases the reference's match_position returns arrays of shape (0,), not (0, 2, 2) (fastmatch.pyx:155-156, 162-167:
    // numpy.array([]) of an empty list), and log_round keeps that shape
    auto log_header = [&](int cell_id, int n_acc) {
        gptr<long long> rec = P.lg_round + (n_rounds - 1) * 6;
        for (int k = 0; k < 4; ++k) rec[k] = __double_as_longlong(cur[k]);
        rec[4] = cell_id; rec[5] = (n_acc == 0 && sh_anyx == 0) ? -2 : n_acc;
        sh_anyx = 0;
    };
    bool skip_pop = false;        // (uniform) the first round of a resumed run takes the saved entry
    bool skip_x = false;          // (uniform) ... and, resumed behind a DELEGATED cross-check (P.resume == 2), goes straight to steps 4 / 5
    int need_cell = -1;
    if constexpr (LAZY || HUGE) {
        if (P.resume) {
            top = P.resume_state[0]; seed_i = P.resume_state[1]; n_matches = P.resume_state[2];
            n_rounds = P.resume_state[3]; n_pairs = P.resume_state[4]; seen_n = P.resume_state[5];
            if (tid == 0) { for (int k = 0; k < 4; ++k) cur[k] = __longlong_as_double(P.resume_state[6 + k]); sh_nlog = P.resume_state[13]; }
            skip_pop = true;
            skip_x = P.resume == 2;
        }
    }

    for (;;) {
        // (the chunked / lazy variants sit at the 256-register ceiling: with the thread number opaque per round, the address
        // arithmetic on it is redone where it is used instead of being hoisted out of the loop and parked in scratch)
#ifndef FM_K7_NO_OPAQUE_TID         // (A/B builds: scripts/README.md)
        if constexpr (HUGE || LAZY || F32) asm volatile("" : "+v"(tid));
#endif
        HUGE_STAMP(6);
        // ---- 1. next unseen (query_pos, target_pos) -----------------------------------------
        if ((LAZY || HUGE) && skip_pop) {
            skip_pop = false;
            if (tid == 0) {
                nxt_valid = 0;
                sh_i[0] = 1; sh_i[1] = blk(cur[3], P.cell_h); sh_i[2] = blk(cur[2], P.cell_w);
                sh_i[3] = status; sh_i[4] = 0; sh_i[7] = 0;
                sh_top = top; sh_seed = seed_i;
            }
            lds_barrier();
        } else {
        if (tid == 0) {
            int have = 0;
            if (nxt_valid && top > 0 && 2 * (seen_n + 1) <= P.seen_cap) {
                // the entry on top is the one the previous round pushed last: it is cached, its key is
                // known to be new (checked when it was pushed; nothing was inserted since) and the slot
                // that ended that probe is still empty
                --top;
                P.seen[nxt_slot] = nxt_key;
                ++seen_n;
                for (int k = 0; k < 4; ++k) cur[k] = nxt_e[k];
                sh_i[1] = blk(nxt_e[3], P.cell_h); sh_i[2] = blk(nxt_e[2], P.cell_w);
                have = 1;
            }
            nxt_valid = 0;
            sh_i[0] = have;
            sh_i[3] = status;
            sh_i[4] = 0;              // the radius query's candidate counter (read long before this point, by everyone)
            sh_i[7] = 0;
            sh_top = top;
            sh_seed = seed_i;
            EXP_STAMP(7);
        }
        lds_barrier();
        // Not the cached entry: entries whose key was matched since they were pushed are skipped
        // (fastmatch.pyx:71-72) -- on average four stale entries per round, each a dependent
        // stack read + hash probe if one thread pops them one by one.  Instead the workgroup looks at
        // the next kExpThreads entries of the source (stack from the top, then the seed list in order) at
        // once; the first one with an unseen key is the round's entry, everything before it is
        // stale and dropped, everything after it stays where it is.
        while (!sh_i[0] && sh_i[3] == kExpOk) {
            const long long t = sh_top, si = sh_seed;
            const bool from_stack = t > 0;
            const long long avail = from_stack ? t : (P.n_seeds - si);
            if (avail <= 0) break;
            const int w = (int)(avail < kExpThreads ? avail : kExpThreads);
            bool unseen = false;
            double e[4] = {0, 0, 0, 0};
            unsigned long long key = 0;
            int ecol = 0, erow = 0;
            if (tid < w) {
                const long long src = from_stack ? (t - 1 - tid) : (si + tid);
                gptr<const double> ep = from_stack ? (gptr<const double>)(P.stack + src * 4) : (P.seeds + src * 4);
                for (int k = 0; k < 4; ++k) e[k] = ep[k];
                ecol = blk(e[3], P.cell_h); erow = blk(e[2], P.cell_w);
                key = pack4x16(ecol, erow, blk(e[1], P.cell_h), blk(e[0], P.cell_w));
                unseen = !set_contains((gptr<const unsigned long long>)P.seen, P.seen_cap, key);
            }
            const unsigned long long um = __builtin_amdgcn_ballot_w64(unseen);
            if ((tid & 63) == 0) sh_first[tid >> 6] = um ? (tid + (int)__builtin_ctzll(um)) : 1 << 20;
            lds_barrier();
            int first = sh_first[0];
#pragma unroll
            for (int q = 1; q < kExpWaves; ++q) first = min(first, sh_first[q]);
            if (first < w && tid == first) {
                for (int k = 0; k < 4; ++k) cur[k] = e[k];
                sh_i[1] = ecol; sh_i[2] = erow;
                sh_key = key;
            }
            lds_barrier();
            if (tid == 0) {
                if (first < w) {
                    if (from_stack) top = t - first - 1; else seed_i = si + first + 1;
                    if (2 * (seen_n + 1) > P.seen_cap) status = kExpTableFull;            // a NEW key needs room
                    else {
                        const int ins = set_insert_new(P.seen, P.seen_cap, sh_key);
                        if (ins < 0) status = kExpTableFull;
                        else { ++seen_n; sh_i[0] = 1; }
                    }
                } else {
                    if (from_stack) top = t - w; else seed_i = si + w;
                }
                sh_i[3] = status;
                sh_top = top;
                sh_seed = seed_i;
            }
            lds_barrier();
        }
        }
        // (the barrier behind the pop block, or the one that ends the last pass of the loop, is in front of these reads)
        status = sh_i[3];
        if (!sh_i[0] || status != kExpOk) break;
        const int col = sh_i[1], row = sh_i[2];
        const long long sh_i_top_before = sh_top;
        EXP_STAMP(0);
        // C-int truncation of the positions (fastmatch.pyx:147-150)
        const int qx = (int)cur[0], qy = (int)cur[1], tx = (int)cur[2], ty = (int)cur[3];
        if (tx > P.width || ty > P.height) { status = kExpOutOfBounds; break; }     // cache.pyx:56-57
        // the cell actually fetched is the one of the TRUNCATED target position (target.get)
        const int gcol = blk((double)ty, P.cell_h), grow = blk((double)tx, P.cell_w);
        // the cell's row range is needed only after the radius query and the sort: fetch it now, so the
        // load's latency is not on the critical path in front of the cross-check
        const int cell = gcol * P.rows + grow;
        int64_t t0, t1;
        if constexpr (LAZY) {
            t0 = P.cell_start[cell];
            t1 = t0 + P.cell_cnt[cell];
            if (P.cell_ready[cell] == 0) {             // (uniform) not computed yet: park the round and hand back to the host
                if (tid == 0) {
                    P.resume_state[0] = top; P.resume_state[1] = seed_i; P.resume_state[2] = n_matches;
                    P.resume_state[3] = n_rounds; P.resume_state[4] = n_pairs; P.resume_state[5] = seen_n;
                    for (int k = 0; k < 4; ++k) P.resume_state[6 + k] = __double_as_longlong(cur[k]);
                    P.resume_state[13] = sh_nlog;
                }
                need_cell = cell;
                status = kExpNeedCell;
                break;
            }
        } else {
            t0 = P.cell_off[cell];
            t1 = P.cell_off[cell + 1];
        }
        int nq = 0;
        bool huge_round = false;                            // (uniform)
        bool fkeys = false;                                 // (uniform) h_qbest holds float32 distance bits (a delegated cross-check's keys)
        char* const cell_lds = dyn_lds + C::kLdsBytes - kCellStageBytes;
        const int nt = (int)(t1 - t0);
        if (HUGE && skip_x) {
            // resumed behind a delegated cross-check: the subset is in h_cand[], the host's dense kernels left the round's
            // cross-checked keys in h_qbest[]; the round goes on with steps 4 / 5
            skip_x = false;
            nq = (int)P.resume_state[10];
            huge_round = true;
            fkeys = true;
        } else {
        ++n_rounds;
        if (P.lg_round && n_rounds > P.lg_round_cap) { status = kExpLogFull; break; }

        // ---- 2. radius query (Position_Index.radius) ------------------------------------------
        block_sort_clear(hist);          // (for the sort behind the radius query: the barrier in between is the query's own)
        {   // (the counter sh_i[4] was cleared with the pop)
            const double r = (double)P.radius, b = P.idx_bucket;
            int bx0 = (int)floor(((double)qx - r - P.idx_x0) / b), bx1 = (int)floor(((double)qx + r - P.idx_x0) / b);
            int by0 = (int)floor(((double)qy - r - P.idx_y0) / b), by1 = (int)floor(((double)qy + r - P.idx_y0) / b);
            bx0 = max(bx0, 0); by0 = max(by0, 0);
            bx1 = min(bx1, P.idx_nbx - 1); by1 = min(by1, P.idx_nby - 1);
            const double r2 = r * r;
            if (P.idx_nbx > 0 && bx1 >= bx0) {
                // The bucket rows' index ranges are fetched first (independent loads), then ONE flat
                // loop walks their concatenation: three dependent memory round trips in all (ranges,
                // keypoint index, position) instead of three per bucket row.  8 rows at a time.
                for (int byb = by0; byb <= by1; byb += 8) {
                    int rs[8], pre[9];
                    pre[0] = 0;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int by = byb + j;
                        int s0 = 0, e0 = 0;
                        if (by <= by1) { s0 = P.idx_start[by * P.idx_nbx + bx0]; e0 = P.idx_start[by * P.idx_nbx + bx1 + 1]; }
                        rs[j] = s0;
                        pre[j + 1] = pre[j] + (e0 - s0);
                    }
                    for (int f = tid; f < pre[8]; f += kExpThreads) {
                        int j = 0;
#pragma unroll
                        for (int q = 1; q < 8; ++q) j += (f >= pre[q]) ? 1 : 0;
                        int base = rs[0], off = pre[0];
#pragma unroll
                        for (int q = 1; q < 8; ++q) { base = (j == q) ? rs[q] : base; off = (j == q) ? pre[q] : off; }
                        // (the positions in index order: fetched beside the keypoint index, not behind it)
                        const int io = base + f - off;
                        const int qi = P.idx_order[io];
                        const double dx = P.q_pos_ord[2 * io] - (double)qx, dy = P.q_pos_ord[2 * io + 1] - (double)qy;
                        // sort key and inclusion test of the pair's metric (BallTree(positions, metric), cache.pyx:276):
                        // squared Euclidean distance against r^2 (no fma: NumPy order), or |dx| + |dy| /
                        // max(|dx|, |dy|) against r
                        double d2, lim;                              // (a uniform branch: one formula is executed)
                        if (P.metric == 0) { d2 = __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)); lim = r2; }
                        else if (P.metric == 1) { d2 = __dadd_rn(fabs(dx), fabs(dy)); lim = r; }
                        else { d2 = fmax(fabs(dx), fabs(dy)); lim = r; }
                        if (d2 <= lim) {
                            const int slot = atomicAdd(&sh_i[4], 1);
                            if (slot < CAND) { keys[slot] = (unsigned long long)__double_as_longlong(d2); cand[slot] = qi; }
                            else if constexpr (HUGE) {
                                // a subset beyond the LDS tables: the rest of the list goes to the run's global tables, so that the
                                // chunked round below reads the subset back instead of walking the index again per chunk
                                P.h_qbest[slot] = (unsigned long long)__double_as_longlong(d2);
                                P.h_ucand[slot] = qi;
                            }
                        }
                    }
                }
            }
        }
        lds_barrier();
        nq = sh_i[4];
        if (nq > CAND) {
            if (HUGE) huge_round = true;
            else { status = kExpCandFull; break; }
        }
#ifdef FM_PARK_PROF
        if (huge_round) { HUGE_STAMP(0); if (tid == 0) ++hp[7]; }
#endif
        EXP_STAMP(1);
        if (!huge_round) {
        // The cell's first 128 descriptor rows (int8 round: the MFMA B operand) go to LDS by DMA from here: the sort
        // touches no global memory, so their latency passes under it and no register waits for them (issued in
        // front of the radius query they only delayed its own loads -- vector-memory waits retire in order; loaded
        // into registers here they were spilled across the sort, which put the wait into the sort).
        if constexpr (!F32) { if (t1 > t0) x1_stage_cell<kExpThreads>(cell_lds, P.t_rows8, t0, (int)(t1 - t0)); }
        // sort by (key bits, index): non-negative doubles order like their bit patterns
        block_sort_pairs<CAND>(keys, cand, nkey, tix, hist, nq,
                               P.metric == 0 ? (double)P.radius * (double)P.radius : (double)P.radius);

        EXP_STAMP(2);
        // ---- 3. cross-checked 1-NN against the cell ---------------------------------------------
        if (nt == 0 || nq == 0) {                           // match_position returns empty arrays (the round is still logged;
            // -1: a cell without features, fastmatch.pyx:155-156, returns arrays of another shape than an empty subset does)
            if (P.lg_round && tid == 0) log_header(cell, nt == 0 ? -1 : 0);
            continue;
        }
        n_pairs += (long long)nq * nt;
        if constexpr (HUGE) {
            // a round whose subset fits LDS but whose CELL is large (thousands of train rows: a blob of keypoints) is
            // delegated like a chunked one: the sorted subset goes to h_cand[], the run parks (see the chunked branch)
            if (P.delegate_min > 0 && !P.tie_guard && (long long)nq * nt >= 2 * P.delegate_min) {
                for (int i = tid; i < nq; i += kExpThreads) P.h_cand[i] = cand[i];
                if (tid == 0) {
                    P.resume_state[0] = top; P.resume_state[1] = seed_i; P.resume_state[2] = n_matches;
                    P.resume_state[3] = n_rounds; P.resume_state[4] = n_pairs; P.resume_state[5] = seen_n;
                    for (int k = 0; k < 4; ++k) P.resume_state[6 + k] = __double_as_longlong(cur[k]);
                    P.resume_state[10] = nq; P.resume_state[11] = t0; P.resume_state[12] = nt; P.resume_state[13] = sh_nlog;
                }
                need_cell = cell;
                status = kExpNeedXcheck;
                break;
            }
        }
        for (int i = tid; i < nq; i += kExpThreads) keys[i] = ~0ull;     // keys[] becomes the qbest table
        if constexpr (F32) {
            // descriptors that are not integer valued: fp16 MFMA filter + exact float32 chain (round_body_f32.h)
            lds_barrier();
            const bool ok = x1_round_f32<kExpThreads>(RF, cand, nq, t0, nt, smem, keys, (unsigned*)nkey, C::kClistCap,
                                         (unsigned long long*)(hist + 2 * kSortBuckets + 16), sh_rf,
                                         P.prof ? pt : nullptr, &tstamp);
            if (!ok) { status = kExpListFull; break; }
        } else {
            x1_round_wsplit<C::kSR, kExpThreads>(P.q_rows8, P.q_norm, cand, nq, P.t_rows8, P.t_norm, t0, nt, smem, keys,
                                    (unsigned long long*)(hist + 2 * kSortBuckets + 16), P.tie_guard, P.prof ? pt : nullptr, &tstamp, cell_lds);
            // (nt > 0: the function's last chunk ends with a barrier behind its updates of keys[])
        }
        } else if constexpr (HUGE) {
            // ---- 2b / 3b. a radius subset beyond the LDS tables: chunks of the sort-key range -----------------
            if (nt == 0) { if (P.lg_round && tid == 0) log_header(cell, -1); continue; }
            n_pairs += (long long)nq * nt;
            __shared__ int chb[kHugeChunks + 1];            // chunk c = sort buckets [chb[c], chb[c + 1])
            __shared__ int chrow[kHugeChunks + 1];          // ... = slots [chrow[c], chrow[c + 1]) of the sorted subset
            const double rr = (double)P.radius;
            const double lim_all = P.metric == 0 ? rr * rr : rr;
            const double bscale = lim_all > 0.0 ? (double)kSortBuckets / lim_all : 0.0;
            auto bucket_of = [&](double d2) __attribute__((always_inline)) {
                const int b = (int)(d2 * bscale);
                return b < kSortBuckets - 1 ? b : kSortBuckets - 1;
            };
            // The subset as step 2's walk found it, (sort key, keypoint index) in no particular order: slots [0, CAND) are in
            // the LDS tables, the rest went to global memory -- h_qbest[] holds the keys until the election needs it, h_ucand[]
            // the indices.  The LDS part joins them, and the list is read back twice with coalesced loads, four entries in
            // flight per thread: once for the histogram over the sort's buckets, once to PARTITION it by chunk (r04: every
            // chunk used to walk the position index again -- seven walks for a 10 000-row subset, ~25 us each).
            // Agent-scope loads: the lines may sit in this CU's vector cache from an earlier round.
            for (int i = tid; i < CAND; i += kExpThreads) { P.h_qbest[i] = keys[i]; P.h_ucand[i] = cand[i]; }
            if (tid == 0) sh_i[5] = 0;
            __syncthreads();
            auto scan = [&](auto fn) __attribute__((always_inline)) {
                for (int i0 = tid; i0 < nq; i0 += 4 * kExpThreads) {
                    unsigned long long kb[4];
                    int qv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int i = i0 + u * kExpThreads;
                        kb[u] = 0; qv[u] = 0;
                        if (i < nq) {
                            kb[u] = __hip_atomic_load(P.h_qbest + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            qv[u] = __hip_atomic_load(P.h_ucand + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (i0 + u * kExpThreads < nq) fn(kb[u], qv[u]);
                }
            };
            // (a) the subset's histogram over the sort's buckets (cleared in front of step 2, untouched since)
            scan([&](unsigned long long kb, int) { atomicAdd(&hist[bucket_of(__longlong_as_double((long long)kb))], 1); });
            if (tid < 128) for (int t = tid; t < nt; t += 128) P.h_tbest[t] = ~0ull;      // (row t belongs to thread t % 128 throughout)
            lds_barrier();
            // Chunks of <= CAND keypoints, greedy over the buckets (r04: one thread walking the 1024 buckets took 100 us of a
            // 10 000-row round -- more than everything else in it).  Exclusive prefix of the bucket counts (the sort's scan,
            // into its cursor words); each boundary is then one binary search: the chunk that starts at bucket s ends in front
            // of the first bucket whose inclusive prefix exceeds pre[s] + CAND.
            int* const pre = hist + kSortBuckets + 4;
            {
                constexpr int kPer = kSortBuckets / kExpThreads;
                int cq[kPer], sum = 0;
                bool big = false;
#pragma unroll
                for (int q = 0; q < kPer; ++q) { cq[q] = hist[tid * kPer + q]; sum += cq[q]; big |= cq[q] > CAND; }
                int off;
                block_exclusive_scan_nosync(sum, &off, hist + 2 * kSortBuckets + 8);
#pragma unroll
                for (int q = 0; q < kPer; ++q) { pre[tid * kPer + q] = off; off += cq[q]; }
                if (big) sh_i[5] = -1;                                       // thousands of keypoints at one distance
            }
            lds_barrier();
            if (tid == 0 && sh_i[5] == 0) {
                int c = 0, ok = 1, start = 0;
                chb[0] = 0; chrow[0] = 0;
                for (;;) {
                    const int lim = pre[start] + CAND;
                    int lo = start, hi = kSortBuckets;
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        const int incl = mid + 1 < kSortBuckets ? pre[mid + 1] : nq;
                        if (incl > lim) hi = mid; else lo = mid + 1;
                    }
                    if (lo >= kSortBuckets) break;                           // the rest fits the chunk
                    if (c + 2 > kHugeChunks) { ok = 0; break; }
                    chb[++c] = lo; chrow[c] = pre[lo]; start = lo;
                }
                chb[++c] = kSortBuckets;
                chrow[c] = nq;
                sh_i[5] = ok ? c : -1;
            }
            lds_barrier();
            int nch = sh_i[5];
            bool by_rank = false;                                            // (uniform)
            // pivots of the chunks by rank below: (key bits, keypoint index) pairs in the stage buffer, free until the cross-check
            unsigned long long* const vk = (unsigned long long*)smem;
            unsigned* const vi = (unsigned*)(vk + kHugeChunks + 1);
            int* const vcnt = (int*)(vi + kHugeChunks + 1);
            if (nch < 0) {
                // r05: a sort bucket holds more than a chunk of keypoints (thousands at ONE distance from the round's position:
                // the sort order inside is by keypoint index, which no range of buckets can split), or the greedy chunks ran
                // out.  Chunks by RANK then: chunk c = the entries of ranks [c CAND, (c + 1) CAND) in (key bits, index) order
                // (huge_rank_pivots); before r05 such a round handed the whole run to the host loop.
                nch = (nq + CAND - 1) / CAND;
                if (nch > kHugeChunks) { status = kExpCandFull; break; }
                by_rank = true;
                huge_rank_pivots(vk, vi, vcnt, chrow, nq, nch, CAND, (const unsigned long long*)P.h_qbest, (const int*)P.h_ucand);
            } else {
                // hist[b] becomes the chunk of bucket b
                constexpr int kPer = kSortBuckets / kExpThreads;
#pragma unroll
                for (int q = 0; q < kPer; ++q) {
                    const int bq = tid * kPer + q;
                    int lo = 0, hi = nch - 1;                                // largest c with chb[c] <= bq
                    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (chb[mid] <= bq) lo = mid; else hi = mid - 1; }
                    hist[bq] = lo;
                }
            }
            // the cursor words, the chunks' entry counters of the partition, are cleared
            for (int i = tid; i < nch; i += kExpThreads) pre[i] = 0;
            lds_barrier();
            HUGE_STAMP(1);
            // (b) the partition: every entry to its chunk's slot range of h_pkey[] / h_cand[] (the chunk's sorted rows land in
            // the same range of h_cand[]); the sort's cursor words, zero since block_sort_clear, count the chunks' entries
            {
                int* const chcnt = hist + kSortBuckets + 4;
                scan([&](unsigned long long kb, int qi) {
                    const int c = by_rank ? rank_first_above(vk, vi, nch, kb, (unsigned)qi, 0ull, 0u) - 1
                                          : hist[bucket_of(__longlong_as_double((long long)kb))];
                    const int at = chrow[c] + atomicAdd(&chcnt[c], 1);
                    P.h_pkey[at] = kb;
                    P.h_cand[at] = qi;
                });
            }
            __syncthreads();
            HUGE_STAMP(2);
            // DELEGATED cross-check (r04): a round of this size is minutes of one CU's matrix cores at a fraction of their
            // rate, and microseconds of the whole chip's.  The round sorts its subset into h_cand[] as always, then PARKS
            // the run (the state a lazy run saves, + the subset's size); fm_expand_run gathers the subset's rows, runs the
            // dense reverse-NN kernel (K1) and the election on the whole GPU into h_qbest[] and resumes the run at steps 4 / 5.
            const bool deleg = P.delegate_min > 0 && (long long)nq * nt >= P.delegate_min;
            __shared__ unsigned long long sh_krange[2];      // chunks by rank: smallest / largest key of the chunk being sorted
            for (int c = 0; c < nch; ++c) {
                const int b0 = by_rank ? 0 : chb[c], b1 = by_rank ? 0 : chb[c + 1];
                const unsigned base = (unsigned)chrow[c];
                const int nc = chrow[c + 1] - chrow[c];
                if (by_rank && tid == 0) { sh_krange[0] = ~0ull; sh_krange[1] = 0ull; }     // (read behind the load barrier below)
                lds_barrier();                          // (everyone has read chrow / chb; the last chunk's tables are done with)
                block_sort_clear(hist);
                {
                    unsigned long long kmin = ~0ull, kmax = 0ull;
                    for (int i = tid; i < nc; i += kExpThreads) {
                        const unsigned long long kv = __hip_atomic_load(P.h_pkey + base + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        keys[i] = kv;
                        cand[i] = __hip_atomic_load(P.h_cand + base + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        kmin = kv < kmin ? kv : kmin; kmax = kv > kmax ? kv : kmax;
                    }
                    if (by_rank && kmin != ~0ull) { atomicMin(&sh_krange[0], kmin); atomicMax(&sh_krange[1], kmax); }
                }
                lds_barrier();
                // the sort's buckets spread over the chunk's key range (non-negative doubles order like their bit patterns)
                const double s_lo = by_rank ? __longlong_as_double((long long)sh_krange[0]) : (double)b0 / bscale;
                const double s_hi = by_rank ? __longlong_as_double((long long)sh_krange[1]) : (double)b1 / bscale;
                block_sort_pairs<CAND>(keys, cand, nkey, tix, hist, nc, s_hi, s_lo);
                for (int i = tid; i < nc; i += kExpThreads) P.h_cand[base + i] = cand[i];
                if (deleg) continue;
                if constexpr (F32) {
                    // (the candidate list aliases the sort scratch, free until the next chunk's sort)
                    lds_barrier();
                    const bool okc = x1_round_f32<kExpThreads, true>(RF, cand, nc, t0, nt, smem, (unsigned long long*)nullptr, (unsigned*)nkey,
                                                 C::kClistCap, (unsigned long long*)(hist + 2 * kSortBuckets + 16), sh_rf,
                                                 nullptr, nullptr, P.h_tbest, base);
                    if (!okc) { status = kExpListFull; break; }
                } else {
                    x1_round_wsplit<C::kSR, kExpThreads, true>(P.q_rows8, P.q_norm, cand, nc, P.t_rows8, P.t_norm, t0, nt, smem,
                                            (unsigned long long*)nullptr, (unsigned long long*)(hist + 2 * kSortBuckets + 16), 0,
                                            nullptr, nullptr, nullptr, P.h_tbest, base);
                }
            }
            HUGE_STAMP(3);
            if (status != kExpOk) break;      // (a float32 chunk whose candidate list overflowed)
            if (deleg) {
                if (tid == 0) {
                    P.resume_state[0] = top; P.resume_state[1] = seed_i; P.resume_state[2] = n_matches;
                    P.resume_state[3] = n_rounds; P.resume_state[4] = n_pairs; P.resume_state[5] = seen_n;
                    for (int k = 0; k < 4; ++k) P.resume_state[6 + k] = __double_as_longlong(cur[k]);
                    P.resume_state[10] = nq; P.resume_state[11] = t0; P.resume_state[12] = nt; P.resume_state[13] = sh_nlog;
                }
                need_cell = cell;
                status = kExpNeedXcheck;
                break;
            }
            // (b) election: train row t elects the slot its minimum names; the slot keeps its closest train row
            for (int i = tid; i < nq; i += kExpThreads) P.h_qbest[i] = ~0ull;      // (the unsorted keys are done with)
            __syncthreads();                  // the fill of h_qbest and the copies of h_cand have reached memory
            if (F32 || !P.tie_guard) {
                if (tid < 128)
                    for (int t = tid; t < nt; t += 128) {
                        const unsigned long long tb = P.h_tbest[t];
                        if (tb != ~0ull)
                            __hip_atomic_fetch_min(P.h_qbest + (unsigned)tb, (tb & 0xffffffff00000000ull) | (unsigned long long)(unsigned)t,
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
            } else if constexpr (!F32) {
                // Pairs under the float32-root guard (cold: never with SIFT-range descriptors; r05, the last round shape that
                // gave a run back to the host): the keys carry the distance's float32 bits, and a train row whose best d2
                // shares its root with d2 + 1 elects the LOWEST slot at either (round_body.h has the one-chunk form) -- the
                // slots below the merged minimum's are rescanned exactly by the whole workgroup, one tied row at a time.
                // (every thread reads the same words: the loop and its barriers are uniform)
                __shared__ unsigned sh_tie;
                for (int t = 0; t < nt; ++t) {
                    const unsigned long long tb = __hip_atomic_load(P.h_tbest + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (tb == ~0ull) continue;
                    const unsigned d2 = (unsigned)(tb >> 32);
                    unsigned smin = (unsigned)tb;
                    if (d2 >= kSqrtTieMin && sqrt_ties_up(d2)) {
                        if (tid == 0) sh_tie = smin;
                        __syncthreads();
                        gptr<const int8_t> trow = P.t_rows8 + (size_t)(t0 + t) * kDim;
                        const int tn = P.t_norm[t0 + t];
                        for (unsigned sl = tid; sl < smin; sl += kExpThreads) {
                            const int qi = __hip_atomic_load(P.h_cand + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (exact_d2_i8(P.q_rows8 + (size_t)qi * kDim, P.q_norm[qi], trow, tn) == d2 + 1u) atomicMin(&sh_tie, sl);
                        }
                        __syncthreads();
                        smin = sh_tie;
                        __syncthreads();
                    }
                    if (tid == 0)
                        __hip_atomic_fetch_min(P.h_qbest + smin, ((unsigned long long)sqrt_bits(d2) << 32) | (unsigned long long)(unsigned)t,
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            __syncthreads();
        }
        if constexpr (F32) lds_barrier();
        }

        EXP_STAMP(3);
        // ---- 4./5. accepted matches: neighbours and new results, in slot order ------------------
        // (a) compact the accepted slots in order:  tix[k] = slot | t_local << C::kSlotBits,
        //     nkey[k] = ratio bits  (k < na)
        const int ccx = center_coord(row, P.cell_w, P.width), ccy = center_coord(col, P.cell_h, P.height);
        double* const pos4 = (double*)(smem + CAND * 8);              // [C::kPosCap][4] behind rk[] (stage buffer)
        int* const dup_tab = (int*)cell_lds;                          // push own | push min | emit own | emit min
        for (int i = tid; i < kDupSlots; i += kExpThreads) {          // (the cell stage was consumed when the cross-check began)
            const int v = ((i >> 8) & 1) ? INT32_MAX : -1;
            ((v4i*)dup_tab)[i] = v4i{v, v, v, v};
        }
        unsigned long long* rk = (unsigned long long*)smem;        // result keys (the stage buffer is free now)
        int n_emit = 0, na = 0;
        int log_na = 0;                                            // (uniform) accepted matches of this round that went to the log
        bool asc_used = false;                                     // (uniform) the round pushed in ascending slot order (several chunks / blocks)
        if (nq <= kExpThreads && !huge_round) {
            // The usual size -- one thread per slot, nothing is compacted: slot order IS the order of the accepted list.
            const int i = tid;
            bool acc = false, known = false;
            double ratio = 0.0, mqx = 0, mqy = 0, px = 0, py = 0, nx = 0, ny = 0;
            unsigned long long nk = ~0ull, k1 = 0, rbits = 0;
            int qrow_idx = 0, t_local = 0;
            long long nslot = 0;
            if (i < nq) {
                const unsigned long long qb = keys[i];
                if (qb != ~0ull) {
                    if (P.lg_round) sh_anyx = 1;          // (read by thread 0 behind the barriers of the ranking below)
                    const float d = F32 ? __uint_as_float((unsigned)(qb >> 32)) : x1_key_distance((unsigned)(qb >> 32), P.tie_guard);
                    qrow_idx = cand[i];
                    t_local = (int)(unsigned)qb;
                    const double sd = P.q_selfdist[qrow_idx];
                    mqx = P.q_pos[2 * qrow_idx]; mqy = P.q_pos[2 * qrow_idx + 1];
                    px = P.t_pos[2 * (t0 + t_local)]; py = P.t_pos[2 * (t0 + t_local) + 1];
                    ratio = (double)d / sd;
                    acc = ratio < P.tau;
                }
            }
            if (P.lg_round) {                  // (uniform) the round's record: every accepted match in slot order, before the dedup
                int ao, unused;
                const int atot = block_rank_flags(acc, false, &ao, &unused, wave_cnt, rank_toggle) & 0xffff;
                if (sh_nlog + atot > P.lg_entry_cap) { status = kExpLogFull; break; }
                if (acc) {
                    const long long e = sh_nlog + ao;
                    P.lg_q[e] = qrow_idx; P.lg_t[e] = (int)(t0 + t_local); P.lg_ratio[e] = ratio;
                }
                if (tid == 0) log_header(cell, atot);
                log_na = atot;
            }
            if (acc) {
                rbits = (unsigned long long)__double_as_longlong(ratio);
                const int xd = (int)px - ccx, yd = (int)py - ccy;          // Grid_Cache.get_neighbor
                int ncol = col, nrow = row;
                if (yd < xd && yd < -xd) ncol = col - 1;
                else if (xd > yd) nrow = row + 1;
                else if (yd > -xd) ncol = col + 1;
                else nrow = row - 1;
                if (ncol >= 0 && ncol < P.cols && nrow >= 0 && nrow < P.rows) {
                    nx = (double)center_coord(nrow, P.cell_w, P.width);
                    ny = (double)center_coord(ncol, P.cell_h, P.height);
                    nk = pack4x16(blk(ny, P.cell_h), blk(nx, P.cell_w), blk(mqy, P.cell_h), blk(mqx, P.cell_w));
                }
                k1 = pack4x16((int)mqx, (int)mqy, (int)px, (int)py);
                bool in_seen;
                probe_both(P.seen, P.seen_cap, nk, P.found, P.found_cap, rbits, k1, &in_seen, &nslot, &known);
                if (in_seen) nk = ~0ull;
                keys[i] = nk;                 // (a thread reads and rewrites its own slot only)
                rk[i] = k1;
                nkey[i] = rbits;
            }
            lds_barrier();
            bool push = acc && nk != ~0ull, emit = acc && !known;
            int sp = -1, se = -1;
            if (push) sp = dup_insert(dup_tab, dup_tab + kDupSlots, mix64(nk), i, [&](int j) { return keys[j] == nk; });
            if (acc) se = dup_insert(dup_tab + 2 * kDupSlots, dup_tab + 3 * kDupSlots, mix64(rbits ^ mix64(k1)), i,
                                     [&](int j) { return rk[j] == k1 && nkey[j] == rbits; });
            lds_barrier();
            if (push && dup_tab[kDupSlots + sp] != i) push = false;
            if (acc && dup_tab[3 * kDupSlots + se] != i) emit = false;
            int po, eo;
            const int petot = block_rank_flags(push, emit, &po, &eo, wave_cnt, rank_toggle);
            const int ptot = petot & 0xffff, etot = petot >> 16;
            if (sh_top + ptot > P.stack_cap) { status = kExpStackFull; break; }
            if (n_matches + etot > P.match_cap || 2 * (n_matches + etot) > P.found_cap) { status = kExpMatchFull; break; }
            if (push) {                        // the first accepted match is popped first: reverse rank order
                const long long dst = sh_top + (ptot - 1 - po);
                P.stack[dst * 4 + 0] = mqx; P.stack[dst * 4 + 1] = mqy;
                P.stack[dst * 4 + 2] = nx;  P.stack[dst * 4 + 3] = ny;
                if (po == 0) {                 // this entry ends up on top: cache it for the next pop
                    nxt_e[0] = mqx; nxt_e[1] = mqy; nxt_e[2] = nx; nxt_e[3] = ny;
                    nxt_key = nk; nxt_slot = nslot; nxt_valid = 1;
                }
            }
            if (emit) {
                const long long dst = n_matches + eo;
                P.m_index[dst] = qrow_idx;
                P.m_pos[dst * 4 + 0] = mqx; P.m_pos[dst * 4 + 1] = mqy;
                P.m_pos[dst * 4 + 2] = px;  P.m_pos[dst * 4 + 3] = py;
                P.m_ratio[dst] = ratio;
                if (!found_insert(P.found, P.found_cap, rbits, k1)) sh_i[7] = 1;
            }
            n_emit = etot;
            if (tid == 0) top += ptot;         // (sh_top is rewritten from `top` at the next pop)
        } else {
          // r05: a chunked round may accept more matches than the LDS lists hold (CAND): the slots are then taken in BLOCKS --
          // step (a) stops in front of the 512-slot pass that would overflow the list, steps (b) / (c) run on what is there,
          // and the next block starts at that pass.  Results of an earlier block are in the found table when a later one
          // probes it (the barrier + fence between blocks); a neighbour key pushed by two blocks is skipped at its second
          // pop like any stale entry.  Pushes of a multi-block round go up in slot order and are reversed once at the end.
          int s_next = 0;                    // (uniform) first slot of the next block
          for (int blk_i = 0; s_next < nq && status == kExpOk; ++blk_i) {
            if (blk_i > 0) {
                __threadfence_block();
                __syncthreads();             // the block's emits / table inserts are visible; its LDS lists are done with
                for (int i = tid; i < kDupSlots; i += kExpThreads) {
                    const int v = ((i >> 8) & 1) ? INT32_MAX : -1;
                    ((v4i*)dup_tab)[i] = v4i{v, v, v, v};
                }
                na = 0;
            }
            int s0 = s_next;
            for (; s0 < nq; s0 += kExpThreads) {
                const int i = s0 + tid;
                bool acc = false;
                double ratio = 0.0, pq0 = 0, pq1 = 0, pt0 = 0, pt1 = 0;
                int t_local = 0;
                int qrow = 0;
                if (i < nq) {
                    // (a huge round's tables are in global memory: agent-scope loads, the election's atomics ran in L2)
                    unsigned long long qb;
                    if (HUGE && huge_round) qb = __hip_atomic_load(P.h_qbest + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else qb = keys[i];
                    if (qb != ~0ull) {
                        if (P.lg_round) sh_anyx = 1;      // (read by thread 0 behind the barriers of the ranking below)
                        // high word: the float32 distance bits (float32 route; int8 route with tie_guard) or the integer d2
                        const float d = (F32 || (HUGE && fkeys)) ? __uint_as_float((unsigned)(qb >> 32))
                                                                 : x1_key_distance((unsigned)(qb >> 32), P.tie_guard);
                        // the positions step (b) needs ride on the same memory round trip as the self distance
                        if (HUGE && huge_round) qrow = __hip_atomic_load(P.h_cand + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        else qrow = cand[i];
                        t_local = (int)(unsigned)qb;
                        const double sd = P.q_selfdist[qrow];
                        pq0 = P.q_pos[2 * qrow]; pq1 = P.q_pos[2 * qrow + 1];
                        pt0 = P.t_pos[2 * (t0 + t_local)]; pt1 = P.t_pos[2 * (t0 + t_local) + 1];
                        ratio = (double)d / sd;
                        acc = ratio < P.tau;
                    }
                }
                int o, o_unused;
                const int cnt = block_rank_flags(acc, false, &o, &o_unused, wave_cnt, rank_toggle) & 0xffff;
                // keys[] (qbest) of slots < s0 + kExpThreads are consumed (the barrier inside the ranking separates
                // those reads from these writes): entries na+o <= i never clobber unread ones
                if constexpr (HUGE) { if (huge_round && na + cnt > CAND) break; }   // (uniform) this pass opens the next block
                if (P.lg_round) {              // (uniform) the accepted list IS the round's record
                    if (sh_nlog + log_na + na + cnt > P.lg_entry_cap) { status = kExpLogFull; break; }
                    if (acc) {
                        const long long e = sh_nlog + log_na + na + o;
                        P.lg_q[e] = qrow; P.lg_t[e] = (int)(t0 + t_local); P.lg_ratio[e] = ratio;
                    }
                }
                if (acc) {
                    // (a huge round's slot numbers do not fit beside t_local: the accepted list keeps the query row itself,
                    // in cand[], which the round no longer needs -- its reads above came from global memory)
                    if (HUGE && huge_round) { tix[na + o] = t_local; cand[na + o] = qrow; }
                    else tix[na + o] = i | (t_local << C::kSlotBits);
                    nkey[na + o] = (unsigned long long)__double_as_longlong(ratio);
                    if (na + o < C::kPosCap) {               // (the stage buffer is free after the cross-check)
                        double* pp = pos4 + 4 * (na + o);
                        pp[0] = pq0; pp[1] = pq1; pp[2] = pt0; pp[3] = pt1;
                    }
                }
                na += cnt;
            }
            s_next = s0;
            if (status != kExpOk) break;
            const bool asc = blk_i > 0 || s_next < nq || na > kExpThreads;     // (uniform) pushes in slot order, reversed at the end
            asc_used = asc_used || asc;
            log_na += na;
            if (P.lg_round && s_next >= nq) { if (tid == 0) log_header(cell, log_na); }
            lds_barrier();
            EXP_STAMP(4);
#ifdef FM_PARK_PROF
            if (huge_round) HUGE_STAMP(4);
#endif
            // (b) per accepted match: neighbour key + seen probe, result key + found probe.
            //     keys[k] = neighbour key (or ~0), rk[k] = result key (int-truncated positions)
            for (int k0 = 0; k0 < na; k0 += kExpThreads) {
                const int k = k0 + tid;
                const bool live = k < na;
                double mqx = 0, mqy = 0, px = 0, py = 0, nx = 0, ny = 0;
                unsigned long long nk = ~0ull, k1 = 0, rbits = 0;
                int qrow_idx = 0;
                bool known = false;
                long long nslot = 0;
                if (live) {
                    int t_local;
                    if (HUGE && huge_round) { t_local = tix[k]; qrow_idx = cand[k]; }
                    else { t_local = tix[k] >> C::kSlotBits; qrow_idx = cand[tix[k] & ((1 << C::kSlotBits) - 1)]; }
                    rbits = nkey[k];
                    if (k < C::kPosCap) {                    // fetched together with the self distances in (a)
                        const double* pp = pos4 + 4 * k;
                        mqx = pp[0]; mqy = pp[1]; px = pp[2]; py = pp[3];
                    } else {
                        mqx = P.q_pos[2 * qrow_idx]; mqy = P.q_pos[2 * qrow_idx + 1];
                        px = P.t_pos[2 * (t0 + t_local)]; py = P.t_pos[2 * (t0 + t_local) + 1];
                    }
                    const int xd = (int)px - ccx, yd = (int)py - ccy;          // Grid_Cache.get_neighbor
                    int ncol = col, nrow = row;
                    if (yd < xd && yd < -xd) ncol = col - 1;
                    else if (xd > yd) nrow = row + 1;
                    else if (yd > -xd) ncol = col + 1;
                    else nrow = row - 1;
                    if (ncol >= 0 && ncol < P.cols && nrow >= 0 && nrow < P.rows) {
                        nx = (double)center_coord(nrow, P.cell_w, P.width);
                        ny = (double)center_coord(ncol, P.cell_h, P.height);
                        nk = pack4x16(blk(ny, P.cell_h), blk(nx, P.cell_w), blk(mqy, P.cell_h), blk(mqx, P.cell_w));
                    }
                    k1 = pack4x16((int)mqx, (int)mqy, (int)px, (int)py);
                    // two independent probes: would the neighbour be skipped when popped? is the
                    // result already in the list?
                    bool in_seen;
                    probe_both(P.seen, P.seen_cap, nk, P.found, P.found_cap, rbits, k1, &in_seen, &nslot, &known);
                    if (in_seen) nk = ~0ull;
                    keys[k] = nk;
                    rk[k] = k1;
                }
                lds_barrier();
                // Earlier entries of this round with the same key win (lists are in slot order).
                bool push = live && nk != ~0ull, emit = live && !known;
                if (na <= kDupMax) {
                    // the usual case (one chunk): every entry looks its two keys up in the LDS tables (dup_insert)
                    int sp = -1, se = -1;
                    if (push) sp = dup_insert(dup_tab, dup_tab + kDupSlots, mix64(nk), k, [&](int j) { return keys[j] == nk; });
                    if (live) se = dup_insert(dup_tab + 2 * kDupSlots, dup_tab + 3 * kDupSlots, mix64(rbits ^ mix64(k1)), k,
                                              [&](int j) { return rk[j] == k1 && nkey[j] == rbits; });
                    lds_barrier();
                    if (push && dup_tab[kDupSlots + sp] != k) push = false;
                    if (live && dup_tab[3 * kDupSlots + se] != k) emit = false;
                } else {
                    // A long list: one thread per entry walking all earlier entries would be a chain of LDS reads as long
                    // as the list.  Every WAVE takes entries k and its lanes the earlier entries j: one pass of <= 64
                    // comparisons per entry, flags in dupf[].
                    int* const dupf = hist;                              // bit 0 push, bit 1 emit duplicate
                    const int kend = min(na, k0 + kExpThreads);
                    const int lane = tid & 63, wave = tid >> 6;
                    for (int kk = k0 + wave; kk < kend; kk += kExpWaves) {
                        const unsigned long long a_nk = keys[kk], a_rk = rk[kk], a_rb = nkey[kk];
                        bool dp = false, de = false;
                        for (int j = lane; j < kk; j += 64) {
                            dp |= a_nk != ~0ull && keys[j] == a_nk;
                            de |= rk[j] == a_rk && nkey[j] == a_rb;
                        }
                        const int f = (__builtin_amdgcn_ballot_w64(dp) != 0ull ? 1 : 0) | (__builtin_amdgcn_ballot_w64(de) != 0ull ? 2 : 0);
                        if (lane == 0) dupf[kk - k0] = f;
                    }
                    lds_barrier();
                    if (live) {
                        const int f = dupf[k - k0];
                        if (f & 1) push = false;
                        if (f & 2) emit = false;
                    }
                }
                // stack push, first accepted match on top: entry of rank r goes to top + (total-1-r);
                // chunks of kExpThreads accepted matches are pushed in reverse chunk order below
                // one scan for both ranks: pushes in the low half-word, emits in the high one (<= kExpThreads each)
                int po, eo;
                const int petot = block_rank_flags(push, emit, &po, &eo, wave_cnt, rank_toggle);
                const int ptot = petot & 0xffff, etot = petot >> 16;
                // (every thread holds the same totals and counters and sh_top is stable here: no flag, no barrier)
                if (sh_top + ptot > P.stack_cap) { status = kExpStackFull; break; }
                if (n_matches + n_emit + etot > P.match_cap || 2 * (n_matches + n_emit + etot) > P.found_cap) { status = kExpMatchFull; break; }
                if (push) {
                    // The first accepted match must be popped first, i.e. sit on top.  One chunk
                    // (na <= kExpThreads, the usual case): write in reverse rank order.  More: chunks are
                    // written in ascending order and the whole region is reversed afterwards.
                    const long long dst = !asc ? sh_top + (ptot - 1 - po) : sh_top + po;
                    P.stack[dst * 4 + 0] = mqx; P.stack[dst * 4 + 1] = mqy;
                    P.stack[dst * 4 + 2] = nx;  P.stack[dst * 4 + 3] = ny;
                    if (!asc && po == 0) {            // this entry ends up on top: cache it for the next pop
                        nxt_e[0] = mqx; nxt_e[1] = mqy; nxt_e[2] = nx; nxt_e[3] = ny;
                        nxt_key = nk; nxt_slot = nslot; nxt_valid = 1;
                    }
                }
                if (emit) {
                    const long long dst = n_matches + n_emit + eo;
                    P.m_index[dst] = qrow_idx;
                    P.m_pos[dst * 4 + 0] = mqx; P.m_pos[dst * 4 + 1] = mqy;
                    P.m_pos[dst * 4 + 2] = px;  P.m_pos[dst * 4 + 3] = py;
                    P.m_ratio[dst] = __longlong_as_double((long long)rbits);
                    if (!found_insert(P.found, P.found_cap, rbits, k1)) sh_i[7] = 1;
                }
                n_emit += etot;
                if (!asc) {
                    // the only chunk of the round (the usual case): thread 0 rewrites sh_top from its own `top` at
                    // the next pop, and the round ends with a full barrier -- none needed here
                    if (tid == 0) top += ptot;
                } else {
                    lds_barrier();
                    if (tid == 0) { sh_top += ptot; top += ptot; }
                    lds_barrier();
                }
            }
          }
        }
        if (status != kExpOk) break;
        // More than one chunk: the pushed region [top_before, top) is in ascending slot order;
        // reverse it in place.
        if (asc_used) {
            __syncthreads();       // entries pushed by other threads are read from global memory below
            const long long lo = sh_i_top_before, hi = sh_top;
            const long long cntp = hi - lo;
            for (long long x = tid; x < cntp / 2; x += kExpThreads) {
                const long long a = lo + x, b = hi - 1 - x;
                for (int c = 0; c < 4; ++c) { const double t = P.stack[a * 4 + c]; P.stack[a * 4 + c] = P.stack[b * 4 + c]; P.stack[b * 4 + c] = t; }
            }
        }
        EXP_STAMP(5);
#ifdef FM_PARK_PROF
        if (huge_round) HUGE_STAMP(5);
#endif
        n_matches += n_emit;
        if (P.lg_round && tid == 0) sh_nlog += log_na;             // (read again only behind the barrier below)
        __threadfence_block();
        __syncthreads();           // table / stack writes visible before the next round reads them
        if (sh_i[7]) { status = kExpTableFull; break; }
        EXP_STAMP(6);
    }
    // (a round with an empty radius subset leaves its cell DMA un-awaited: nothing may be in flight towards this
    // workgroup's LDS when it is given back)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) {
        P.result[0] = n_matches;
        P.result[1] = n_rounds;
        P.result[2] = n_pairs;
        P.result[3] = status;
        P.result[8] = sh_nlog;
        if (P.prof) for (int k = 0; k < 12; ++k) P.result[16 + k] = pt[k];
#ifdef FM_PARK_PROF
        HUGE_STAMP(6);
        for (int k = 0; k < 8; ++k) P.result[32 + k] = (P.resume ? P.result[32 + k] : 0) + hp[k];
#endif
        if constexpr (LAZY || HUGE) {
            P.result[4] = need_cell;
            if (status == kExpNeedXcheck) { P.result[5] = P.resume_state[10]; P.result[6] = P.resume_state[11]; P.result[7] = P.resume_state[12]; }
        }
    }
}

// tier: 0 = the 2048-row kernel, 1 = the 4096-row one (int8), 2 = the chunked one (int8 without the float32-root guard, float32),
// 3 = the lazy-target variant of tier 2 (int8, and since r05 float32)
hipError_t launch_expand(const void* d_pairs, int n_pairs, bool f32, int tier, hipStream_t stream)
{
    static bool attr_set = false;
    const bool big = tier == 1;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)expand_kernel<false, kExpCand>, hipFuncAttributeMaxDynamicSharedMemorySize, ExpCfg<kExpCand>::kLdsBytes);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)expand_kernel<false, kExpCand, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ExpCfg<kExpCand>::kLdsBytes);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)expand_kernel<true, kExpCand, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ExpCfg<kExpCand>::kLdsBytes);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)expand_kernel<false, kExpCand, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ExpCfg<kExpCand>::kLdsBytes);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)expand_kernel<true, kExpCand, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, ExpCfg<kExpCand>::kLdsBytes);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)expand_kernel<true, kExpCand>, hipFuncAttributeMaxDynamicSharedMemorySize, ExpCfg<kExpCand>::kLdsBytes);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)expand_kernel<false, kExpCandBig>, hipFuncAttributeMaxDynamicSharedMemorySize, ExpCfg<kExpCandBig>::kLdsBytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    if (f32 && tier == 1) return hipErrorInvalidValue;
    if (tier == 3) {           // lazy targets (the chunked kernels, so a radius subset of any size stays on the device)
        if (f32) hipLaunchKernelGGL((expand_kernel<true, kExpCand, true, true>), dim3(n_pairs), dim3(kExpThreads), ExpCfg<kExpCand>::kLdsBytes, stream, (const ExpandPair*)d_pairs);
        else     hipLaunchKernelGGL((expand_kernel<false, kExpCand, true, true>), dim3(n_pairs), dim3(kExpThreads), ExpCfg<kExpCand>::kLdsBytes, stream, (const ExpandPair*)d_pairs);
        return hipGetLastError();
    }
    if (tier == 2 && f32) {
        hipLaunchKernelGGL((expand_kernel<true, kExpCand, true>), dim3(n_pairs), dim3(kExpThreads), ExpCfg<kExpCand>::kLdsBytes, stream, (const ExpandPair*)d_pairs);
        return hipGetLastError();
    }
    if (tier == 2) {
        hipLaunchKernelGGL((expand_kernel<false, kExpCand, true>), dim3(n_pairs), dim3(kExpThreads), ExpCfg<kExpCand>::kLdsBytes, stream, (const ExpandPair*)d_pairs);
        return hipGetLastError();
    }
    if (f32)      hipLaunchKernelGGL((expand_kernel<true, kExpCand>), dim3(n_pairs), dim3(kExpThreads), ExpCfg<kExpCand>::kLdsBytes, stream, (const ExpandPair*)d_pairs);
    else if (big) hipLaunchKernelGGL((expand_kernel<false, kExpCandBig>), dim3(n_pairs), dim3(kExpThreads), ExpCfg<kExpCandBig>::kLdsBytes, stream, (const ExpandPair*)d_pairs);
    else          hipLaunchKernelGGL((expand_kernel<false, kExpCand>), dim3(n_pairs), dim3(kExpThreads), ExpCfg<kExpCand>::kLdsBytes, stream, (const ExpandPair*)d_pairs);
    return hipGetLastError();
}

int expand_cand_cap() { return kExpCand; }
int expand_cand_cap_big() { return kExpCandBig; }

}  // namespace fm
