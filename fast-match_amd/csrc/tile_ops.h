// Device helpers shared by the dense row-reduce kernel (rowreduce.hip) and the per-round
// kernel (rounds.hip): MFMA vector types, stage geometry, the in-lane max tree and the
// exact top-K state of one lane.
#pragma once
#include "fm_internal.h"

namespace fm {

typedef int v4i  __attribute__((ext_vector_type(4)));

// Pointer into GLOBAL memory (address space 1).  A pointer that a kernel reads from memory (not
// from its argument list) is a generic one to the compiler, and every access through it becomes a
// FLAT instruction, which counts on the LDS counter as well as on the vector-memory counter: each
// LDS wait then also waits for all outstanding global traffic.  The one-workgroup kernels
// (expand.hip) convert such pointers once; the device functions they share with kernels whose
// pointers come from the argument list are templates over the pointer type.
template <class T> using gptr = __attribute__((address_space(1))) T*;
typedef int v16i __attribute__((ext_vector_type(16)));

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every
// outstanding GLOBAL load and store of the wave (s_waitcnt vmcnt(0)); in the latency-bound
// one-workgroup kernel (K7) that drain would sit on the critical path of each of the ~30 steps of
// a round.  Use where the threads exchange data through LDS only: global data written by one
// thread and read by another needs a real __syncthreads() in between.  (Only meaningful when the
// global accesses are global_* instructions, see gptr above: FLAT ones count on lgkmcnt too.)
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

constexpr int kStageRowBytes = kStageRows * kDim;                           // 16384
constexpr int kStageAuxBytes = (kStageRows / kTileRows) * kAuxPerTile * 4;  // 1024
constexpr int kStageBytes    = kStageRowBytes + kStageAuxBytes;             // 17408

// v_max3_i32 tree over the 16 accumulator registers of one 32x32 tile.
__device__ __forceinline__ int max16(const v16i& a)
{
    int m0 = max(max(a[0], a[1]), a[2]);
    int m1 = max(max(a[3], a[4]), a[5]);
    int m2 = max(max(a[6], a[7]), a[8]);
    int m3 = max(max(a[9], a[10]), a[11]);
    int m4 = max(max(a[12], a[13]), a[14]);
    int m5 = max(max(m0, m1), m2);
    int m6 = max(max(m3, m4), a[15]);
    return max(m5, m6);
}

// Read the 16 accumulator-order words (cinit or low) of one tile from LDS.
__device__ __forceinline__ v16i lds_read16(const char* p)
{
    const v4i* ax = (const v4i*)p;
    const v4i c0 = ax[0], c1 = ax[1], c2 = ax[2], c3 = ax[3];
    return v16i{c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3],
                c2[0], c2[1], c2[2], c2[3], c3[0], c3[1], c3[2], c3[3]};
}

// Row of a tile that accumulator register r of lane half h holds (C/D map of
// v_mfma_i32_32x32x32_i8): (r & 3) + 8 (r >> 2) + 4 h.
__device__ __forceinline__ int tile_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---- exact top-K state of one lane for one output row -----------------------------------
// A candidate is the packed key  (acc << 5) | (npar << 4) | (15 - r):
//   key >> 4 = hi = 2*acc + npar, and |m|^2 - 2 c.m = 1 - hi, so larger hi = smaller d2;
//   the low nibble makes the lowest register (= lowest row in this lane's tile share)
//   win ties inside a tile.  Across tiles candidates arrive in ascending row order, so the
//   comparison against the running state uses hi only and is strict: the earlier (lower
//   index) candidate is kept -- cv::batchDistance's rule.
// tile < 0: -1 = empty slot.
constexpr int kNoKey = INT32_MIN;

// Best candidate of one lane for one train row over 32x32 tiles (the per-round kernels keep one: the
// cross-check needs the nearest query row only).
struct TopTile {
    int key;
    int tile;

    __device__ __forceinline__ void init() { key = kNoKey; tile = -1; }
    // Smallest accumulator value that can still enter: needs 2*acc + npar > hi, possible iff acc >= ceil(hi / 2).
    __device__ __forceinline__ int own_threshold() const { return tile >= 0 ? (((key >> 4) + 1) >> 1) : INT32_MIN; }

    // Exact update with the 16 candidates of tile `t`.
    __device__ __forceinline__ void update(const v16i& acc, const v16i& low, int t)
    {
        int k[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) k[r] = (acc[r] << 5) | low[r];
        const int m0 = max(max(k[0], k[1]), k[2]);
        const int m1 = max(max(k[3], k[4]), k[5]);
        const int m2 = max(max(k[6], k[7]), k[8]);
        const int m3 = max(max(k[9], k[10]), k[11]);
        const int m4 = max(max(k[12], k[13]), k[14]);
        const int km = max(max(max(m0, m1), m2), max(max(m3, m4), k[15]));
        const bool up = (km >> 4) > (key >> 4);
        key = up ? km : key;
        tile = up ? t : tile;
    }

    // hi and the row index inside the gathered subset (valid only if tile >= 0).
    __device__ __forceinline__ int hi() const { return key >> 4; }
    __device__ __forceinline__ int index(int h) const { return tile * kTileRows + tile_row(15 - (key & 15), h); }
};

// ---- the same state for the 16x16x64 tile shape (rowreduce.hip) ---------------------------
// C/D map of v_mfma_i32_16x16x64_i8: col = lane & 15, row = 4 (lane >> 4) + reg, 4 registers
// per 16-row tile.  A lane examines a 32-row unit = two tiles = 8 candidates at once; the
// candidate key is (acc << 5) | (npar << 4) | (15 - id) with id = 4 * sub + reg (sub = tile
// within the unit), so ties inside a unit go to the lowest row of this lane's share.
typedef int v4i_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int unit_row(int id, int g) { return 16 * (id >> 2) + 4 * g + (id & 3); }

// median of three (v_med3_i32)
__device__ __forceinline__ int med3i(int a, int b, int c)
{
    int r;      // (the compiler forms v_med3 only for clamps against constants)
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

template <int KTOP>
struct TopK8 {
    int key[KTOP];
    int unit[KTOP];

    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int k = 0; k < KTOP; ++k) { key[k] = kNoKey; unit[k] = -1; }
    }
    __device__ __forceinline__ int kth_hi() const { return key[KTOP - 1] >> 4; }
    __device__ __forceinline__ bool full() const { return unit[KTOP - 1] >= 0; }
    __device__ __forceinline__ int own_threshold() const { return full() ? ((kth_hi() + 1) >> 1) : INT32_MIN; }

    __device__ __forceinline__ bool update(const v4i_t& a0, const v4i_t& a1, const v4i_t& l0, const v4i_t& l1, int u)
    {
        int k[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { k[r] = (a0[r] << 5) | l0[r]; k[4 + r] = (a1[r] << 5) | l1[r]; }
        if constexpr (KTOP == 1) {
            const int m0 = max(max(k[0], k[1]), k[2]);
            const int m1 = max(max(k[3], k[4]), k[5]);
            const int km = max(max(max(k[6], k[7]), m0), m1);
            const bool up = (km >> 4) > (key[0] >> 4);
            key[0] = up ? km : key[0];
            unit[0] = up ? u : unit[0];
            return up;
        } else {
            // top-2 of the 8 keys (unique inside a unit): two triples (largest = v_max3, second = v_med3)
            // and a pair, merged as sorted pairs -- 12 VALU (the pairwise tree: 20)
            const int a1_ = max(max(k[0], k[1]), k[2]), a2_ = med3i(k[0], k[1], k[2]);
            const int b1_ = max(max(k[3], k[4]), k[5]), b2_ = med3i(k[3], k[4], k[5]);
            const int c1_ = max(k[6], k[7]), c2_ = min(k[6], k[7]);
            const int m1 = max(a1_, b1_);
            const int m2 = max(max(min(a1_, b1_), a2_), b2_);
            const int k1 = max(m1, c1_);
            const int k2 = max(max(min(m1, c1_), m2), c2_);
            const int h1 = k1 >> 4, h2 = k2 >> 4, g1 = key[0] >> 4, g2 = key[1] >> 4;
            const bool enter = h1 > g2, first = h1 > g1, both = h2 > g1;
            const int n0k = first ? k1 : key[0];
            const int n0u = first ? u : unit[0];
            const int n1k = first ? (both ? k2 : key[0]) : (enter ? k1 : key[1]);
            const int n1u = first ? (both ? u : unit[0]) : (enter ? u : unit[1]);
            key[0] = n0k; unit[0] = n0u; key[1] = n1k; unit[1] = n1u;
            return enter;
        }
    }
    __device__ __forceinline__ int hi(int k) const { return key[k] >> 4; }
    __device__ __forceinline__ int index(int k, int g) const
    {
        return unit[k] * kTileRows + unit_row(15 - (key[k] & 15), g);
    }
};

// ---- the triangular self sweeps (rowreduce.hip TRI, filter_f16.hip TRI) ------------------------------------
// Entry e of the triangular sweep's workgroup list (plan_tri / tri_pieces in api_grid.hip build the same list on the host, for
// fm_self_dist_plan and the tests): the first nchunks entries are the diagonal blocks (chunk e against its own four stages);
// then piece-number major -- round i holds piece i of every chunk k that still has stages from 4 k + 4 + i S on, i.e. the first
// ceil((nstages - 4 - i S) / 4) chunks.  (r05, last: a device table per bank size -- hipMalloc, upload, a cache of 64 -- made
// the plan of a NEW size cost ~50 us; a dataset of small images has a new size per image.)  Scalar: at most nstages / S rounds.
__device__ __forceinline__ void tri_entry(int e, int nchunks, int nstages, int S, int& chunk, int& st0, int& st1)
{
    if (e < nchunks) { chunk = e; st0 = 4 * e; st1 = min(nstages, 4 * e + 4); return; }
    int r = e - nchunks, i = 0;
    for (;;) {
        const int cnt = (nstages - 4 - i * S + 3) >> 2;       // chunks of round i (> 0 for every entry of the list)
        if (r < cnt || cnt <= 0) break;
        r -= cnt;
        ++i;
    }
    chunk = r;
    st0 = 4 * r + 4 + i * S;
    st1 = min(nstages, st0 + S);
}

// ---- float32 sqrt ties of the integer route -------------------------------------------------
// cv::batchDistance takes dist = sqrtf((float)d2) BEFORE the k-NN insertion and the cross-check
// compare, so candidates are ordered by (float32 bits of the distance,
// index), not by (d2, index).  The two orders differ only where two integers share one float32
// square root: never below kSqrtTieMin (exhaustive check over [0, 128 * 255^2]), and above it only
// as pairs {n, n + 1} (no three integers share a root).  The matrix-core kernels order by d2; the
// callers repair the rows where that can matter -- the K-th best d2 reached kSqrtTieMin -- by an
// exact rescan (sqrt_fix_kernel in api_match.hip, the cold branch of x1_round_wsplit), and only for bank
// pairs whose row norms allow such a distance at all (Bank::usq_max): SIFT-range descriptors
// (|d|^2 ~ 2.6e5) never take any of it.
constexpr unsigned kSqrtTieMin = 4197200u;

__device__ __forceinline__ unsigned sqrt_bits(unsigned d2) { return __float_as_uint(sqrtf((float)d2)); }
// d2 (>= kSqrtTieMin) shares its float32 root with d2 + 1
__device__ __forceinline__ bool sqrt_ties_up(unsigned d2) { return sqrt_bits(d2 + 1u) == sqrt_bits(d2); }

// Exact squared distance of two bank rows (int8 = u8 - 128, 128 bytes each) from their norms.
template <class PA, class PB>
__device__ __forceinline__ unsigned exact_d2_i8(PA a, int na, PB b, int nb)
{
    int dot = 0;
#pragma unroll
    for (int c = 0; c < kDim / 16; ++c) {
        const v4i x = *(const __attribute__((address_space(1))) v4i*)(a + 16 * c);
        const v4i y = *(const __attribute__((address_space(1))) v4i*)(b + 16 * c);
#pragma unroll
        for (int w = 0; w < 4; ++w) dot = __builtin_amdgcn_sdot4(x[w], y[w], dot, false);
    }
    return (unsigned)(na + nb - 2 * dot);
}

// (hi, idx) a is better than b: larger hi, then lower index.  idx < 0 means "none".
__device__ __forceinline__ bool better(int ah, int ai, int bh_, int bi_)
{
    if (ai < 0) return false;
    if (bi_ < 0) return true;
    return ah > bh_ || (ah == bh_ && ai < bi_);
}

}  // namespace fm
