// K1 -- int8-MFMA pairwise squared-L2 tile kernel with a fused reduce-over-M epilogue.
//
// Replaces the arithmetic of cv::batchDistance behind
//   cv2.BFMatcher(NORM_L2, crossCheck).knnMatch        (fastmatch.pyx:122-123, 161-162;
//                                                       matchutil.py:42-43)
// for uint8 / integer-valued descriptors: d2(c, m) = |c|^2 + |m|^2 - 2 c.m exactly in
// int32 on bytes shifted by -128.
//
// Mapping onto v_mfma_i32_16x16x64_i8 (D[16 x 16] += A[16 x 64] B[64 x 16], two per
// 16 x 16 x 128 tile).  This shape sustains ~80 % of the nominal int8 rate on MI355X where
// 32x32x32 holds 53-69 % (scripts/ablate/shape.hip: the chip clocks the smaller shape higher).
//   N (lane & 15)        = the output row ("column" c) whose nearest neighbours we want;
//                          its two 64-byte K-halves stay in VGPRs for the whole sweep.
//   M (4 (lane>>4) + reg) = the rows being reduced over, streamed through LDS.
//   Accumulator init     = -(|m|^2 >> 1) from the bank's aux array, so the accumulator is
//                          acc = c.m - (|m|^2 >> 1) and  |m|^2 - 2 c.m = 1 - (2 acc + npar).
// A lane holds 4 candidates of ONE output row per tile and examines a 32-row unit (two
// tiles, 8 candidates) at a time: the fast path is a 4-op v_max3 tree and one compare against
// the lane's current K-th best accumulator value; only units that can change a lane's top-K
// take the exact (hi = 2 acc + npar, index) update.  Lane groups, waves and blocks never
// exchange data until one merge after the sweep.
//
// Grid: blockIdx -> (chunk of 16*NC*NW output rows, split of the reduction range), split
// major.  Staging: 128 rows (16 KiB) + 1 KiB aux per step, global_load_lds_dwordx4 into a
// double buffer shared by the NW waves, XOR swizzle applied on the source address so that
// the ds_read_b128 A-fragment reads are bank-conflict free.
#include "tile_ops.h"
#include <type_traits>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <functional>

#ifdef FM_COUNT_VISITS
__device__ unsigned long long g_visits[256];     // [split]: exact-path visits (wave, block), [64 + split]: lanes that want them, [128 + split]: units x blocks
extern "C" int fm_debug_visits(unsigned long long* out, int reset)
{
    if (out) (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_visits), sizeof(g_visits));
    if (reset) { static unsigned long long z[256]; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_visits), z, sizeof(z)); }
    return 0;
}
#endif

#ifdef FM_CLOCK_STAMP
// Diagnostic build only (MI355X_MICROARCH.md, DVFS give-back item 6): shader cycles (s_memtime) and 100 MHz
// wall ticks (s_memrealtime) around the stage loop of every workgroup; in-kernel clock = 100 MHz x cycles / ticks.
// The stamps go to a buffer nothing else reads; the product build executes none of this.
__device__ unsigned long long g_clock[2 * 8192];
extern "C" int fm_debug_clock(unsigned long long* out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_clock), (size_t)(n < 8192 ? n : 8192) * 16);
}
#endif

namespace fm {

struct RRParams {
    const int8_t*  col_rows;
    const int32_t* col_norm;
    int            ncols_pad;
    const int8_t*  red_rows;
    const int32_t* red_aux;
    int            nred;          // real rows of the reduced bank
    int            nstages;       // nred_pad / 128
    int            nsplit;
    int            nchunks;
    int            stages_per_split;
    int            ncols_alloc;
    unsigned long long* partial;
    int*           bound;         // [ncols_alloc] shared K-th-best bounds (INT32_MIN filled) or null
    int            order;         // workgroup -> (chunk, split) mapping, see map_block
    int            bound_mask;    // shared bounds are re-read at every stage of a sweep's first 8 and then at the stages
                                  // whose number & bound_mask == 0 (0: every stage; option "bound_every" 1 | 2 | 4 | 8)
#ifdef FM_ABLATE
    int            tri_nocol;     // TRI, -DFM_ABLATE builds only (FM_TRI_NOCOL=1, WRONG results): the column direction never fires
#endif
    int            tri_first;     // TRI kernels: workgroup bid of a launch is entry tri_first + bid of the bank's workgroup list
    int            tri_S;         // ... whose entries follow from (nchunks, nstages, tri_S) by arithmetic: tri_entry below
};

// (tri_entry -- entry e of the triangular sweep's workgroup list -- is in tile_ops.h: filter_f16.hip's float32 form shares it)

// Accumulator value of a masked (output row == reduced row) pair in the SELF kernels: below the padding
// rows' -2^25, so the diagonal never beats anything, and (value << 5) still fits int32.
constexpr int kSelfMasked = kPadCinit - 1;

// ---- triangular self sweep (TRI) ------------------------------------------------------------------------
// d(i, j) = d(j, i): the masked sweep above computes every distance of a bank against itself twice.  The TRI
// kernels sweep, for an output chunk, only the stages from the chunk's own rows on (the square block on the
// diagonal in full, masked as above) and use every tile in BOTH directions: along the lane as before (the
// output row's best over the streamed rows) and across it -- the streamed row's best over the output rows --
// as a filter against the streamed rows' own words of bound[], which hold the same thing (the best hi a row
// has reached as an OUTPUT row).  Only the VALUE of the minimum is kept (cache.pyx:252, 273 keep r[1].distance),
// so bound[] itself carries the result: every improvement of either direction is published with an atomic
// maximum, and d2(i) = |i|^2 + 1 - bound[i] after the sweep (selfdist_tri_finish_kernel).
//   frame of bound[m], m as output row, partner c:   B = 2 c.m + 1 - |c|^2
//   what a lane holds of the pair (m streamed, c = its output row):  acc = c.m + cinit[m],  cinit = -(|m|^2 >> 1)
//   B > bound[m]   <=>   2 acc - |c|^2 >= bound[m] + 2 cinit[m] =: X[m]                       (exact, integers)
// X[m] is staged per stage in LDS beside the rows (the bounds are read two stages ahead: a stale bound is
// merely weaker), together with X8 = the minimum of X over the 8 rows a lane holds of a 32-row unit: the fast
// path tests max(acc) of the unit against X8 (one v_lshl_add + one compare per unit and block), the exact
// path tests row by row and publishes.
constexpr int kTriNever   = 1 << 28;          // X of a row that takes nothing (padding, the diagonal block)
constexpr int kTriAlways  = -(1 << 28);       // X of a row without a bound yet
constexpr int kTriXRow    = kStageBytes;                 // int32[128] X per staged row, accumulator order
constexpr int kTriX8      = kStageBytes + 512;           // int32 X8 at 64 * unit + 16 * lane group
constexpr int kStageBytesTri = kStageBytes + 512 + 256;  // 18176
constexpr int kTriRaw     = 3 * kStageBytesTri;          // behind the three stage buffers: bound[] words of 128 rows, their norms
constexpr int kTriGn      = kTriRaw + 1024;              // per wave 4 x 64 words: the refreshed bounds of its own output rows
constexpr int kTriLdsBytes = kTriGn + 8 * 1024;

// Workgroups of a bank pair that the grid holds under `order` (RowReducePlan::order, option "k1_order"):
//   0  split major: bid -> (chunk = bid % nchunks, split = bid / nchunks).  Consecutive workgroups -- dealt
//      round-robin over the 8 XCDs -- sweep the SAME slice of the reduced bank for different output chunks,
//      so every XCD's L2 fetches every slice, and the 13 workgroups of an output chunk sit on different XCDs.
//   1  chunks owned by an XCD: the workgroups with equal (bid & 7) -- one XCD under the observed round-robin
//      placement, a speed assumption only -- take the output chunks = (bid & 7) mod 8 for ALL splits, split
//      major in time; the stationary operand of a chunk is then re-read from that XCD's own L2.  The
//      nchunks % 8 chunks that are left over are dealt out one (chunk, split) at a time at the end.
//   2  contiguous share of the split-major order per XCD (the guide's T1 remap): an XCD sweeps 1-3 slices of
//      the reduced bank instead of all of them, a slice is fetched by 1-2 L2s instead of eight.
// Orders 1 and 2 pad the grid of a pair to a multiple of 8 workgroups so that (bid & 7) names the same XCD
// group for every pair of a batched launch; the workgroups beyond the last (chunk, split) exit at once.
__host__ __device__ inline int blocks_per_pair(int nchunks, int nsplit, int order)
{
    const int nblk = nchunks * nsplit;
    if (order == 1) return 8 * ((nchunks >> 3) * nsplit + (((nchunks & 7) * nsplit + 7) >> 3));
    if (order == 2) return 8 * ((nblk + 7) >> 3);
    return nblk;
}

__device__ __forceinline__ bool map_block(const RRParams& p, const int bid, int& chunk, int& split)
{
    const int nblk = p.nchunks * p.nsplit;
    if (p.order == 0) {
        chunk = bid % p.nchunks;
        split = bid / p.nchunks;
        return bid < nblk;
    }
    const int x = bid & 7, r = bid >> 3;
    if (p.order == 2) {
        const int per = (nblk + 7) >> 3;
        const int l = x * per + r;
        chunk = l % p.nchunks;
        split = l / p.nchunks;
        return l < nblk;
    }
    const int nc8 = p.nchunks >> 3, rem = p.nchunks & 7, owned = nc8 * p.nsplit;
    if (r < owned) {
        split = r / nc8;
        chunk = (r - split * nc8) * 8 + x;
        return true;
    }
    const int l = (r - owned) * 8 + x;
    if (rem == 0 || l >= rem * p.nsplit) { chunk = 0; split = 0; return false; }
    split = l / rem;
    chunk = nc8 * 8 + (l - split * rem);
    return true;
}

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// Relaxed agent-scope load of a shared bound whose completion the CALLER tracks (NBUF == 3): the
// compiler would put s_waitcnt vmcnt(0) in front of the first use, which also waits for the LDS-DMA
// issued after the load.  The value is valid after a later s_waitcnt vmcnt(k) with k <= number of
// VMEM operations issued after it (wait_dma_and_bounds below ties the registers to that wait).
__device__ __forceinline__ void load_bound_untracked(int& dst, const int* ptr)
{
    asm volatile("global_load_dword %0, %1, off sc1" : "=v"(dst) : "v"(ptr) : "memory");
}
// The same through a scalar base and a 32-bit byte offset (+ immediate): no 64-bit address lives in VGPRs.
template <int IMM, bool SC1 = true>
__device__ __forceinline__ void load_word_untracked_s(int& dst, const int* base, unsigned voff)
{
    // ("+v": the loop-carried variable keeps ONE register through the unrolled stage loop -- with "=v" the copies of the loop
    // body got registers of their own and the compiler moved values between them in front of the wait, tests/test_isa_hazards.py)
    if constexpr (SC1) asm volatile("global_load_dword %0, %1, %2 offset:%3 sc1" : "+v"(dst) : "v"(voff), "s"(base), "n"(IMM) : "memory");
    else               asm volatile("global_load_dword %0, %1, %2 offset:%3" : "+v"(dst) : "v"(voff), "s"(base), "n"(IMM) : "memory");
}

template <bool GLDS, int NW>
__device__ __forceinline__ void issue_stage(const RRParams& p, int stage, char* buf, int wave, int lane)
{
    const int8_t* src_rows = p.red_rows + (size_t)stage * kStageRowBytes;
    const int slot = lane & 7;
    constexpr int kPieces = 16 / NW;              // 1-KiB pieces (8 rows) per wave
#pragma unroll
    for (int i = 0; i < kPieces; ++i) {
        const int g   = wave * kPieces + i;
        const int row = g * 8 + (lane >> 3);
        // (uniform base + unsigned 32-bit lane offset: scalar-base addressing, no 64-bit VGPR pointers)
        const int8_t* src = src_rows + (unsigned)(row * kDim + 16 * (slot ^ ((row >> 1) & 7)));
        if constexpr (GLDS) {
            __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(buf + g * 1024), 16, 0, 0);
        } else {
            *(v4i*)(buf + g * 1024 + lane * 16) = *(const v4i*)src;
        }
    }
    if (wave == NW - 1) {
        const int32_t* src = p.red_aux + (size_t)stage * (kStageAuxBytes / 4) + (unsigned)(lane * 4);
        if constexpr (GLDS) {
            __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(buf + kStageRowBytes), 16, 0, 0);
        } else {
            *(v4i*)(buf + kStageRowBytes + lane * 16) = *(const v4i*)src;
        }
    }
}

// The same for 8 waves with a wave-UNIFORM wave number (TRI kernels): scalar base + one 32-bit lane offset per
// LDS-DMA instruction and the LDS destination in M0 straight from scalars -- the compiler's selection for the
// builtin keeps a 64-bit address pair per piece and the LDS addresses in VGPRs (8 registers the TRI kernel lacks).
// lo = 128 (lane >> 3) + 16 ((lane & 7) ^ (lane >> 4)): a wave's pieces g = 2 wave, 2 wave + 1 differ in the
// swizzle term (row >> 1) & 7 by 4 (g & 1) only, i.e. in bit 6 of the byte offset.
// (M0 is a reserved register to the compiler and is refused in a clobber list ("may lead to undefined behaviour"), so the
// statements below change it behind the compiler's back.  That is safe while every use sets M0 in the same statement and
// no kernel that uses them also holds a compiler-generated reader of M0 -- the builtin's global_load_lds, movrel,
// sendmsg: tests/test_isa_hazards.py reads the generated ISA for exactly that.)
__device__ __forceinline__ void lds_dma_16(unsigned lds_addr, const void* sbase, unsigned voff)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 :: "s"(lds_addr), "v"(voff), "s"(sbase) : "memory");
}
// Agent-scope atomic maximum (no return) through a scalar base and a 32-bit byte offset + immediate.
template <int IMM>
__device__ __forceinline__ void atomic_max_s(const int* base, unsigned voff, int val)
{
    asm volatile("global_atomic_smax %0, %1, %2 offset:%3" :: "v"(voff), "v"(val), "s"(base), "n"(IMM) : "memory");
}
// Maximum over the 16 lanes of a DPP row (the lanes of one lane group: the 16 output rows that face one streamed row).
__device__ __forceinline__ int rowmax16(int x)
{
    x = max(x, __builtin_amdgcn_update_dpp(INT32_MIN, x, 0x128, 0xf, 0xf, false));     // row_ror:8
    x = max(x, __builtin_amdgcn_update_dpp(INT32_MIN, x, 0x124, 0xf, 0xf, false));     // row_ror:4
    x = max(x, __builtin_amdgcn_update_dpp(INT32_MIN, x, 0x122, 0xf, 0xf, false));     // row_ror:2
    x = max(x, __builtin_amdgcn_update_dpp(INT32_MIN, x, 0x121, 0xf, 0xf, false));     // row_ror:1
    return x;
}
// Lane number recomputed where it is used (two VALU): keeps lane-derived addresses of the once-per-stage paths out
// of the registers that live across the unit loop (volatile: not hoisted out of the stage loop again).
__device__ __forceinline__ int lane_now()
{
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
// One dword per lane into LDS (64 consecutive words from lds_addr on); sc1: the bounds are other workgroups' atomics.
template <bool SC1>
__device__ __forceinline__ void lds_dma_4(unsigned lds_addr, const void* sbase, unsigned voff)
{
    if constexpr (SC1) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2 sc1" :: "s"(lds_addr), "v"(voff), "s"(sbase) : "memory");
    else               asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" :: "s"(lds_addr), "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void issue_stage_u8(const RRParams& p, int stage, char* buf, int wave, int lane, unsigned lo)
{
    const int8_t* base = p.red_rows + (size_t)stage * kStageRowBytes + wave * 2048;
    const unsigned dst = (unsigned)(size_t)LDS_PTR(buf) + (unsigned)wave * 2048u;
    lds_dma_16(dst, base, lo);
    unsigned lo1;       // (lo ^ 64 formed here: as a loop invariant it would be one more register across the unit loop)
    asm volatile("v_xor_b32 %0, 64, %1" : "=v"(lo1) : "v"(lo));
    lds_dma_16(dst + 1024u, base + 1024, lo1);
    if (wave == 7)
        lds_dma_16((unsigned)(size_t)LDS_PTR(buf) + kStageRowBytes, p.red_aux + (size_t)stage * (kStageAuxBytes / 4), (unsigned)lane_now() * 16u);
}

// NW waves per workgroup share every staged tile; each wave owns NC blocks of 16 output rows.
// NBUF = LDS stage buffers: 2 = the stage consumed next is the one prefetched last (its LDS-DMA is
// waited for with vmcnt(0) at every stage hand-over); 3 = prefetch two stages ahead, so the
// hand-over only waits for a DMA issued a whole stage earlier and the newest one stays in flight.
// bid = index of the workgroup inside ITS launch of one bank pair (rowreduce_kernel: blockIdx.x;
// rowreduce_batch_kernel: blockIdx.x modulo the workgroups of a pair).
// SELF: both banks are the same bank and a row is not its own neighbour -- the pair (n, n) is masked, so
// the top-1 is min over m != n of d(n, m): what Metric_Cache keeps of bf_match(d, d, k = 2), r[1].distance
// (cache.pyx:250-252, 271-273; d(n, n) = 0 is always rank 0, and a duplicate's 0 is that minimum).  The
// 32-row unit that holds a wave's own rows is a wave-uniform test, twice per sweep at most.
template <int NC, int KTOP, bool GLDS, int NW, int NBUF, int PRIO, bool SELF = false, bool TRI = false>
__device__ __forceinline__ void rowreduce_body(const RRParams& p, const int bid, char* smem)
{
    static_assert(NBUF == 2 || (NBUF == 3 && GLDS), "three stage buffers need the LDS-DMA path");
    static_assert(!SELF || KTOP == 1, "the masked-diagonal sweep is a top-1");
    static_assert(!TRI || (SELF && NBUF == 3 && NW == 8 && NC == 4), "the triangular sweep is built for one shape");
    constexpr int kStride = TRI ? kStageBytesTri : kStageBytes;     // bytes of one LDS stage buffer
    // SW ("scalar wave"): the wave number is made wave-uniform up front, the LDS-DMA goes through a scalar base + one
    // 32-bit lane offset (issue_stage_u8) and the shared bounds through a scalar base + immediates -- the registers that
    // the compiler's 64-bit address pairs take are what the top-2 kernel spills (r05: 5 spilled VGPRs with two stage
    // buffers, 10 with three -> 0)
#ifdef FM_K1_SW          // (A/B builds: the top-1 kernel in the same mode)
    constexpr bool SW = TRI || (NW == 8 && GLDS && NC == 4);
#else
    constexpr bool SW = TRI || (KTOP == 2 && NW == 8 && GLDS && NC == 4);
#endif

    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = SW ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6;
    const int g    = lane >> 4;          // lane group: rows 4g .. 4g+3 of every 16-row tile
    const unsigned dma_lo = 128u * (lane >> 3) + 16u * ((lane & 7) ^ (lane >> 4));      // (TRI: issue_stage_u8)
    const int c16  = lane & 15;
    // Split-major in time under every order (map_block): the first wave of resident workgroups covers
    // (nearly) every output chunk for the first few slices, so the bounds it publishes serve all later
    // workgroups (which reduce other slices for the same output rows) from their first tile on.
    int chunk, split, st0, st1;
    if constexpr (TRI) {
        tri_entry(__builtin_amdgcn_readfirstlane(p.tri_first + bid), p.nchunks, p.nstages, p.tri_S, chunk, st0, st1);   // (uniform: scalar)
        split = 0;
    } else {
        if (!map_block(p, bid, chunk, split)) return;
        st0 = split * p.stages_per_split;
        st1 = min(st0 + p.stages_per_split, p.nstages);
    }
    const int cb    = chunk * (16 * NC * NW) + wave * (16 * NC);

    // Stationary operand: this wave's NC x 16 output rows, two 64-byte K-halves each.
    v4i bf[NC][2];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int n = cb + 16 * j + c16;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (n < p.ncols_pad)
                bf[j][c] = *(const v4i*)(p.col_rows + (size_t)n * kDim + 64 * c + 16 * g);
            else
                bf[j][c] = v4i{0, 0, 0, 0};
        }
    }

    TopK8<KTOP> top[NC];
    int thr[NC];          // a unit is examined exactly only if some acc >= thr.  Both sources of thr -- the
                          // lane's own K-th best and the published bounds -- only ever rise, so thr = max(thr, new)
    int ret1[NC];         // KTOP == 2: what bound1[] held when this lane last published a new best (see below)
#pragma unroll
    for (int j = 0; j < NC; ++j) { top[j].init(); thr[j] = INT32_MIN; ret1[j] = INT32_MIN; }
    // KTOP == 2: bound[0 .. ncols_alloc) = bound1 (best hi any lane has reached), bound[ncols_alloc ..) =
    // bound2, a lower bound of the GLOBAL second best, which is what the thresholds come from.  A lane
    // feeds bound2 with its own second best and, when its best changes, with min(new best, previous
    // content of bound1): that content is the hi of a different candidate (another lane's best, or this
    // lane's displaced one), so two candidates reach the minimum.  The fetch-max that returns it is
    // consumed at the next stage hand-over, behind the wait the hand-over needs anyway.
    int* const bound_thr = (KTOP == 2 && p.bound) ? p.bound + p.ncols_alloc : p.bound;

    // Per-lane LDS offsets of the A fragments (swizzled) and of the aux words.
    const int sw = (c16 >> 1) & 7;
    int aoff[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) aoff[c] = c16 * kDim + 16 * ((g + 4 * c) ^ sw);
    const int xoff = kStageRowBytes + 16 * g;

    constexpr int kDmaPerWave = 16 / NW;          // LDS-DMA instructions a wave issues per stage (+1 aux on the last wave)
    // SELF: the 32-row unit of the reduced bank in which this wave's own rows start
    const int own_unit = SELF ? __builtin_amdgcn_readfirstlane(cb >> 5) : 0;

    // TRI: -|c|^2 of this lane's output rows (an output row beyond the bank takes part in nothing), the end of
    // the chunk's own stages (the diagonal block: both directions come out of the row direction there), and the
    // bound / norm words of the streamed rows two stages ahead (waves 0 and 1: one row per lane; they travel by LDS-DMA
    // into kTriRaw, not through registers: a loop-carried VGPR with a load in flight was copied by the compiler in front
    // of the hand-over's wait -- tests/test_isa_hazards.py).
    int negcn[NC];
    const int diag_end = TRI ? (chunk + 1) * (16 * NC * NW / kStageRows) : 0;
    if constexpr (TRI) {
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int n = cb + 16 * j + c16;
            negcn[j] = (n < p.nred) ? -p.col_norm[n] : -(1 << 30);
        }
    }
    // X of stage `stg`'s row (64 wave + lane) from its bound and norm words, into that stage's buffer
    auto tri_store = [&](int stg, char* sbuf, int b, int nm) {
        const int lane = lane_now();
        const int rho = 64 * wave + lane;
        int X = (b < kTriAlways) ? kTriAlways : b - nm + (nm & 1);
        if (stg * kStageRows + rho >= p.nred || stg < diag_end) X = kTriNever;
#ifdef FM_ABLATE
        if (p.tri_nocol) X = kTriNever;
#endif
        *(int*)(sbuf + kTriXRow + 4 * rho) = X;
        // (lane exchanges addressed from `lane` above, not __shfl_xor: its lane number is hoisted out of the stage loop and spilled)
        int m8 = min(X, __builtin_amdgcn_ds_bpermute((lane ^ 1) << 2, X));
        m8 = min(m8, __builtin_amdgcn_ds_bpermute((lane ^ 2) << 2, m8));
        m8 = min(m8, __builtin_amdgcn_ds_bpermute((lane ^ 16) << 2, m8));
        if ((lane & 19) == 0) *(int*)(sbuf + kTriX8 + 64 * (rho >> 5) + 4 * (lane & 12)) = m8;
    };

    // Bounds published by the blocks that reduce other slices for the same output rows:
    // bound[n] is the K-th best hi some block has reached, so the final K-th best is
    // >= bound[n] and a candidate with hi < bound[n] can be dropped.  Stale values are
    // merely weaker bounds, so relaxed agent-scope loads suffice; the comparison is
    // non-strict (hi >= bound) because the owner of the bound may have a higher index.
    // The loads for stage s+1 are issued during stage s (next to the LDS-DMA prefetch) and
    // consumed after the wait that retires that prefetch, so they never stall the wave.
    int gnext[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int n = cb + 16 * j + c16;
        gnext[j] = (p.bound && n < p.ncols_alloc)
            ? __hip_atomic_load(bound_thr + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : INT32_MIN;
    }
    // TRI: the refreshed bounds travel by LDS-DMA into kTriGn (no loop-carried register has a load in flight: the compiler
    // moved such registers between the copies of the unrolled stage loop in front of the wait, tests/test_isa_hazards.py);
    // the words read here apply at once
    bool gn_fresh = false;            // (uniform) the previous hand-over requested a refresh
    if constexpr (TRI) {
#pragma unroll
        for (int j = 0; j < NC; ++j) thr[j] = max(thr[j], gnext[j] >> 1);
        if (wave < 2) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (st0 + k < st1) {
                    const int m = (st0 + k) * kStageRows + 64 * wave + lane;      // (< nred_pad: the arrays cover it)
                    const int b = __hip_atomic_load(p.bound + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    tri_store(st0 + k, smem + k * kStride, b, p.col_norm[m]);
                }
            }
        }
    }
    // prologue prefetch (issued AFTER the bound loads: vmcnt retires in order, so "at most the newest
    // stage's DMA outstanding" implies that the bound loads and every older DMA have landed)
    if constexpr (SW) {
        if (st0 < st1) issue_stage_u8(p, st0, smem, wave, lane, dma_lo);
        if constexpr (NBUF == 3) { if (st0 + 1 < st1) issue_stage_u8(p, st0 + 1, smem + kStride, wave, lane, dma_lo); }
    } else {
        if (st0 < st1) issue_stage<GLDS, NW>(p, st0, smem, wave, lane);
        if constexpr (NBUF == 3) {
            if (st0 + 1 < st1) issue_stage<GLDS, NW>(p, st0 + 1, smem + kStride, wave, lane);
        }
    }

    // One pipeline step on LDS buffer BUF (compile-time, so every ds_read address is
    // base register + immediate); the stage loop below is unrolled by two.
    auto stage = [&](auto buf_tag, int st) {
        constexpr int BUF = decltype(buf_tag)::value;
        char* buf = smem + BUF * kStride;
        if constexpr (NBUF == 3) {
            // stage st's DMA was issued two hand-overs ago; only the DMA of stage st + 1 (the newest
            // VMEM operations of this wave, unless an exact path published a bound since) may still
            // be in flight.  The bound loads (gnext) are older than that DMA, so they have landed too.
            if (st + 1 < st1) {
                if (wave == NW - 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kDmaPerWave + 1) : "memory");
                else                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kDmaPerWave) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            // (volatile asm statements keep their order: the bound registers count as written here,
            // behind the wait, so no use of them can be scheduled in front of it)
            if constexpr (!TRI) {
#pragma unroll
                for (int j = 0; j < NC; ++j) asm volatile("" : "+v"(gnext[j]));
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else {
            if constexpr (GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();   // stage st landed; every wave is done with the other buffer
        }
        if constexpr (TRI) {
            if (gn_fresh) {            // (the words are this wave's own request: landed behind the wait above)
                const int* gw = (const int*)(smem + kTriGn + 1024 * wave) + lane_now();
#pragma unroll
                for (int j = 0; j < NC; ++j) thr[j] = max(thr[j], gw[64 * j] >> 1);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NC; ++j) thr[j] = max(thr[j], gnext[j] >> 1);      // hi >= g possible iff acc >= floor(g / 2)
        }
        if constexpr (KTOP == 2) {
            if (p.bound) {
#pragma unroll
                for (int j = 0; j < NC; ++j) {
                    if (__builtin_amdgcn_ballot_w64(ret1[j] != INT32_MIN) != 0ull) {       // (rare after the first stages)
                        const int c = min(ret1[j], top[j].hi(0));
                        const int n = cb + 16 * j + c16;
                        if (ret1[j] != INT32_MIN && c > gnext[j] && n < p.ncols_alloc)
                            __hip_atomic_fetch_max(bound_thr + n, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ret1[j] = INT32_MIN;
                    }
                }
            }
        }
        if constexpr (NBUF == 2) {
            if constexpr (SW) { if (st + 1 < st1) issue_stage_u8(p, st + 1, smem + (BUF ^ 1) * kStride, wave, lane, dma_lo); }
            else if (st + 1 < st1) issue_stage<GLDS, NW>(p, st + 1, smem + (BUF ^ 1) * kStride, wave, lane);
        }
        if constexpr (TRI) {
            // the words requested at the previous hand-over (LDS-DMA older than the one that wait left in flight, so they
            // have landed; the barrier covers wave 0 <-> 1 only in that each reads what it requested itself) become stage
            // st + 1's X in its buffer, which every wave left two barriers ago and the row DMA fills beside it; then the
            // words of stage st + 2 are requested, in front of its row DMA
            if (wave < 2) {
                if (st > st0 && st + 1 < st1) {
                    const int l = lane_now();
                    const int b = *(const int*)(smem + kTriRaw + 4 * (64 * wave + l));
                    const int nm = *(const int*)(smem + kTriRaw + 512 + 4 * (64 * wave + l));
                    tri_store(st + 1, smem + ((BUF + 1) % 3) * kStride, b, nm);
                }
                if (st + 2 < st1) {
                    const unsigned raw = (unsigned)(size_t)LDS_PTR(smem) + (unsigned)kTriRaw + 256u * (unsigned)wave;
                    const unsigned voff = (unsigned)lane_now() * 4u;
                    const size_t m0 = (size_t)(st + 2) * kStageRows + 64 * wave;
                    lds_dma_4<true>(raw, p.bound + m0, voff);
                    lds_dma_4<false>(raw + 512u, p.col_norm + m0, voff);
                }
            }
        }
        // (a bound that is a few stages old is merely weaker; late in a sweep the bounds hardly move, and each of these loads
        // is an agent-scope read that goes past the XCD's L2: 5e6 of them per 100k x 100k pair were 3/4 of the kernel's
        // fabric traffic -- r04: re-read at every stage only while the sweep is young)
        if constexpr (TRI) {
            gn_fresh = st + 1 < st1 && (st - st0 < 8 || ((st - st0) & p.bound_mask) == 0);
            if (gn_fresh) {
                const unsigned dst = (unsigned)(size_t)LDS_PTR(smem) + (unsigned)kTriGn + 1024u * (unsigned)wave;
                const unsigned voff = (unsigned)(lane_now() & 15) * 4u;
#pragma unroll
                for (int j = 0; j < NC; ++j) lds_dma_4<true>(dst + 256u * j, p.bound + cb + 16 * j, voff);
            }
        } else
        if (p.bound && (st - st0 < 8 || ((st - st0) & p.bound_mask) == 0)) {
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const int n = cb + 16 * j + c16;
                if constexpr (NBUF == 3 && SW) {
                    const unsigned boff = (unsigned)(cb + c16) * 4u;
                    if (j == 0) load_word_untracked_s<0>(gnext[0], bound_thr, boff);
                    if (j == 1) load_word_untracked_s<64>(gnext[1], bound_thr, boff);
                    if (j == 2) load_word_untracked_s<128>(gnext[2], bound_thr, boff);
                    if (j == 3) load_word_untracked_s<192>(gnext[3], bound_thr, boff);
                } else if constexpr (NBUF == 3) {
                    // (ncols_alloc is a multiple of the chunk, so n is always inside the array)
                    load_bound_untracked(gnext[j], bound_thr + n);
                } else {
                    if (n < p.ncols_alloc)
                        gnext[j] = __hip_atomic_load(bound_thr + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        if constexpr (NBUF == 3) {
            // (after the bound loads, see the prologue) into the buffer every wave left at the barrier above
            if constexpr (SW) { if (st + 2 < st1) issue_stage_u8(p, st + 2, smem + ((BUF + 2) % 3) * kStride, wave, lane, dma_lo); }
            else if (st + 2 < st1) issue_stage<GLDS, NW>(p, st + 2, smem + ((BUF + 2) % 3) * kStride, wave, lane);
        }

#pragma unroll
        for (int u = 0; u < kStageRows / kTileRows; ++u) {          // 32-row units
            v4i acc[2][NC];
            if constexpr (PRIO != 0) __builtin_amdgcn_s_setprio(2);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const char* rows = buf + (32 * u + 16 * s) * kDim;
                const v4i af0 = *(const v4i*)(rows + aoff[0]);
                const v4i af1 = *(const v4i*)(rows + aoff[1]);
                const v4i ci  = *(const v4i*)(buf + xoff + u * (kAuxPerTile * 4) + s * 128);
#pragma unroll
                for (int j = 0; j < NC; ++j) acc[s][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af0, bf[j][0], ci, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NC; ++j) acc[s][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af1, bf[j][1], acc[s][j], 0, 0, 0);
            }
            if constexpr (PRIO != 0) __builtin_amdgcn_s_setprio(0);
            if constexpr (SELF) {
                // unit d of the wave's own 16 NC rows: its tiles s = 0, 1 face the blocks j = 2 d + s, and lane
                // (c16, g) holds the pair (row 4 g + reg of the tile, output row c16): the diagonal is reg = c16 & 3
                // of lane group c16 >> 2
                // (own_unit is an SGPR: a scalar compare and branch per unit, nothing on the vector ALU)
                const unsigned d = (unsigned)(st * (kStageRows / kTileRows) + u - own_unit);
                if (d < (unsigned)(NC / 2)) {
                    const bool dl = g == (c16 >> 2);
#pragma unroll
                    for (int j = 0; j < NC; ++j) {
                        if ((unsigned)(j >> 1) == d) {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                acc[j & 1][j][r] = (dl && (c16 & 3) == r) ? kSelfMasked : acc[j & 1][j][r];
                        }
                    }
                }
            }
            int tmax[NC];
            bool any = false;
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const int m0 = max(max(acc[0][j][0], acc[0][j][1]), acc[0][j][2]);
                const int m1 = max(max(acc[0][j][3], acc[1][j][0]), acc[1][j][1]);
                tmax[j] = max(max(max(acc[1][j][2], acc[1][j][3]), m0), m1);
                any |= tmax[j] >= thr[j];
            }
            if constexpr (TRI) {
                // column direction, fast path: can any of this lane's 8 rows gain from its output row?  2 acc - |c|^2 >= X8
                const int x8v = *(const int*)(buf + xoff + (kTriX8 - kStageRowBytes) + 64 * u);
                bool anyc = false;
#pragma unroll
                for (int j = 0; j < NC; ++j) anyc |= ((tmax[j] << 1) + negcn[j]) >= x8v;
                if (__builtin_amdgcn_ballot_w64(any || anyc) != 0ull) {
                    if (__builtin_amdgcn_ballot_w64(any) != 0ull) {
                        const v4i low0 = *(const v4i*)(buf + xoff + u * (kAuxPerTile * 4) + 64);
                        const v4i low1 = *(const v4i*)(buf + xoff + u * (kAuxPerTile * 4) + 128 + 64);
#pragma unroll
                        for (int j = 0; j < NC; ++j) {
                            if (__builtin_amdgcn_ballot_w64(tmax[j] >= thr[j]) != 0ull) {
#ifdef FM_COUNT_VISITS
                                if (lane == 0) atomicAdd(&g_visits[0], 1ull);
#endif
                                // (only the value is kept: top[j].unit is dead code here)
                                const bool improved = top[j].update(acc[0][j], acc[1][j], low0, low1, 0);
                                const int h = top[j].key[0] >> 4;
                                thr[j] = max(thr[j], (h + 1) >> 1);
                                if (improved) {        // (cb + 16 j + c16 < ncols_alloc: whole chunks)
                                    const unsigned boff = (unsigned)(cb + c16) * 4u;
                                    if (j == 0) atomic_max_s<0>(p.bound, boff, h);
                                    if (j == 1) atomic_max_s<64>(p.bound, boff, h);
                                    if (j == 2) atomic_max_s<128>(p.bound, boff, h);
                                    if (j == 3) atomic_max_s<192>(p.bound, boff, h);
                                }
                            }
                        }
                    }
                    if (__builtin_amdgcn_ballot_w64(anyc) != 0ull) {
                        // exact, row by row: B = 2 (acc - cinit) + 1 - |c|^2 beats bound[m] iff 2 acc - |c|^2 >= X[m]
                        char* const xrow = buf + xoff + (kTriXRow - kStageRowBytes) + 128 * u;
                        const v4i xr0 = *(const v4i*)xrow, xr1 = *(const v4i*)(xrow + 64);
                        const v4i ci0 = *(const v4i*)(buf + xoff + u * (kAuxPerTile * 4));
                        const v4i ci1 = *(const v4i*)(buf + xoff + u * (kAuxPerTile * 4) + 128);
                        const unsigned browoff = (unsigned)(st * kStageRows + 32 * u + 4 * g) * 4u;
#pragma unroll
                        for (int j = 0; j < NC; ++j) {
                            if (__builtin_amdgcn_ballot_w64(((tmax[j] << 1) + negcn[j]) >= x8v) != 0ull) {
#ifdef FM_COUNT_VISITS
                                if (lane == 0) atomicAdd(&g_visits[1], 1ull);
#endif
                                // (most visits end here: the lane's best row cleared the loosest of its 8 bounds, none its own)
                                bool hit = false;
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    hit |= ((acc[0][j][r] << 1) + negcn[j]) >= xr0[r];
                                    hit |= ((acc[1][j][r] << 1) + negcn[j]) >= xr1[r];
                                }
                                if (__builtin_amdgcn_ballot_w64(hit) == 0ull) continue;
                                const bool lane0 = (lane_now() & 15) == 0;
#pragma unroll
                                for (int sr = 0; sr < 8; ++sr) {
                                    const int s = sr >> 2, r = sr & 3;
                                    const int v = (acc[s][j][r] << 1) + negcn[j];
                                    const int xr = s ? xr1[r] : xr0[r];
                                    if (__builtin_amdgcn_ballot_w64(v >= xr) != 0ull) {
                                        // one atomic per streamed row: the best of the 16 output rows that face it
                                        const int best = rowmax16(v >= xr ? v : INT32_MIN);
                                        if (lane0 && best >= xr) {
                                            const int ci = s ? ci1[r] : ci0[r];
                                            const int B = best - 2 * ci + 1;
                                            switch (sr) {     // (unrolled: the offset is an immediate)
                                            case 0: atomic_max_s<0>(p.bound, browoff, B); break;
                                            case 1: atomic_max_s<4>(p.bound, browoff, B); break;
                                            case 2: atomic_max_s<8>(p.bound, browoff, B); break;
                                            case 3: atomic_max_s<12>(p.bound, browoff, B); break;
                                            case 4: atomic_max_s<64>(p.bound, browoff, B); break;
                                            case 5: atomic_max_s<68>(p.bound, browoff, B); break;
                                            case 6: atomic_max_s<72>(p.bound, browoff, B); break;
                                            default: atomic_max_s<76>(p.bound, browoff, B); break;
                                            }
                                            __hip_atomic_fetch_max((int*)xrow + 16 * s + r, best + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifdef FM_COUNT_VISITS
                                            atomicAdd(&g_visits[2], 1ull);
#endif
                                        }
                                    }
                                }
                            }
                        }
                    }
                }
            } else
            if (__builtin_amdgcn_ballot_w64(any) != 0ull) {
                const v4i low0 = *(const v4i*)(buf + xoff + u * (kAuxPerTile * 4) + 64);
                const v4i low1 = *(const v4i*)(buf + xoff + u * (kAuxPerTile * 4) + 128 + 64);
#pragma unroll
                for (int j = 0; j < NC; ++j) {
                    if (__builtin_amdgcn_ballot_w64(tmax[j] >= thr[j]) != 0ull) {
#ifdef FM_COUNT_VISITS
                        {   // [split]: visits, [64 + split]: lanes of those visits whose own candidates reach their threshold
                            const unsigned long long wm = __builtin_amdgcn_ballot_w64(tmax[j] >= thr[j]);
                            if (lane == 0) { atomicAdd(&g_visits[split & 63], 1ull); atomicAdd(&g_visits[64 + (split & 63)], (unsigned long long)__popcll(wm)); }
                        }
#endif
                        const int best_before = top[j].key[0];
                        const bool improved = top[j].update(acc[0][j], acc[1][j], low0, low1,
                                                            st * (kStageRows / kTileRows) + u);
                        thr[j] = max(thr[j], top[j].own_threshold());
                        if (p.bound && improved && top[j].full()) {
                            const int n = cb + 16 * j + c16;
                            if (n < p.ncols_alloc) {
                                __hip_atomic_fetch_max(bound_thr + n, top[j].kth_hi(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if constexpr (KTOP == 2) {
                                    if (top[j].key[0] != best_before)
                                        ret1[j] = __hip_atomic_fetch_max(p.bound + n, top[j].hi(0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                }
                            }
                        }
                    }
                }
            }
        }
    };

#ifdef FM_CLOCK_STAMP
    const unsigned long long ck0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    if constexpr (NBUF == 3) {
        for (int st = st0; st < st1; st += 3) {
            stage(std::integral_constant<int, 0>{}, st);
            if (st + 1 < st1) stage(std::integral_constant<int, 1>{}, st + 1);
            if (st + 2 < st1) stage(std::integral_constant<int, 2>{}, st + 2);
        }
    } else {
        for (int st = st0; st < st1; st += 2) {
            stage(std::integral_constant<int, 0>{}, st);
            if (st + 1 < st1) stage(std::integral_constant<int, 1>{}, st + 1);
        }
    }

#ifdef FM_CLOCK_STAMP
    if (tid == 0) {
        const unsigned long long ck1 = __builtin_amdgcn_s_memtime(), rt1 = __builtin_amdgcn_s_memrealtime();
        g_clock[2 * (blockIdx.x & 8191)] = ck1 - ck0;
        g_clock[2 * (blockIdx.x & 8191) + 1] = rt1 - rt0;
    }
#endif
#ifdef FM_COUNT_VISITS
    if (lane == 0) atomicAdd(&g_visits[128 + (split & 127)], (unsigned long long)(st1 - st0) * 4 * NC);
#endif
    // Merge the four lane groups (same output row, interleaved reduced rows), then emit.
    // (TRI: every improvement went to bound[], which IS the result)
    if constexpr (!TRI)
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        int bh[KTOP], bi[KTOP];
#pragma unroll
        for (int k = 0; k < KTOP; ++k) {
            bh[k] = top[j].hi(k);
            bi[k] = (top[j].unit[k] >= 0) ? top[j].index(k, g) : -1;
            if (bi[k] >= p.nred) bi[k] = -1;          // a padding row is not a candidate
            if constexpr (SELF) { if (bi[k] == cb + 16 * j + c16) bi[k] = -1; }   // (cannot win: 7 unmasked rows share its unit)
        }
#pragma unroll
        for (int mask = 16; mask <= 32; mask <<= 1) {
            int oh[KTOP], oi[KTOP];
#pragma unroll
            for (int k = 0; k < KTOP; ++k) {
                oh[k] = __shfl_xor(bh[k], mask);
                oi[k] = __shfl_xor(bi[k], mask);
            }
            if constexpr (KTOP == 1) {
                const bool mine = !better(oh[0], oi[0], bh[0], bi[0]);
                bh[0] = mine ? bh[0] : oh[0];
                bi[0] = mine ? bi[0] : oi[0];
            } else {
                const bool m0 = !better(oh[0], oi[0], bh[0], bi[0]);
                const int r0h = m0 ? bh[0] : oh[0], r0i = m0 ? bi[0] : oi[0];
                // runner-up: the loser of the first comparison against the winner side's 2nd
                const int ah = m0 ? bh[1] : bh[0], ai = m0 ? bi[1] : bi[0];
                const int ch = m0 ? oh[0] : oh[1], cidx = m0 ? oi[0] : oi[1];
                const bool m1 = !better(ch, cidx, ah, ai);
                bh[0] = r0h; bi[0] = r0i;
                bh[1] = m1 ? ah : ch;
                bi[1] = m1 ? ai : cidx;
            }
        }
        const int n = cb + 16 * j + c16;
        if (g == 0 && n < p.ncols_alloc) {
            const int cn = (n < p.ncols_pad) ? p.col_norm[n] : 0;
            unsigned long long* out = p.partial + ((size_t)split * p.ncols_alloc + n) * KTOP;
#pragma unroll
            for (int k = 0; k < KTOP; ++k) {
                unsigned long long key = ~0ull;
                if (bi[k] >= 0 && n < p.ncols_pad) {
                    const unsigned d2 = (unsigned)(cn + 1 - bh[k]);
                    key = ((unsigned long long)d2 << 32) | (unsigned)bi[k];
                }
                out[k] = key;
            }
        }
    }
}

template <int NC, int KTOP, bool GLDS, int NW, int NBUF = 2, int PRIO = 0, bool SELF = false>
__global__ __launch_bounds__(64 * NW, (NC >= 8 ? 2 : (NC >= 6 ? 3 : 4)))
void rowreduce_kernel(RRParams p)
{
    __shared__ __attribute__((aligned(16))) char smem[NBUF * kStageBytes];
    rowreduce_body<NC, KTOP, GLDS, NW, NBUF, PRIO, SELF>(p, (int)blockIdx.x, smem);
}

// Several bank pairs of ONE shape in one launch, pair after pair in block order: when the workgroups of
// pair i run out, the CUs they leave take workgroups of pair i + 1 at once -- between two separate
// launches the chip drains (the last of five rounds of workgroups finish at different times) and a
// launch gap follows, together ~4 % of a 100k x 100k launch.
// (r05, last: the pairs of a launch need not share a shape any more -- a dataset's images all differ in size, and pairs that
// fell out of the batched launch for that cost 7 % more per descriptor pair; pair i owns the workgroups
// [first_block[i], first_block[i + 1]), entries from n on hold INT32_MAX)
struct RRBatch {
    RRParams p[kRRBatchMax];
    int      n;
    int      first_block[kRRBatchMax + 1];
};
// the pair a workgroup belongs to: scalar compares on kernel arguments
__device__ __forceinline__ int batch_pair_of(const RRBatch& b, int bid)
{
    int pair = 0;
#pragma unroll
    for (int i = 1; i < kRRBatchMax; ++i) pair += bid >= b.first_block[i] ? 1 : 0;
    return pair;
}

template <int NC, int KTOP, int NW, int NBUF, int PRIO, bool SELF = false>
__global__ __launch_bounds__(64 * NW, 4)
void rowreduce_batch_kernel(RRBatch b)
{
    __shared__ __attribute__((aligned(16))) char smem[NBUF * kStageBytes];
    const int pair = batch_pair_of(b, (int)blockIdx.x);
    const RRParams p = b.p[pair];
    rowreduce_body<NC, KTOP, true, NW, NBUF, PRIO, SELF>(p, (int)blockIdx.x - b.first_block[pair], smem);
}

// The triangular self sweep (see kTriNever above): one bank, or up to kRRBatchMax banks of one padded size.
template <int PRIO>
__global__ __launch_bounds__(64 * 8, 4)
void rowreduce_tri_kernel(RRParams p)
{
    __shared__ __attribute__((aligned(16))) char smem[kTriLdsBytes];
    rowreduce_body<4, 1, true, 8, 3, PRIO, true, true>(p, (int)blockIdx.x, smem);
}

template <int PRIO>
__global__ __launch_bounds__(64 * 8, 4)
void rowreduce_tri_batch_kernel(RRBatch b)
{
    __shared__ __attribute__((aligned(16))) char smem[kTriLdsBytes];
    const int pair = batch_pair_of(b, (int)blockIdx.x);
    const RRParams p = b.p[pair];
    rowreduce_body<4, 1, true, 8, 3, PRIO, true, true>(p, (int)blockIdx.x - b.first_block[pair], smem);
}

// (plan_tri -- the workgroup table of the two launches -- is host code and lives in api_grid.hip beside the cell planner)
hipError_t launch_rowreduce_tri(int n, const Bank* const* banks, const TriPlan* plans, int* const* bound, bool prio, hipStream_t stream)
{
    if (n < 1 || n > kRRBatchMax) return hipErrorInvalidValue;
    RRBatch b;
    for (int i = 0; i < n; ++i) {
        const TriPlan& plan = plans[i];
        if (plan.npieces < 1 || plan.stages < 4) return hipErrorInvalidValue;
        RRParams& p = b.p[i];
        p = RRParams{};
        p.col_rows = banks[i]->rows8;  p.col_norm = banks[i]->norm;  p.ncols_pad = (int)banks[i]->n_pad;
        p.red_rows = banks[i]->rows8;  p.red_aux = banks[i]->aux;    p.nred = (int)banks[i]->n;
        p.nstages = (int)(banks[i]->n_pad / kStageRows);
        p.nsplit = 1;  p.nchunks = plan.nchunks;  p.stages_per_split = p.nstages;  p.ncols_alloc = plan.ncols_alloc;
        p.partial = nullptr;  p.bound = bound[i];  p.order = 0;  p.bound_mask = plan.bound_every > 1 ? plan.bound_every - 1 : 0;
        p.tri_first = 0;  p.tri_S = plan.stages;
#ifdef FM_ABLATE          // (measurement builds: no variable of the environment changes what the product library computes)
        p.tri_nocol = getenv("FM_TRI_NOCOL") ? 1 : 0;
#endif
    }
    // (one instantiation, with the s_setprio around the MFMA burst: the one without it does not fit 128 VGPRs)
    (void)prio;
#ifdef FM_ABLATE
    const bool merge = getenv("FM_TRI_MERGE") != nullptr;       // (measurement: both phases in ONE grid, diagonal blocks first)
#else
    constexpr bool merge = false;
#endif
    for (int phase = 0; phase < 2; ++phase) {
        if (merge && phase == 1) break;
        // the banks of a launch may differ in size (r05, last): bank i under its own plan, its workgroups
        // [first_block[i], first_block[i + 1]) of the grid
        long long total = 0;
        for (int i = 0; i < n; ++i) {
            const TriPlan& plan = plans[i];
            int first = phase == 0 ? 0 : plan.ndiag, count = phase == 0 ? plan.ndiag : plan.npieces - plan.ndiag;
            if (merge) { first = 0; count = plan.npieces; }
            b.p[i].tri_first = first;
            b.first_block[i] = (int)total;
            total += count > 0 ? count : 0;
        }
        if (total <= 0) continue;
        if (total > INT32_MAX) return hipErrorInvalidValue;
        if (n == 1) {
            hipLaunchKernelGGL((rowreduce_tri_kernel<1>), dim3((unsigned)total), dim3(512), 0, stream, b.p[0]);
        } else {
            for (int i = n; i < kRRBatchMax; ++i) b.p[i] = b.p[0];
            b.n = n;
            b.first_block[n] = (int)total;
            for (int i = n + 1; i <= kRRBatchMax; ++i) b.first_block[i] = INT32_MAX;
            hipLaunchKernelGGL((rowreduce_tri_batch_kernel<1>), dim3((unsigned)total), dim3(512), 0, stream, b);
        }
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

RowReducePlan plan_rowreduce(int64_t ncols_pad, int64_t nred_pad, const Tuning& tn)
{
    const int force_nb = tn.nb, force_nsplit = tn.nsplit, force_nw = tn.nw;
    RowReducePlan pl;
    pl.nbuf = (tn.nbuf == 2 || tn.nbuf == 3) ? tn.nbuf : 0;
    pl.prio = tn.prio;
    pl.order = (tn.k1_order >= 0 && tn.k1_order <= 2) ? tn.k1_order : 0;
    pl.bound_every = 1;
    for (int b = 2; b <= 1024; b <<= 1) if (tn.bound_every == b) pl.bound_every = b;      // (powers of two: the test is a mask)
    const int64_t nstages = nred_pad / kStageRows;
    // nb = blocks of 16 output rows per wave: 4 (64 rows, ~110 VGPRs, 4 waves/SIMD) by default,
    // 8 via FM_NB; nw = waves per workgroup sharing the staged tiles: 8 for big problems,
    // 4 for small ones so that enough workgroups exist.
    int nb = 4, nw = 8;
    if (ncols_pad < 512 * 64) nw = 4;
    if (force_nb == 4 || force_nb == 6 || force_nb == 8) nb = force_nb;
    if (force_nw == 4 || force_nw == 8 || force_nw == 16) nw = force_nw;
    pl.nb = nb;
    pl.nw = nw;
    const int cb = 16 * nb * nw;
    pl.nchunks = (int)((ncols_pad + cb - 1) / cb);
    if (pl.nchunks < 1) pl.nchunks = 1;
    pl.ncols_alloc = pl.nchunks * cb;
    // Just UNDER five rounds of the workgroups the chip holds (16 waves per CU): a grid slightly over
    // a whole number of rounds leaves most CUs idle for the run time of a workgroup at the end
    // (100k x 100k: 16 splits = 6.1 rounds 0.867 ms, 13 splits = 4.98 rounds 0.842 ms; K2 1.095 -> 1.041).
    const int64_t resident = 256 * (16 / nw);
    int64_t nsplit = (5 * resident) / pl.nchunks;
    if (nsplit > nstages / 8) nsplit = nstages / 8;   // keep >= 8 stages (1024 rows) per split
    if (nsplit < 1) nsplit = 1;
    if (force_nsplit > 0) nsplit = force_nsplit;
    if (nsplit > nstages) nsplit = nstages > 0 ? nstages : 1;
    int64_t per = (nstages + nsplit - 1) / nsplit;
    if (per < 1) per = 1;
    nsplit = (nstages + per - 1) / per;
    if (nsplit < 1) nsplit = 1;
    pl.nsplit = (int)nsplit;
    pl.stages_per_split = (int)per;
    return pl;
}

// stage buffers: 3 for the top-1 kernel (its DMA wait leaves the hand-over) and, since r05, for the top-2 kernel in its
// built-in shape (4 blocks per wave: the scalar-wave addressing, SW in rowreduce_body, freed the registers whose spills
// made the third buffer a loss -- A/B on one box: 1.051 -> 0.983 ms per 100k x 100k pair, profiles/r05f_k2_nbuf_ab.log);
// 2 for the other top-2 shapes; the plan (Tuning::nbuf) overrides
static int nbuf_choice(int ktop, int plan_nbuf, int nc = 4)
{
    if (plan_nbuf == 2 || plan_nbuf == 3) return plan_nbuf;
    return (ktop == 1 || nc == 4) ? 3 : 2;
}

// The masked-diagonal top-1 (fm_self_dist): 16 x 4 output rows per wave only (the unit test in the sweep
// assumes nothing else, but only this shape is built), 8 or 4 waves.
template <int NW>
static hipError_t launch_self_t(const RRParams& p, int grid, bool glds, int plan_nbuf, bool prio, hipStream_t stream)
{
    if constexpr (NW == 8) {
        if (glds && nbuf_choice(1, plan_nbuf) == 3) {
            if (prio) hipLaunchKernelGGL((rowreduce_kernel<4, 1, true, NW, 3, 1, true>), dim3(grid), dim3(64 * NW), 0, stream, p);
            else      hipLaunchKernelGGL((rowreduce_kernel<4, 1, true, NW, 3, 0, true>), dim3(grid), dim3(64 * NW), 0, stream, p);
            return hipGetLastError();
        }
    }
    if (glds) hipLaunchKernelGGL((rowreduce_kernel<4, 1, true, NW, 2, 0, true>), dim3(grid), dim3(64 * NW), 0, stream, p);
    else      hipLaunchKernelGGL((rowreduce_kernel<4, 1, false, NW, 2, 0, true>), dim3(grid), dim3(64 * NW), 0, stream, p);
    return hipGetLastError();
}

template <int NC, int KTOP, int NW>
static hipError_t launch_t(const RRParams& p, int grid, bool glds, int plan_nbuf, bool prio, hipStream_t stream)
{
    if constexpr (NW == 8) {
        // PRIO: s_setprio 2 while a wave issues a unit's 16 MFMAs as one burst, back to 0 for the
        // epilogue (A/B on one box: 0.897 -> 0.887 ms); Tuning::prio = 0 selects the variant without it.
        // (The two-buffer top-2 kernel gets 1 % slower with it: 1.034 -> 1.044 ms.)
        if (glds && nbuf_choice(KTOP, plan_nbuf, NC) == 3) {
            if (prio) hipLaunchKernelGGL((rowreduce_kernel<NC, KTOP, true, NW, 3, 1>), dim3(grid), dim3(64 * NW), 0, stream, p);
            else      hipLaunchKernelGGL((rowreduce_kernel<NC, KTOP, true, NW, 3, 0>), dim3(grid), dim3(64 * NW), 0, stream, p);
            return hipGetLastError();
        }
    }
    if (glds) hipLaunchKernelGGL((rowreduce_kernel<NC, KTOP, true, NW>), dim3(grid), dim3(64 * NW), 0, stream, p);
    else      hipLaunchKernelGGL((rowreduce_kernel<NC, KTOP, false, NW>), dim3(grid), dim3(64 * NW), 0, stream, p);
    return hipGetLastError();
}

template <int KTOP>
static hipError_t launch_k(const RRParams& p, int grid, int nb, int nw, bool glds, int plan_nbuf, bool prio, hipStream_t stream)
{
    if (nb == 6) {
        if (nw == 8) return launch_t<6, KTOP, 8>(p, grid, glds, plan_nbuf, prio, stream);
        return launch_t<6, KTOP, 4>(p, grid, glds, plan_nbuf, prio, stream);
    }
    if (nb == 8) {
        if (nw == 16) return launch_t<8, KTOP, 16>(p, grid, glds, plan_nbuf, prio, stream);
        if (nw == 8) return launch_t<8, KTOP, 8>(p, grid, glds, plan_nbuf, prio, stream);
        return launch_t<8, KTOP, 4>(p, grid, glds, plan_nbuf, prio, stream);
    }
    if (nw == 16) return launch_t<4, KTOP, 16>(p, grid, glds, plan_nbuf, prio, stream);
    if (nw == 8) return launch_t<4, KTOP, 8>(p, grid, glds, plan_nbuf, prio, stream);
    return launch_t<4, KTOP, 4>(p, grid, glds, plan_nbuf, prio, stream);
}

static void fill_params(RRParams& p, const Bank& cols, const Bank& red, const RowReducePlan& plan,
                        unsigned long long* partial, int* bound)
{
    p.bound = (plan.nsplit > 1) ? bound : nullptr;
    p.col_rows = cols.rows8;
    p.col_norm = cols.norm;
    p.ncols_pad = (int)cols.n_pad;
    p.red_rows = red.rows8;
    p.red_aux = red.aux;
    p.nred = (int)red.n;
    p.nstages = (int)(red.n_pad / kStageRows);
    p.nsplit = plan.nsplit;
    p.nchunks = plan.nchunks;
    p.stages_per_split = plan.stages_per_split;
    p.ncols_alloc = plan.ncols_alloc;
    p.partial = partial;
    p.order = plan.order;
    p.bound_mask = plan.bound_every > 1 ? plan.bound_every - 1 : 0;
}

int rowreduce_grid(const RowReducePlan& plan) { return blocks_per_pair(plan.nchunks, plan.nsplit, plan.order); }

// Plan of the masked-diagonal sweep of a bank against itself: the built-in shape (4 blocks per wave, 8 or 4 waves)
// whatever the "nb" / "nw" options say -- only that shape is built for it.
RowReducePlan plan_rowreduce_self(int64_t n_pad, const Tuning& tn)
{
    Tuning t = tn;
    t.nb = 0;
    if (t.nw != 4 && t.nw != 8) t.nw = 0;
    return plan_rowreduce(n_pad, n_pad, t);
}

// Top-1 row-reduce of n <= kRRBatchMax bank pairs in one launch, pair i under plans[i] (any sizes; every plan in the shape the
// batched kernel is built for: 4 blocks per wave, 8 waves).
hipError_t launch_rowreduce_batch(int n, const Bank* const* cols, const Bank* const* red, const RowReducePlan* plans,
                                  unsigned long long* const* partial, int* const* bound, hipStream_t stream, bool self)
{
    if (n < 1 || n > kRRBatchMax) return hipErrorInvalidValue;
    RRBatch b;
    long long total = 0;
    for (int i = 0; i < n; ++i) {
        if (plans[i].nb != 4 || plans[i].nw != 8) return hipErrorInvalidValue;
        fill_params(b.p[i], *cols[i], *red[i], plans[i], partial[i], bound[i]);
        b.first_block[i] = (int)total;
        total += rowreduce_grid(plans[i]);
    }
    if (total > INT32_MAX) return hipErrorInvalidValue;
    for (int i = n; i < kRRBatchMax; ++i) b.p[i] = b.p[0];
    b.first_block[n] = (int)total;
    for (int i = n + 1; i <= kRRBatchMax; ++i) b.first_block[i] = INT32_MAX;
    b.n = n;
    if (self) hipLaunchKernelGGL((rowreduce_batch_kernel<4, 1, 8, 3, 1, true>), dim3((unsigned)total), dim3(64 * 8), 0, stream, b);
    else      hipLaunchKernelGGL((rowreduce_batch_kernel<4, 1, 8, 3, 1>), dim3((unsigned)total), dim3(64 * 8), 0, stream, b);
    return hipGetLastError();
}

// Top-1 of every row of `bank` over the OTHER rows of the same bank (plan from plan_rowreduce_self).
hipError_t launch_rowreduce_self(const Bank& bank, const RowReducePlan& plan, unsigned long long* partial, int* bound,
                                 bool use_glds, hipStream_t stream)
{
    if (plan.nb != 4 || (plan.nw != 4 && plan.nw != 8)) return hipErrorInvalidValue;
    RRParams p;
    fill_params(p, bank, bank, plan, partial, bound);
    const int grid = rowreduce_grid(plan);
    return plan.nw == 8 ? launch_self_t<8>(p, grid, use_glds, plan.nbuf, plan.prio != 0, stream)
                        : launch_self_t<4>(p, grid, use_glds, plan.nbuf, plan.prio != 0, stream);
}

hipError_t launch_rowreduce(const Bank& cols, const Bank& red, int ktop, const RowReducePlan& plan,
                            unsigned long long* partial, int* bound, bool use_glds, hipStream_t stream)
{
    RRParams p;
    fill_params(p, cols, red, plan, partial, bound);
    const int grid = rowreduce_grid(plan);
    return ktop == 1 ? launch_k<1>(p, grid, plan.nb, plan.nw, use_glds, plan.nbuf, plan.prio != 0, stream)
                     : launch_k<2>(p, grid, plan.nb, plan.nw, use_glds, plan.nbuf, plan.prio != 0, stream);
}

}  // namespace fm
