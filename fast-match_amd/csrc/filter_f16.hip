// K8 -- fp16 MFMA filter + exact float32 rescoring for the float32 route.
//
// dist_f32.hip (K5) evaluates OpenCV's float32 L2 chain for every
// pair on the vector ALUs.  Only K (1 or 2) rows per output row survive the reduce, so this
// file finds, on the matrix cores, a small superset of the rows that can survive, and then
// evaluates the exact chain for those rows only.  The result is bit-identical to K5's.
//
//   filter_kernel   approximate A(c,m) = |c|^2 + |m|^2 - 2 c.m with c.m computed by
//                   v_mfma_f32_16x16x32_f16 on the rows rounded to fp16 (4 per 16 x 16 x 128
//                   tile, float32 accumulate).  Every bank is stored scaled by its own power
//                   of two 2^k so that its largest magnitude lies in [2^13, 2^14): exact, and
//                   it keeps fp16 away from overflow and (relative to the bank) from underflow.
//                   The accumulator starts at -|m|^2/2, so acc = c.m - |m|^2/2 (in units of
//                   2^(kc+km)) and a LARGER acc is a SMALLER distance, as in rowreduce.hip.
//                   |A - D| <= M for the exact chain value D, with M = 1.1 * 2^-10 *
//                   (|c|^2 + max|m|^2): fp16 rounding (unit roundoff 2^-11) of both factors
//                   gives <= 2.0005 * 2^-11 |c||m| <= 2^-11 (|c|^2 + |m|^2) on c.m, twice that
//                   on A; the float32 accumulation, K5's own chain error (2^-17), flushed
//                   fp16 subnormals (2^-20) and the sqrt-tie slack (2^-21) fit in the 10 %.
//                   Each lane keeps the P = 4 largest acc of its share of the rows, and a row
//                   is examined only if acc >= (K-th best acc known for this output row) - M;
//                   that bound is shared between lane groups, waves and blocks through
//                   bound[n] as in K1.  At the end every lane emits its P entries.
//   rescore_kernel  16 or 64 lanes per output row: exact chain s = fmaf(v, v, s), k ascending,
//                   sqrtf, for the emitted rows with acc >= (final K-th best acc) - M; top-K of
//                   (distance bits, index) -> the same packed keys K5 writes.  A row that
//                   belongs to the exact top-K has acc >= bound - M; it can be missing from its
//                   lane's entries only if P rows of that lane do: that case (the lane's P-th
//                   entry inside the margin) sends the output row to rescan_kernel (a full exact
//                   scan of that row; up to 256 rows), or, beyond that, the whole call to K5.
//
// Layout: rowsh [n_pad][128] fp16 (256 B per row); auxf [n_pad] float32 = -|m|^2/2 of the
// scaled row (padding rows: -3.4e38, below every real accumulator value).
// Staging: 128 rows (32 KiB) + 512 B aux per step by global_load_lds_dwordx4, double buffered
// (LDS: rows of buffer 0 | rows of buffer 1 | aux 0 | aux 1, so that the buffers alternate by an XOR on
// the address registers and the stage loop is a real loop); the 16-byte chunk index of a row is XORed
// with (row & 15) on the source side so that the ds_read_b128 fragment reads of 16 consecutive rows
// hit 16 different bank groups.
#include "tile_ops.h"
#include <type_traits>
#include <stdlib.h>

namespace fm {

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float  v4f  __attribute__((ext_vector_type(4)));

constexpr int   kFRowBytes      = 256;
constexpr int   kFStageRows     = 128;
constexpr int   kFStageRowBytes = kFStageRows * kFRowBytes;      // 32768
constexpr int   kFStageBytes    = kFStageRowBytes + kFStageRows * 4;   // + aux
constexpr int   kFP             = 4;                             // entries per lane and output row
constexpr float kFEmpty         = -3.0e38f;                      // acc of an empty entry
constexpr int   kFMaxRescan     = 256;                           // output rows rescan_kernel can take

struct FParams {
    const char*  col_rows;     // fp16 rows of the output rows
    const float* col_norm;
    int          ncols;        // real output rows
    int          ncols_pad;
    const char*  red_rows;
    const float* red_aux;
    int          nred;
    int          nstages;      // nred_pad / kFStageRows
    int          nsplit;
    int          nchunks;
    int          stages_per_split;
    int          ncols_alloc;
    float        eps_c;        // eps * 2^(km-kc): times the stored |c|^2 = eps |c|^2 in acc units
    float        eps_nm;       // eps * max |m|^2 of the reduced bank, in acc units
    float        aux_mul;      // 2^(kc-km): stored accumulator inits -> acc units
    unsigned long long* slots; // [nsplit][ncols_alloc][4][kFP]  (acc bits << 32 | row), ~0 = none
    int*         bound;        // [2][ncols_alloc] ordered-int images: best acc, 2nd best acc (see below)
    int*         flag;
    // nsplit == 1: the wave that owns an output row has seen all of its reduced rows, so it
    // rescores its own entries at the end instead of emitting them for rescore_kernel
    int          fused;
    const float* col_rowsf;
    const float* red_rowsf;
    unsigned long long* partial;   // [n][KTOP] packed keys (split 0 of the caller's layout)
    int          bound_mask;    // the shared bounds are re-read at every stage of a sweep's first 8, then at
                                // the stages whose number & bound_mask == 0
    // TRI (float32 triangular self sweep, below): workgroup list entry = tri_first + blockIdx.x, piece length tri_S;
    // per-row candidate lists cand[row][kFTriCap] with their counters cnt[row] (bit 30: the list may be incomplete)
    int          tri_first, tri_S;
    unsigned long long* cand;
    int*         cnt;
};

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// float <-> int with the same order (an involution)
__device__ __forceinline__ int fmap(float f)
{
    const int i = __float_as_int(f);
    return i ^ ((i >> 31) & 0x7fffffff);
}
__device__ __forceinline__ float fmax3(float a, float b, float c)
{
    return __builtin_elementwise_maximum(__builtin_elementwise_maximum(a, b), c);
}
__device__ __forceinline__ float funmap(int i) { return __int_as_float(i ^ ((i >> 31) & 0x7fffffff)); }

constexpr int kFAuxBase = 2 * kFStageRowBytes;       // LDS offset of the aux words (512 B per buffer)

// LDS-DMA of one stage: 32 pieces of 4 rows (1 KiB) + the aux words.  Scalar base + ONE 32-bit lane offset per piece and
// the LDS destination in M0 straight from scalars (`wave` must be wave-uniform): the builtin's selection keeps a 64-bit
// address pair per piece in VGPRs, which this kernel does not have.  Source chunk (lane & 15) ^ (row & 15) of row
// 4 g + (lane >> 4) lands at chunk position lane & 15; (row & 15) = 4 (g & 3) | (lane >> 4), so a piece's lane offsets
// are dma_lo ^ 64 (g & 3).
// (M0 is a reserved register to the compiler and not accepted as a clobber: every use here sets it first, and the kernel
// uses no builtin that reads it.)
__device__ __forceinline__ void f_lds_dma_16(unsigned lds_addr, const void* sbase, unsigned voff)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(lds_addr), "v"(voff), "s"(sbase) : "memory");
}

template <int NW>
__device__ __forceinline__ void f_issue_stage(const FParams& p, int stage, char* smem, int buf, int wave, int lane)
{
    const char* src_rows = p.red_rows + (size_t)stage * kFStageRowBytes;
    const unsigned dma_lo = 256u * (unsigned)(lane >> 4) + 16u * (unsigned)((lane & 15) ^ (lane >> 4));
    const unsigned lds0 = (unsigned)(size_t)LDS_PTR(smem);
    constexpr int kPieces = (kFStageRows / 4) / NW;    // pieces per wave
#pragma unroll
    for (int i = 0; i < kPieces; ++i) {
        const int g = wave * kPieces + i;
        unsigned vo;
        asm volatile("v_xor_b32 %0, %1, %2" : "=v"(vo) : "s"(64 * (g & 3)), "v"(dma_lo));
        f_lds_dma_16(lds0 + (unsigned)buf * kFStageRowBytes + 1024u * (unsigned)g, src_rows + 1024 * g, vo);
    }
    if (wave == NW - 1 && lane < kFStageRows / 4)
        f_lds_dma_16(lds0 + kFAuxBase + 512u * (unsigned)buf, p.red_aux + (size_t)stage * kFStageRows, 16u * (unsigned)lane);
}

// SELF (K = 1): both banks are the same bank and the pair (n, n) is masked -- the top-1 over the OTHER rows,
// which is what Metric_Cache keeps of the self 2-NN (cache.pyx:250-252, 271-273; rowreduce.hip has the int8 form).
// TRI (r06): the float32 form of rowreduce.hip's triangular self sweep.  d(i, j) = d(j, i) bit for bit
// in K5's chain (v = a - b or b - a, then v * v), so the distances of a bank against itself need computing once: an
// output chunk sweeps only the stages from its own rows on (launch A: the diagonal blocks, masked, row direction only;
// launch B: the pieces beyond, piece-number major -- plan_tri / tri_entry as for the integer banks) and every tile of
// launch B serves BOTH of its rows:
//   row direction     as ever -- the lane's output row n keeps its P best streamed rows m in registers;
//   column direction  the streamed row m gains the candidate n.  In row m's own frame the pair is worth
//                     a' = m.c - |c|^2/2 = acc + |m|^2/2 - |c|^2/2   (acc = c.m - |m|^2/2 is what the lane holds),
//                     and it matters iff a' >= bound[m] - M_m, i.e.  acc - |c|^2/2 >= bound[m] - M_m - |m|^2/2 =: X[m].
//                     X[m] (from a bound read two stages ahead: a stale bound is merely weaker), |m|^2/2 and the minimum
//                     of X over the four rows a lane holds of a tile are staged in LDS beside the rows; the fast path is
//                     one subtract and one compare per block and tile on the tile maximum the row direction forms anyway.
//                     A pair that passes publishes a' to bound[m] (atomic maximum: both directions feed the same word)
//                     and appends (a', n) to row m's list in memory.  The filter is approximate, so -- unlike the
//                     integer sweep, where bound[] IS the result -- the candidates' identities are needed: the lists.
// At the end of a piece the lane's own entries that are inside the margin of their row's CURRENT bound go to the same
// lists; tri_rescore_kernel then evaluates a row's list exactly against its FINAL bound.  A list that overflows, or a lane
// whose fourth entry is inside the margin, sends the row to rescan_kernel (bit 30 of its counter), as the P-lists always did.
constexpr int kFTriCap   = 32;                      // entries of a row's candidate list
constexpr int kFTriRedo  = 1 << 30;                 // in cnt[row]: the list may be incomplete
constexpr int kFTriXBytes = 128 * 4 + 128 * 4 + 32 * 4;          // per staged stage: X[128], |m|^2/2 [128], X4[32]

template <int NC, int KTOP, int NW, bool SELF = false, bool TRI = false>
__global__ __launch_bounds__(64 * NW, 2)
void filter_kernel(FParams p)
{
    static_assert(!SELF || KTOP == 1, "the masked-diagonal sweep is a top-1");
    static_assert(!TRI || (SELF && NW == 8 && NC == 4), "the triangular sweep is built for one shape: 512-row chunks");
    __shared__ __attribute__((aligned(16))) char smem[2 * kFStageBytes + (TRI ? 3 * kFTriXBytes : 0)];

    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g    = lane >> 4;
    const int c16  = lane & 15;
    int chunk, split, st0, st1;
    if constexpr (TRI) {
        tri_entry(__builtin_amdgcn_readfirstlane(p.tri_first + (int)blockIdx.x), p.nchunks, p.nstages, p.tri_S, chunk, st0, st1);
        split = 0;
    } else {
        chunk = blockIdx.x % p.nchunks;              // split major, as in rowreduce.hip
        split = blockIdx.x / p.nchunks;
        st0 = split * p.stages_per_split;
        st1 = min(st0 + p.stages_per_split, p.nstages);
    }
    const int cb    = chunk * (16 * NC * NW) + wave * (16 * NC);
    // TRI: the chunk's own stages end here (the diagonal block: both directions come out of the row direction there)
    const int diag_end = TRI ? (chunk + 1) * (16 * NC * NW / kFStageRows) : 0;
    // the first stage is in flight while the stationary operand is loaded
    if (st0 < st1) f_issue_stage<NW>(p, st0, smem, 0, wave, lane);

    // Stationary operand: NC x 16 output rows, 4 K-steps of 32.
    v8h bh[NC][4];
    float marg[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int n = cb + 16 * j + c16;
        const bool ok = n < p.ncols_pad;
        marg[j] = ok ? fmaf(p.eps_c, p.col_norm[n], p.eps_nm) : 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            v4i h = v4i{0, 0, 0, 0};
            if (ok) h = *(const v4i*)(p.col_rows + (size_t)n * kFRowBytes + (4 * s + g) * 16);
            bh[j][s] = __builtin_bit_cast(v8h, h);
        }
    }

    // TRI: |c|^2 / 2 of the lane's output rows, and the words of the streamed rows two stages ahead (threads 0 .. 127:
    // one row each): bound, |m|^2, accumulator init
    float hc[NC];
    int   xb_next = 0;
    float xn_next = 0.f, xa_next = 0.f;
    char* const xbase = smem + 2 * kFStageBytes;          // three slots of kFTriXBytes: stage s uses slot s % 3
    auto x_request = [&](int stg) __attribute__((always_inline)) {
        if (tid < kFStageRows && stg < st1) {
            const int m = stg * kFStageRows + tid;                        // (< nred_pad: the arrays cover it)
            xb_next = __hip_atomic_load(p.bound + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            xn_next = p.col_norm[m];
            xa_next = p.red_aux[m];
        }
    };
    auto x_store = [&](int stg) __attribute__((always_inline)) {         // from the words requested for stage stg
        if (tid < kFStageRows && stg < st1) {
            const int m = stg * kFStageRows + tid;
            float X = funmap(xb_next) - fmaf(p.eps_c, xn_next, p.eps_nm) + xa_next;          // bound - M_m - |m|^2/2
            if (m >= p.nred || stg < diag_end) X = INFINITY;                                   // takes nothing
#ifdef FM_ABLATE_TRI_NOCOL          // ablation build only (results are wrong): the sweep when the column direction never fires
            X = INFINITY;
#endif
            char* slot = xbase + (stg % 3) * kFTriXBytes;
            ((float*)slot)[tid] = X;
            ((float*)(slot + 512))[tid] = -xa_next;
            float m4 = fminf(X, __shfl_xor(X, 1));
            m4 = fminf(m4, __shfl_xor(m4, 2));
            if ((tid & 3) == 0) ((float*)(slot + 1024))[tid >> 2] = m4;
        }
    };
    if constexpr (TRI) {
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int n = cb + 16 * j + c16;
            hc[j] = (n < p.ncols_pad) ? 0.5f * p.col_norm[n] : 0.f;
        }

        // stages st0 and st0 + 1 at once (their loads are waited for here: once per piece), st0 + 2 requested
        x_request(st0);     x_store(st0);
        x_request(st0 + 1); x_store(st0 + 1);
        x_request(st0 + 2);
    }

    float ea[NC][kFP];
    int   ei[NC][kFP];
    // Shared bounds, exact once every lane is done.  bound1[n] = max over all lanes of their
    // best acc.  K = 2: a lane publishes a row to bound1 at most once, by a RETURNING fetch-max,
    // and min(value before, lane's best) is the acc of the worse of two different rows, i.e. a
    // valid bound on the 2nd best; so is a lane's own 2nd best.  bound2[n] = max of those, which
    // ends as exactly the 2nd best acc overall (the fetch-maxes are serialised in L2: whichever
    // of the two best rows is published later sees the other or something better).
    // All atomics are issued at the stage hand-over, right behind its wait + barrier, and the
    // returned values are consumed at the next hand-over: nothing issued in between is waited
    // for, and the wait of the hand-over finds only a stage-old LDS-DMA, loads and atomics.
    // The lane's threshold thr[] only ever rises.
    // r06: a publish happens only for what CHANGED since the last hand-over (flag bits set by the visits;
    // r05 re-published the K-th best of every owner lane at every stage: in the 2-NN shape of config 5 --
    // 2560 workgroups x 122 stages -- that, the per-stage refresh and the per-tile rescaling of the
    // accumulator inits were 0.4 ms of 2.4: profiles/r06a_*), and the shared bounds are re-read at every
    // stage of a sweep's first 8, then at the stages whose number & bound_mask == 0.
    // flags: bits 0-3 best changed (per block j; K = 2), 4-7 K-th best changed, 20-23 (K = 2) the value a
    // publish to bound1 returned is pending in pend[j].
    float thr[NC];
    int   gnext[NC];             // bound of rank K, loaded ahead
    int   pend[NC];              // K = 2: value returned by this lane's last bound1 publish
    unsigned flags = 0u;
    int* const bound1 = p.bound;
    int* const boundk = p.bound + (KTOP == 2 ? p.ncols_alloc : 0);
    const int kNone = fmap(kFEmpty);
#pragma unroll
    for (int j = 0; j < NC; ++j) {
#pragma unroll
        for (int i = 0; i < kFP; ++i) { ea[j][i] = kFEmpty; ei[j][i] = -1; }
        thr[j] = kFEmpty;
        pend[j] = kNone;
        gnext[j] = __hip_atomic_load(boundk + cb + 16 * j + c16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    auto publish_all = [&]() __attribute__((always_inline)) {
        if (__builtin_amdgcn_ballot_w64((flags & 0xf000ffu) != 0u) != 0ull) {
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                if constexpr (KTOP == 2) {
                    // turn the value returned by the last bound1 publish into a bound2 publish
                    if (flags & (0x100000u << j)) {
                        const float v2 = fminf(funmap(pend[j]), ea[j][0]);
                        __hip_atomic_fetch_max(boundk + cb + 16 * j + c16, fmap(v2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                // the lane's K-th best if that is what its threshold rests on
                if ((flags & (0x10u << j)) && ea[j][KTOP - 1] - marg[j] >= thr[j])
                    __hip_atomic_fetch_max(boundk + cb + 16 * j + c16, fmap(ea[j][KTOP - 1]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if constexpr (KTOP == 2) {
                    // a new best row (only a CHANGED best: one published twice would meet its own first publish in bound1
                    // and pass for two rows)
                    if (flags & (1u << j)) {
                        pend[j] = __hip_atomic_fetch_max(bound1 + cb + 16 * j + c16, fmap(ea[j][0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        flags |= 0x100000u << j;
                    } else {
                        flags &= ~(0x100000u << j);
                    }
                }
            }
            flags &= ~0xffu;
        }
    };

    // per-lane LDS addresses of the A fragments of the CURRENT half stage (bit 14) of the CURRENT buffer (bit 15):
    // row c16, chunk (4s+g) ^ c16; the tiles of a half are the immediates 0, 4096, 8192, 12288
    int aoff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) aoff[s] = c16 * kFRowBytes + 16 * ((4 * s + g) ^ c16);
    int xoff = kFAuxBase + 16 * g;      // aux words: + 64 per tile, bit 8 = half, bit 9 = buffer
    int cur = 0;                        // (uniform) buffer being consumed

    // ONE fragment set: fragment register s of the next tile is re-read from LDS right behind the four MFMAs that consume
    // it (12 MFMAs = ~190 cycles before its next use).  TWO accumulator sets: the MFMAs of tile t and the reduce of tile
    // t - 1 are one basic block (the compiler interleaves them), so a wave's next MFMAs never wait for a reduce chain.
    // A stage is two passes over a chain of four tiles (the halves differ by an XOR on the address registers, as the
    // buffers do): four copies of the visit.  (r06: this loop against r05's -- two fragment sets, reduce per 32-row unit
    // behind a scheduling barrier, eight unrolled tiles -- 1.82-1.89 against 2.2 ms without visits in the 2-NN shape of
    // config 5, profiles/r06a_*.)
    v8h fs[4];
    v4f cs;
    // (red_aux is in accumulator units already: launch_filter rescales it into a scratch array for banks of different scales)
    if (st0 < st1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (st0 + 1 < st1) f_issue_stage<NW>(p, st0 + 1, smem, 1, wave, lane);
        cs = *(const v4f*)(smem + xoff);
#pragma unroll
        for (int s = 0; s < 4; ++s) fs[s] = __builtin_bit_cast(v8h, *(const v4i*)(smem + aoff[s]));
        // bounds other workgroups have published already (loaded at the top) apply from the first tile on
#pragma unroll
        for (int j = 0; j < NC; ++j) thr[j] = fmaxf(thr[j], funmap(gnext[j]) - marg[j]);
    }

    v4f acc[2][NC];
    unsigned long long hit[NC];         // the lanes whose fast test fired for block j (of the tile reduced last)
    unsigned long long chit[NC];        // TRI: ... whose column-direction first-level test fired
#pragma unroll
    for (int j = 0; j < NC; ++j) acc[1][j] = v4f{-INFINITY, -INFINITY, -INFINITY, -INFINITY};       // "tile -1": nothing fires
    // reduce of the tile in set S (number tile_no): hit[]
    auto reduce_tile = [&](auto s_tag, int tile_no) __attribute__((always_inline)) -> bool {
        constexpr int S = decltype(s_tag)::value;
        if constexpr (SELF) {
            // 16-row tile d of the wave's own rows faces block j = d: lane (c16, g) holds the pair (row 4 g + reg, output
            // row c16), so the diagonal is reg = c16 & 3 of lane group c16 >> 2
            const unsigned d = (unsigned)(tile_no - (cb >> 4));
            if (d < (unsigned)NC) {
                const bool dl = g == (c16 >> 2);
#pragma unroll
                for (int j = 0; j < NC; ++j) {
                    if ((unsigned)j == d) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            acc[S][j][r] = (dl && (c16 & 3) == r) ? -INFINITY : acc[S][j][r];
                    }
                }
            }
        }
        float x4 = INFINITY;
        if constexpr (TRI) {
            // the loosest X of the lane's four rows of this tile (the priming call in front of a piece's first tile -- set S holds
            // -inf, the slot of that tile number is not this piece's -- tests against +inf: nothing fires)
            if (tile_no >= st0 * (kFStageRows / 16))          // (uniform)
                x4 = *(const float*)(xbase + ((tile_no >> 3) % 3) * kFTriXBytes + 1024 + 4 * (4 * (tile_no & 7) + g));
        }
#pragma unroll
        for (int j = 0; j < NC; ++j) {      // v_maximum3_f32 (no NaN-quieting pre-pass as fmaxf needs); NaN cannot occur here
            const float tm = fmax3(fmax3(acc[S][j][0], acc[S][j][1], acc[S][j][2]), acc[S][j][3], acc[S][j][3]);
            hit[j] = __builtin_amdgcn_ballot_w64(tm >= thr[j]);
            if constexpr (TRI) chit[j] = __builtin_amdgcn_ballot_w64(tm - hc[j] >= x4);       // column direction, first level
        }
        unsigned long long any = hit[0] | hit[1] | hit[2] | hit[3];
        if constexpr (TRI) any |= chit[0] | chit[1] | chit[2] | chit[3];
#ifdef FM_ABLATE_F32_NOEXACT
        // ablation build only (scripts/README.md): what the filter costs when NO tile takes the exact path.  Results are wrong.
        return any != 0ull && p.nstages < 0;
#else
        return any != 0ull;
#endif
    };
    // tile K of half h of stage st: its MFMAs into set K & 1 and, in the same block, the reduce of the tile before it
    auto tile = [&](auto k_tag, int st, int h) __attribute__((always_inline)) -> bool {
        constexpr int K = decltype(k_tag)::value;
        constexpr int KN = (K + 1) & 3;
        constexpr int S = K & 1;
        if constexpr (K == 3) {
            if (h != 0 && st + 1 < st1) {
                // stage boundary: this tile's fragments are in registers; behind "my share of stage st + 1 has landed" + one
                // barrier every wave may read stage st + 1, and buffer `cur` may be refilled with stage st + 2
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#ifndef FM_ABLATE_K8_NOBARRIER      // ablation builds (scripts/gpu_k8_ablate.sh; their results are wrong): the waves' lock step
                __syncthreads();
#endif
                const bool fresh = (st - st0 < 8) || (((st - st0) & p.bound_mask) == 0);
                publish_all();
                if (fresh) {
#pragma unroll
                    for (int j = 0; j < NC; ++j) thr[j] = fmaxf(thr[j], funmap(gnext[j]) - marg[j]);
                }
                if (st + 2 < st1) f_issue_stage<NW>(p, st + 2, smem, cur, wave, lane);
                cur ^= 1;
                if (fresh) {
#pragma unroll
                    for (int j = 0; j < NC; ++j)
                        gnext[j] = __hip_atomic_load(boundk + cb + 16 * j + c16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if constexpr (TRI) {
                    // stage st + 2's words (requested one hand-over ago, landed behind the wait above) into its slot -- the
                    // slot of stage st - 1, which every wave left before the barrier above; then stage st + 3's are requested
                    x_store(st + 2);
                    x_request(st + 3);
                }
            }
            // the next tile is the first of the other half -- of the other buffer behind the second half
            const int tog = h != 0 ? (kFStageRowBytes | (kFStageRowBytes >> 1)) : (kFStageRowBytes >> 1);
#pragma unroll
            for (int s = 0; s < 4; ++s) aoff[s] ^= tog;
            xoff ^= h != 0 ? (512 | 256) : 256;
        }
#pragma unroll
        for (int j = 0; j < NC; ++j) acc[S][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fs[0], bh[j][0], cs, 0, 0, 0);
        cs = *(const v4f*)(smem + xoff + KN * 64);
        fs[0] = __builtin_bit_cast(v8h, *(const v4i*)(smem + aoff[0] + KN * 16 * kFRowBytes));
#pragma unroll
        for (int s = 1; s < 4; ++s) {
#pragma unroll
            for (int j = 0; j < NC; ++j) acc[S][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fs[s], bh[j][s], acc[S][j], 0, 0, 0);
            fs[s] = __builtin_bit_cast(v8h, *(const v4i*)(smem + aoff[s] + KN * 16 * kFRowBytes));
        }
        return reduce_tile(std::integral_constant<int, S ^ 1>{}, st * (kFStageRows / 16) + 4 * h + K - 1);
    };
    // the exact path of one tile (number tile_no, accumulators in set S) whose fast test fired for some lane
    auto visit = [&](auto s_tag, int tile_no) __attribute__((always_inline)) {
        constexpr int S = decltype(s_tag)::value;
        const int row0 = tile_no * 16 + 4 * g;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            if (hit[j] == 0ull) continue;
            // A tile reaches this point because SOME lane of the wave holds a row at or above its threshold -- typically
            // one lane and one of its 4 rows.  The P-step insertion costs ~6 VALU per step, so it runs only for the rows
            // that some lane actually wants (wave-uniform skip; a row below the lane's threshold leaves its list unchanged).
            // The wave-wide tests first, back to back, as scalar masks: the loop below then branches on SGPRs.
            unsigned long long wm[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) wm[r] = __builtin_amdgcn_ballot_w64(acc[S][j][r] >= thr[j]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {                   // ascending row order
                if (wm[r] == 0ull) continue;
                const float av = acc[S][j][r];
                float a = av >= thr[j] ? av : -INFINITY;
                int id = row0 + r;
#pragma unroll
                for (int i = 0; i < kFP; ++i) {
                    const bool b = a > ea[j][i];
                    const float ta = b ? ea[j][i] : a;
                    const int ti = b ? ei[j][i] : id;
                    ea[j][i] = b ? a : ea[j][i];
                    ei[j][i] = b ? id : ei[j][i];
                    if (KTOP == 2 && i == 0) flags |= b ? (1u << j) : 0u;
                    if (i == KTOP - 1) flags |= b ? (0x10u << j) : 0u;
                    a = ta; id = ti;
                }
            }
            thr[j] = fmaxf(thr[j], ea[j][KTOP - 1] - marg[j]);
        }
        if constexpr (TRI) {
            // column direction: the streamed rows of this tile gain candidates from the lane's output rows
            // (r06, measured and not kept: ONE first-level test for the four blocks -- more false visits than it saves --, the
            // append's returning atomic consumed a stage later instead of at once (1.820 against 1.816 ms), and a fire raising
            // the row's X in LDS for the rest of the workgroup, as the integer sweep does (1.90 against 1.81 ms))
            if ((chit[0] | chit[1] | chit[2] | chit[3]) != 0ull) {
                const char* slot = xbase + ((tile_no >> 3) % 3) * kFTriXBytes;
                const v4f xr = *(const v4f*)(slot + 4 * (16 * (tile_no & 7) + 4 * g));
                const v4f hm = *(const v4f*)(slot + 512 + 4 * (16 * (tile_no & 7) + 4 * g));
#pragma unroll
                for (int j = 0; j < NC; ++j) {
                    if (chit[j] == 0ull) continue;
                    const int n = cb + 16 * j + c16;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = acc[S][j][r] - hc[j];
                        const bool fire = v >= xr[r] && n < p.ncols;
                        if (__builtin_amdgcn_ballot_w64(fire) == 0ull) continue;
                        if (fire) {
                            const int m = row0 + r;
                            const float ap = v + hm[r];                    // the pair in row m's frame
                            __hip_atomic_fetch_max(p.bound + m, fmap(ap), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef FM_ABLATE_TRI_NOAPPEND       // ablation build only (results are wrong): the column direction without its list append
                            continue;
#endif
                            const int pos = __hip_atomic_fetch_add(p.cnt + m, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (kFTriRedo - 1);
                            if (pos < kFTriCap) p.cand[(size_t)m * kFTriCap + pos] = ((unsigned long long)__float_as_uint(ap) << 32) | (unsigned)n;
                            else __hip_atomic_fetch_or(p.cnt + m, kFTriRedo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
            }
        }
    };

    int t0 = st0 * (kFStageRows / 16);
#pragma unroll 1
    for (int st = st0; st < st1; ++st) {
#pragma unroll 1
        for (int h = 0; h < 2; ++h, t0 += 4) {
            if (tile(std::integral_constant<int, 0>{}, st, h)) visit(std::integral_constant<int, 1>{}, t0 - 1);
            if (tile(std::integral_constant<int, 1>{}, st, h)) visit(std::integral_constant<int, 0>{}, t0);
            if (tile(std::integral_constant<int, 2>{}, st, h)) visit(std::integral_constant<int, 1>{}, t0 + 1);
            if (tile(std::integral_constant<int, 3>{}, st, h)) visit(std::integral_constant<int, 0>{}, t0 + 2);
        }
    }
    // the last tile of the sweep
    if (st0 < st1 && reduce_tile(std::integral_constant<int, 1>{}, t0 - 1)) visit(std::integral_constant<int, 1>{}, t0 - 1);

    // last publishes: what changed since the last hand-over (K = 2: the value a publish returns is needed, so twice);
    // then emit every entry; rescore_kernel filters them against the final bound
    publish_all();
    if constexpr (KTOP == 2) publish_all();
#ifdef FM_ABLATE_K8_NORESCORE       // ... the sweep without its fused epilogue
    if (p.fused && p.nstages > 0) return;
#endif
    if constexpr (TRI) {
        // the lane's entries that are inside the margin of their row's bound as it stands now (it only rises: a superset of
        // what the final bound admits) join the row's list; a fourth entry inside it: rows the lane dropped may be too
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int n = cb + 16 * j + c16;
            if (n >= p.ncols) continue;
            const float t = funmap(__hip_atomic_load(p.bound + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) - marg[j];
#pragma unroll
            for (int i = 0; i < kFP; ++i) {
                if (ei[j][i] >= 0 && ei[j][i] < p.nred && ea[j][i] >= t) {
                    const int pos = __hip_atomic_fetch_add(p.cnt + n, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (kFTriRedo - 1);
                    if (pos < kFTriCap) p.cand[(size_t)n * kFTriCap + pos] = ((unsigned long long)__float_as_uint(ea[j][i]) << 32) | (unsigned)ei[j][i];
                    if (pos >= kFTriCap || i == kFP - 1)
                        __hip_atomic_fetch_or(p.cnt + n, kFTriRedo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        return;
    }
    if (p.fused) {
        // ---- exact rescoring in place (the rule of rescore_kernel below, on registers) --------------
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int n = cb + 16 * j + c16;
            const bool ok = n < p.ncols;
            // K-th best accumulator over the four lane groups (disjoint rows of the same output rows)
            float b1 = ea[j][0], b2 = KTOP == 2 ? ea[j][1] : kFEmpty;
#pragma unroll
            for (int mask = 16; mask <= 32; mask <<= 1) {
                const float o1 = __shfl_xor(b1, mask), o2 = __shfl_xor(b2, mask);
                const float lo = fminf(b1, o1);
                b1 = fmaxf(b1, o1);
                b2 = fmaxf(fmaxf(b2, o2), lo);
            }
            const float fthr = (KTOP == 2 ? b2 : b1) - marg[j];
            const float4* cp = (const float4*)(p.col_rowsf + (size_t)(ok ? n : 0) * kDim);
            unsigned long long k0 = ~0ull, k1 = ~0ull;
            int redo = 0;
#pragma unroll
            for (int i = 0; i < kFP; ++i) {
                const bool valid = ok && ei[j][i] >= 0 && ei[j][i] < p.nred && ea[j][i] >= fthr;
                if (__builtin_amdgcn_ballot_w64(valid) == 0ull) continue;
                if (valid) {
                    if (i == kFP - 1) redo = 1;      // the lane's last entry inside the margin: rows it dropped may be too
                    const float4* rp = (const float4*)(p.red_rowsf + (size_t)ei[j][i] * kDim);
                    float sum = 0.f;
#pragma unroll 8
                    for (int k4 = 0; k4 < kDim / 4; ++k4) {
#ifdef FM_ABL_RS_NOA                // ... the epilogue without the loads of its output rows / of its candidates
                        const float4 a = make_float4(sum, 1.f, 2.f, 3.f);
#else
                        const float4 a = cp[k4];
#endif
#ifdef FM_ABL_RS_NOB
                        const float4 b = make_float4(4.f, sum, 5.f, 6.f);
#else
                        const float4 b = rp[k4];
#endif
                        float v;
                        v = a.x - b.x; sum = __builtin_fmaf(v, v, sum);
                        v = a.y - b.y; sum = __builtin_fmaf(v, v, sum);
                        v = a.z - b.z; sum = __builtin_fmaf(v, v, sum);
                        v = a.w - b.w; sum = __builtin_fmaf(v, v, sum);
                    }
                    const unsigned long long key = ((unsigned long long)__float_as_uint(sqrtf(sum)) << 32) | (unsigned)ei[j][i];
                    if (key < k0) { k1 = k0; k0 = key; }
                    else if (key < k1) { k1 = key; }
                }
            }
#pragma unroll
            for (int mask = 16; mask <= 32; mask <<= 1) {
                const unsigned long long o0 = __shfl_xor(k0, mask), o1 = __shfl_xor(k1, mask);
                const unsigned long long lo = k0 < o0 ? k0 : o0, hi = k0 < o0 ? o0 : k0;
                const unsigned long long m1 = k1 < o1 ? k1 : o1;
                k0 = lo;
                k1 = hi < m1 ? hi : m1;
                redo |= __shfl_xor(redo, mask);
            }
            if (ok && g == 0) {
                p.partial[(size_t)n * KTOP] = k0;
                if constexpr (KTOP == 2) p.partial[(size_t)n * KTOP + 1] = k1;
                if (redo) {
                    const int pos = atomicAdd(p.flag + 1, 1);
                    atomicAdd(p.flag + 3, 1);
                    if (pos < kFMaxRescan) p.flag[4 + pos] = n;
                    else atomicOr(p.flag, 1);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int n = cb + 16 * j + c16;
        unsigned long long* out = p.slots + (((size_t)split * p.ncols_alloc + n) * 4 + g) * kFP;
#pragma unroll
        for (int i = 0; i < kFP; ++i) {
            const bool v = ei[j][i] >= 0 && ei[j][i] < p.nred && n < p.ncols;
            out[i] = v ? (((unsigned long long)__float_as_uint(ea[j][i]) << 32) | (unsigned)ei[j][i]) : ~0ull;
        }
    }
}

// ---- exact rescoring ------------------------------------------------------------------------
struct RParams {
    const unsigned long long* slots;
    int          nsplit;
    int          ncols_alloc;     // of the slots / bound arrays
    const int*   bound;           // the K-th best bound array (bound1 for K = 1, bound2 for K = 2)
    const float* col_rowsf;
    const float* col_norm;
    const float* red_rowsf;
    float        eps_c, eps_nm;   // as in FParams
    int          ncols;           // real output rows
    unsigned long long* partial;  // split 0 of the caller's layout: [n][KTOP]
    int*         flag;            // [0] raised when more than kFMaxRescan output rows need a full scan,
                                  // [1] number of such rows, [3] total (diagnostic), [4 ..] their indices
    int          self;            // the banks are one bank and row n is not a candidate for output row n
};

template <int KTOP, int LPC>       // LPC lanes per output row (16 or 64)
__global__ __launch_bounds__(256)
void rescore_kernel(RParams p)
{
    const int sub = threadIdx.x & (LPC - 1);
    const int n = (int)((blockIdx.x * 256 + threadIdx.x) / LPC);
    const bool live = n < p.ncols;
    const int nn = live ? n : 0;
    const float thr = funmap(p.bound[nn]) - fmaf(p.eps_c, p.col_norm[nn], p.eps_nm);
    const float4* cp = (const float4*)(p.col_rowsf + (size_t)nn * kDim);
    unsigned long long k0 = ~0ull, k1 = ~0ull;
    bool incomplete = false;
    const int nslots = p.nsplit * 4 * kFP;
    for (int s0 = 0; s0 < nslots; s0 += LPC) {
        const int s = s0 + sub;
        unsigned long long slot = ~0ull;
        if (live && s < nslots) slot = p.slots[((size_t)(s >> 4) * p.ncols_alloc + nn) * (4 * kFP) + (s & 15)];
        const bool valid = slot != ~0ull && __uint_as_float((unsigned)(slot >> 32)) >= thr;
        const unsigned long long vm = __builtin_amdgcn_ballot_w64(valid);
        // a lane's last (smallest) entry inside the margin: rows it dropped may be inside too
        incomplete |= valid && (s & (kFP - 1)) == kFP - 1;
        if (vm == 0ull) continue;
        if (valid) {
            const unsigned idx = (unsigned)slot;
            const float4* rp = (const float4*)(p.red_rowsf + (size_t)idx * kDim);
            float sum = 0.f;
#pragma unroll 8
            for (int k4 = 0; k4 < kDim / 4; ++k4) {
                const float4 a = cp[k4];
                const float4 b = rp[k4];
                float v;
                v = a.x - b.x; sum = __builtin_fmaf(v, v, sum);
                v = a.y - b.y; sum = __builtin_fmaf(v, v, sum);
                v = a.z - b.z; sum = __builtin_fmaf(v, v, sum);
                v = a.w - b.w; sum = __builtin_fmaf(v, v, sum);
            }
            const unsigned long long key = ((unsigned long long)__float_as_uint(sqrtf(sum)) << 32) | idx;
            if (key < k0) { k1 = k0; k0 = key; }
            else if (key < k1) { k1 = key; }
        }
    }
    int redo = incomplete ? 1 : 0;
#pragma unroll
    for (int mask = 1; mask < LPC; mask <<= 1) {
        const unsigned long long o0 = __shfl_xor(k0, mask), o1 = __shfl_xor(k1, mask);
        const unsigned long long lo = k0 < o0 ? k0 : o0, hi = k0 < o0 ? o0 : k0;
        const unsigned long long m1 = k1 < o1 ? k1 : o1;
        k0 = lo;
        k1 = hi < m1 ? hi : m1;
        redo |= __shfl_xor(redo, mask);
    }
    if (live && sub == 0) {
        p.partial[(size_t)n * KTOP] = k0;
        if constexpr (KTOP == 2) p.partial[(size_t)n * KTOP + 1] = k1;
        if (redo) {
            // this output row gets a full exact scan (rescan_kernel); too many of them: K5 instead
            const int pos = atomicAdd(p.flag + 1, 1);
            atomicAdd(p.flag + 3, 1);
            if (pos < kFMaxRescan) p.flag[4 + pos] = n;
            else atomicOr(p.flag, 1);
        }
    }
}

// Full exact scan for the few output rows whose candidate lists could be incomplete.  A row's
// scan is split over kFRescanSplit workgroups (thread t of part s takes rows s * chunk + t,
// + 256, ...), each part leaves its top-2 in the scratch behind the flag words, and the part that
// arrives last (a ticket per row) merges them: a 10k-row scan takes ~8 us instead of ~120 us.
constexpr int kFRescanSplit = 16;

template <int KTOP>
__global__ __launch_bounds__(256)
void rescan_kernel(RParams p, int nred)
{
    __shared__ unsigned long long sk[256 * 2];
    __shared__ int last;
    const int cnt = min(p.flag[1], kFMaxRescan);
    const int slot = blockIdx.x / kFRescanSplit, part = blockIdx.x % kFRescanSplit;
    if (slot >= cnt || p.flag[0] != 0) return;
    const int n = p.flag[4 + slot];
    int* tickets = p.flag + 4 + kFMaxRescan;
    unsigned long long* scratch = (unsigned long long*)(p.flag + 4 + 2 * kFMaxRescan) + (size_t)slot * kFRescanSplit * 2;
    const int chunk = (nred + kFRescanSplit - 1) / kFRescanSplit;
    const int m0 = part * chunk, m1 = min(nred, m0 + chunk);
    const float4* cp = (const float4*)(p.col_rowsf + (size_t)n * kDim);
    unsigned long long k0 = ~0ull, k1 = ~0ull;
    for (int m = m0 + threadIdx.x; m < m1; m += 256) {
        if (p.self && m == n) continue;
        const float4* rp = (const float4*)(p.red_rowsf + (size_t)m * kDim);
        float sum = 0.f;
#pragma unroll 8
        for (int k4 = 0; k4 < kDim / 4; ++k4) {
            const float4 a = cp[k4];
            const float4 b = rp[k4];
            float v;
            v = a.x - b.x; sum = __builtin_fmaf(v, v, sum);
            v = a.y - b.y; sum = __builtin_fmaf(v, v, sum);
            v = a.z - b.z; sum = __builtin_fmaf(v, v, sum);
            v = a.w - b.w; sum = __builtin_fmaf(v, v, sum);
        }
        const unsigned long long key = ((unsigned long long)__float_as_uint(sqrtf(sum)) << 32) | (unsigned)m;
        if (key < k0) { k1 = k0; k0 = key; }
        else if (key < k1) { k1 = key; }
    }
    sk[2 * threadIdx.x] = k0;
    sk[2 * threadIdx.x + 1] = k1;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) {
            const unsigned long long a0 = sk[2 * threadIdx.x], a1 = sk[2 * threadIdx.x + 1];
            const unsigned long long o0 = sk[2 * (threadIdx.x + w)], o1 = sk[2 * (threadIdx.x + w) + 1];
            const unsigned long long lo = a0 < o0 ? a0 : o0, hi = a0 < o0 ? o0 : a0;
            const unsigned long long m1k = a1 < o1 ? a1 : o1;
            sk[2 * threadIdx.x] = lo;
            sk[2 * threadIdx.x + 1] = hi < m1k ? hi : m1k;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        __hip_atomic_store(scratch + 2 * part, sk[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(scratch + 2 * part + 1, sk[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        last = (atomicAdd(&tickets[slot], 1) == kFRescanSplit - 1) ? 1 : 0;
        if (last) {
            __threadfence();
            unsigned long long b0 = ~0ull, b1 = ~0ull;
            for (int q = 0; q < 2 * kFRescanSplit; ++q) {
                const unsigned long long v = __hip_atomic_load(scratch + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v < b0) { b1 = b0; b0 = v; }
                else if (v < b1) { b1 = v; }
            }
            p.partial[(size_t)n * KTOP] = b0;
            if constexpr (KTOP == 2) p.partial[(size_t)n * KTOP + 1] = b1;
            tickets[slot] = 0;                              // ready for the next call
        }
    }
}

// TRI: a row's candidate list (both directions of the triangular sweep) against its final bound, exactly.  16 lanes per row.
struct TParams {
    const unsigned long long* cand;
    const int*   cnt;
    const int*   bound;
    const float* rowsf;
    const float* norm;
    float        eps_c, eps_nm;
    int          n;
    unsigned long long* partial;    // [n]
    int*         flag;              // as RParams::flag
};

__global__ __launch_bounds__(256)
void tri_rescore_kernel(TParams p)
{
    const int sub = threadIdx.x & 15;
    const int n = (int)((blockIdx.x * 256 + threadIdx.x) >> 4);
    const bool live = n < p.n;
    const int nn = live ? n : 0;
    const int c = p.cnt[nn];
    const int total = c & (kFTriRedo - 1);
    const int nent = live ? min(total, kFTriCap) : 0;
    const float thr = funmap(p.bound[nn]) - fmaf(p.eps_c, p.norm[nn], p.eps_nm);
    const float4* cp = (const float4*)(p.rowsf + (size_t)nn * kDim);
    unsigned long long k0 = ~0ull;
    for (int e0 = 0; e0 < kFTriCap; e0 += 16) {
        const int e = e0 + sub;
        unsigned long long slot = ~0ull;
        if (e < nent) slot = p.cand[(size_t)nn * kFTriCap + e];
        const unsigned idx = (unsigned)slot;
        const bool valid = e < nent && __uint_as_float((unsigned)(slot >> 32)) >= thr && idx != (unsigned)nn;
        if (__builtin_amdgcn_ballot_w64(valid) == 0ull) continue;
        if (valid) {
            const float4* rp = (const float4*)(p.rowsf + (size_t)idx * kDim);
            float sum = 0.f;
#pragma unroll 8
            for (int k4 = 0; k4 < kDim / 4; ++k4) {
                const float4 a = cp[k4];
                const float4 b = rp[k4];
                float v;
                v = a.x - b.x; sum = __builtin_fmaf(v, v, sum);
                v = a.y - b.y; sum = __builtin_fmaf(v, v, sum);
                v = a.z - b.z; sum = __builtin_fmaf(v, v, sum);
                v = a.w - b.w; sum = __builtin_fmaf(v, v, sum);
            }
            const unsigned long long key = ((unsigned long long)__float_as_uint(sqrtf(sum)) << 32) | idx;
            k0 = key < k0 ? key : k0;
        }
    }
#pragma unroll
    for (int mask = 1; mask < 16; mask <<= 1) {
        const unsigned long long o = __shfl_xor(k0, mask);
        k0 = o < k0 ? o : k0;
    }
    if (live && sub == 0) {
        p.partial[n] = k0;
        if ((c & kFTriRedo) || total > kFTriCap) {          // the list may be incomplete: a full exact scan of this row
            const int pos = atomicAdd(p.flag + 1, 1);
            atomicAdd(p.flag + 3, 1);
            if (pos < kFMaxRescan) p.flag[4 + pos] = n;
            else atomicOr(p.flag, 1);
        }
    }
}

// Accumulator inits of a reduced bank in the accumulator units of a pair of banks of different scales (times 2^(kc - km),
// exact); padding rows keep their -3.4e38 so that no scale brings them above an empty threshold.
__global__ __launch_bounds__(256)
void aux_rescale_kernel(const float* __restrict__ aux, float* __restrict__ out, int64_t n, float mul)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const float a = aux[i]; out[i] = a <= kFEmpty ? a : a * mul; }
}

FilterPlan plan_filter(int64_t ncols_pad, int64_t nred_pad, const Tuning& tn)
{
    FilterPlan pl;
    pl.nw = (tn.f32_nw == 4 || tn.f32_nw == 8) ? tn.f32_nw : 4;
    pl.fused = tn.f32_fused;
    pl.lpc = tn.f32_lpc;
    pl.nc = 4;                                        // (NC = 2 at 4 waves/SIMD was tried: it spills)
    pl.bound_every = tn.f32_bound_every;
    const int cb = 16 * pl.nc * pl.nw;
    pl.nchunks = (int)((ncols_pad + cb - 1) / cb);
    if (pl.nchunks < 1) pl.nchunks = 1;
    pl.ncols_alloc = pl.nchunks * cb;
    const int64_t nstages = nred_pad / kFStageRows;
    // workgroups the chip holds at once: 2 per CU at 4 waves, 1 at 8
    const int64_t slots = 256 * (pl.nw == 8 ? 1 : 2);
    int64_t want = 5 * slots;                         // ~5 rounds of them
    // a workgroup should sweep >= 64 stages (8192 rows: prologue and cold thresholds amortised),
    // unless that leaves fewer workgroups than the chip holds; never fewer than 4 stages
    int64_t nsplit = (want + pl.nchunks - 1) / pl.nchunks;
    if (nsplit > nstages / 64) nsplit = nstages / 64;
    if (nsplit < 1) nsplit = 1;
    if (nsplit * pl.nchunks < slots) {
        nsplit = (slots + pl.nchunks - 1) / pl.nchunks;
        if (nsplit > nstages / 4) nsplit = nstages / 4;
        if (nsplit < 1) nsplit = 1;
    }
    if (tn.f32_nsplit > 0) nsplit = tn.f32_nsplit;
    if (nsplit > nstages) nsplit = nstages > 0 ? nstages : 1;
    int64_t per = (nstages + nsplit - 1) / nsplit;
    if (per < 1) per = 1;
    nsplit = (nstages + per - 1) / per;
    if (nsplit < 1) nsplit = 1;
    pl.nsplit = (int)nsplit;
    pl.stages_per_split = (int)per;
    pl.aux_elems = nred_pad;
    return pl;
}

size_t filter_flag_bytes() { return (size_t)(4 + 2 * kFMaxRescan) * 4 + (size_t)kFMaxRescan * kFRescanSplit * 2 * 8; }

int filter_empty_bound()
{
    union { float f; int i; } u;
    u.f = kFEmpty;
    return u.i ^ ((u.i >> 31) & 0x7fffffff);
}

bool filter_usable(const Bank& cols, const Bank& red)
{
    if (!cols.filt_ok || !red.filt_ok || !cols.rowsh || !red.rowsh) return false;
    const int d = cols.kscale - red.kscale;
    return d >= -40 && d <= 40;
}

hipError_t launch_filter(const Bank& cols, const Bank& red, int ktop, const FilterPlan& pl,
                         unsigned long long* slots, int* bound, int* flag,
                         unsigned long long* partial, hipStream_t stream, bool self, float* aux_scratch)
{
    if (self && (ktop != 1 || &cols != &red)) return hipErrorInvalidValue;
    if (cols.kscale != red.kscale && !aux_scratch) return hipErrorInvalidValue;
    const float eps = 1.1f / 1024.0f;
    const int dk = cols.kscale - red.kscale;          // acc units are 2^(kc + km)
    FParams p;
    p.col_rows = (const char*)cols.rowsh;
    p.col_norm = cols.normf;
    p.ncols = (int)cols.n;
    p.ncols_pad = (int)cols.n_pad;
    p.red_rows = (const char*)red.rowsh;
    p.red_aux = red.auxf;
    p.nred = (int)red.n;
    p.nstages = (int)(red.n_pad / kFStageRows);
    p.nsplit = pl.nsplit;
    p.nchunks = pl.nchunks;
    p.stages_per_split = pl.stages_per_split;
    p.ncols_alloc = pl.ncols_alloc;
    p.eps_c = ldexpf(eps, -dk);
    p.eps_nm = ldexpf(eps * red.nm_max, dk);
    p.aux_mul = ldexpf(1.0f, dk);
    p.slots = slots;
    p.bound = bound;
    p.flag = flag;
    p.fused = (pl.nsplit == 1 && cols.n > 0) ? 1 : 0;
    if (pl.fused >= 0) p.fused = (pl.fused != 0 && pl.nsplit == 1 && cols.n > 0) ? 1 : 0;
    p.col_rowsf = cols.rowsf;
    p.red_rowsf = red.rowsf;
    p.partial = partial;
    p.bound_mask = 0;
    for (int b = 2; b <= 64; b <<= 1) if (pl.bound_every == b) p.bound_mask = b - 1;
    if (dk != 0) {
        hipLaunchKernelGGL(aux_rescale_kernel, dim3((unsigned)((red.n_pad + 255) / 256)), dim3(256), 0, stream,
                           (const float*)red.auxf, aux_scratch, (int64_t)red.n_pad, p.aux_mul);
        p.red_aux = aux_scratch;
    }
    const int grid = pl.nchunks * pl.nsplit;
#define FM_LAUNCH_FILTER(NC_, NW_)                                                                         \
    do {                                                                                                   \
        if (self)           hipLaunchKernelGGL((filter_kernel<NC_, 1, NW_, true>), dim3(grid), dim3(64 * NW_), 0, stream, p); \
        else if (ktop == 1) hipLaunchKernelGGL((filter_kernel<NC_, 1, NW_>), dim3(grid), dim3(64 * NW_), 0, stream, p); \
        else                hipLaunchKernelGGL((filter_kernel<NC_, 2, NW_>), dim3(grid), dim3(64 * NW_), 0, stream, p); \
    } while (0)
    if (pl.nw == 8) FM_LAUNCH_FILTER(4, 8); else FM_LAUNCH_FILTER(4, 4);
#undef FM_LAUNCH_FILTER
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;

    RParams r;
    r.slots = slots;
    r.nsplit = pl.nsplit;
    r.ncols_alloc = pl.ncols_alloc;
    r.bound = (ktop == 2) ? bound + pl.ncols_alloc : bound;
    r.col_rowsf = cols.rowsf;
    r.col_norm = cols.normf;
    r.red_rowsf = red.rowsf;
    r.eps_c = p.eps_c;
    r.eps_nm = p.eps_nm;
    r.ncols = (int)cols.n;
    r.partial = partial;
    r.flag = flag;
    r.self = self ? 1 : 0;
    if (cols.n > 0 && p.fused) {
        // the filter rescored its own entries; only the flagged rows are left
        if (ktop == 1) hipLaunchKernelGGL((rescan_kernel<1>), dim3(kFMaxRescan * kFRescanSplit), dim3(256), 0, stream, r, (int)red.n);
        else           hipLaunchKernelGGL((rescan_kernel<2>), dim3(kFMaxRescan * kFRescanSplit), dim3(256), 0, stream, r, (int)red.n);
    } else if (cols.n > 0) {
        // lanes per output row: few slots and many rows -> one lane each (every lane of a wave
        // then runs a chain), else 16 or 64 lanes share a row's slots
        const int nslots = pl.nsplit * 4 * kFP;
        int lpc = nslots <= 32 ? (cols.n >= 65536 ? 1 : 16) : 64;
        if (pl.lpc == 1 || pl.lpc == 16 || pl.lpc == 64) lpc = pl.lpc;
        const int rgrid = (int)((cols.n * lpc + 255) / 256);
#define FM_LAUNCH_RESCORE(K_)                                                                                  \
        do {                                                                                                   \
            if (lpc == 1)       hipLaunchKernelGGL((rescore_kernel<K_, 1>), dim3(rgrid), dim3(256), 0, stream, r);  \
            else if (lpc == 16) hipLaunchKernelGGL((rescore_kernel<K_, 16>), dim3(rgrid), dim3(256), 0, stream, r); \
            else                hipLaunchKernelGGL((rescore_kernel<K_, 64>), dim3(rgrid), dim3(256), 0, stream, r); \
            hipLaunchKernelGGL((rescan_kernel<K_>), dim3(kFMaxRescan * kFRescanSplit), dim3(256), 0, stream, r, (int)red.n);   \
        } while (0)
        if (ktop == 1) FM_LAUNCH_RESCORE(1); else FM_LAUNCH_RESCORE(2);
#undef FM_LAUNCH_RESCORE
    }
    return hipGetLastError();
}

size_t filter_tri_bytes(int ncols_alloc) { return (size_t)ncols_alloc * kFTriCap * 8 + (size_t)ncols_alloc * 4 + 256; }

// fm_self_dist of a float32-route bank by the triangular sweep: partial[n] = key of row n's nearest OTHER row (K5's key).
// ws: filter_tri_bytes(plan.ncols_alloc) bytes (lists, then counters); bound: plan.ncols_alloc words preset to
// filter_empty_bound(); flag as launch_filter's.
hipError_t launch_filter_tri(const Bank& bank, const TriPlan& plan, int bound_every, void* ws, int* bound, int* flag,
                             unsigned long long* partial, hipStream_t stream)
{
    if (plan.npieces < 1 || plan.stages < 4 || !bank.rowsh || !bank.filt_ok) return hipErrorInvalidValue;
    const float eps = 1.1f / 1024.0f;
    FParams p{};
    p.col_rows = (const char*)bank.rowsh;  p.col_norm = bank.normf;  p.ncols = (int)bank.n;  p.ncols_pad = (int)bank.n_pad;
    p.red_rows = (const char*)bank.rowsh;  p.red_aux = bank.auxf;    p.nred = (int)bank.n;
    p.nstages = (int)(bank.n_pad / kFStageRows);
    p.nsplit = 1;  p.nchunks = plan.nchunks;  p.stages_per_split = p.nstages;  p.ncols_alloc = plan.ncols_alloc;
    p.eps_c = eps;  p.eps_nm = eps * bank.nm_max;  p.aux_mul = 1.0f;
    p.slots = nullptr;  p.bound = bound;  p.flag = flag;  p.fused = 0;
    p.col_rowsf = bank.rowsf;  p.red_rowsf = bank.rowsf;  p.partial = partial;
    p.bound_mask = 0;
    for (int b = 2; b <= 64; b <<= 1) if (bound_every == b) p.bound_mask = b - 1;
    p.tri_S = plan.stages;
    p.cand = (unsigned long long*)ws;
    p.cnt = (int*)((char*)ws + (size_t)plan.ncols_alloc * kFTriCap * 8);
    hipError_t e = hipMemsetAsync(p.cnt, 0, (size_t)plan.ncols_alloc * 4, stream);
    if (e != hipSuccess) return e;
    for (int phase = 0; phase < 2; ++phase) {
        const int first = phase == 0 ? 0 : plan.ndiag, count = phase == 0 ? plan.ndiag : plan.npieces - plan.ndiag;
        if (count <= 0) continue;
        p.tri_first = first;
        hipLaunchKernelGGL((filter_kernel<4, 1, 8, true, true>), dim3((unsigned)count), dim3(512), 0, stream, p);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    TParams t;
    t.cand = p.cand;  t.cnt = p.cnt;  t.bound = bound;  t.rowsf = bank.rowsf;  t.norm = bank.normf;
    t.eps_c = p.eps_c;  t.eps_nm = p.eps_nm;  t.n = (int)bank.n;  t.partial = partial;  t.flag = flag;
    hipLaunchKernelGGL(tri_rescore_kernel, dim3((unsigned)(((size_t)bank.n * 16 + 255) / 256)), dim3(256), 0, stream, t);
    RParams r{};
    r.slots = nullptr;  r.nsplit = 1;  r.ncols_alloc = plan.ncols_alloc;  r.bound = bound;
    r.col_rowsf = bank.rowsf;  r.col_norm = bank.normf;  r.red_rowsf = bank.rowsf;
    r.eps_c = p.eps_c;  r.eps_nm = p.eps_nm;  r.ncols = (int)bank.n;  r.partial = partial;  r.flag = flag;  r.self = 1;
    hipLaunchKernelGGL((rescan_kernel<1>), dim3(kFMaxRescan * kFRescanSplit), dim3(256), 0, stream, r, (int)bank.n);
    return hipGetLastError();
}

}  // namespace fm
