// The arithmetic of one expansion round on integer-valued banks (cross-checked 1-NN, SURVEY.md
// Appendix A.3) as ONE workgroup-wide device function shared by round_kernel (rounds.hip) and
// expand_kernel (expand.hip).  NT threads; contains barriers: every thread of the block must call it.
#pragma once
#include "tile_ops.h"

namespace fm {

// x1_round_wsplit: on return (after the caller's next __syncthreads) qbest[slot], slot in
// [0, nq), holds (distance key << 32 | local train index) of the cross-checked match of query slot
// `slot`, or ~0.  Distance key: with tie_guard the float32 bits of the distance itself (OpenCV compares
// the square roots, and above kSqrtTieMin two d2 can share one; tile_ops.h); without it -- the banks' norms
// rule such distances out, e.g. every SIFT pair -- the integer d2, which then orders the same way and
// spares the election a square root per train row (x1_key_distance decodes either).
// q_rows[slot] = row of the query bank; train rows are [t0, t0 + nt) of the train bank.
// qbest must be pre-filled with ~0 for slots [0, nq) (visible to all threads).  The bank pointers are
// global-memory pointers (gptr, tile_ops.h): callers convert theirs once.
// SR = query rows gathered per staging step (128 in round_kernel, 512 in expand_kernel so that a
// typical round needs ONE global round trip); smem must hold SR * 128 + SR / 32 * 256 bytes.
// NT = threads of the workgroup (256 in round_kernel, 512 in expand_kernel).
// tie_guard: the banks' row norms allow d2 >= kSqrtTieMin (uniform); tbest must hold kTbestWords words.
//
// Every wave owns ALL four 32-column
// blocks of a 128-column chunk (four independent MFMA chains per tile, so the dependent
// accumulate latency is hidden) but only every fourth 32-row tile of the query subset; the
// four waves' reverse-NN candidates meet in an LDS table tbest[128] through 64-bit
// atomicMin on (d2 << 32 | slot) -- min d2, then lowest slot, exactly the order
// cv::batchDistance keeps -- before the scatter-min into qbest.
// float32 distance of a qbest key's high word (see x1_round_wsplit)
__device__ __forceinline__ float x1_key_distance(unsigned hi, int tie_guard)
{
    return tie_guard ? __uint_as_float(hi) : sqrtf((float)hi);
}

// The first 128-column chunk of a cell's train rows can be staged in LDS ahead of time by a caller that knows
// the cell long before it knows the query subset (K7: while it sorts the radius subset): 16 KiB, the rows in the
// swizzled layout of the query stages (the B fragments are read with the A fragments' pattern: row = lane & 31,
// 16-byte chunk 2 c + h).  Rows past the cell repeat its last row; their columns are never looked at.
constexpr int kCellStageBytes = 128 * kDim;

template <int NT>
__device__ __forceinline__ void x1_stage_cell(char* lds, gptr<const int8_t> t_rows8, int64_t t0, int nt)
{
    constexpr int kPieces = 16 / (NT / 64);           // 1-KiB pieces (8 rows) per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < kPieces; ++i) {
        const int g = wave * kPieces + i;
        const int row = g * 8 + (lane >> 3);
        const int r = row < nt ? row : nt - 1;
        gptr<const int8_t> src = t_rows8 + (size_t)(t0 + r) * kDim + 16 * ((lane & 7) ^ ((row >> 1) & 7));
        __builtin_amdgcn_global_load_lds((gptr<const void>)src, (__attribute__((address_space(3))) void*)(lds + g * 1024), 16, 0, 0);
    }
}

constexpr int kTbestWords = 128 + 4;       // 128 train rows of a chunk + the tie repair's two masks and its accumulator

// MERGE (K7's huge rounds, expand.hip): the query rows are ONE CHUNK (slots slot_base .. slot_base + nq of a radius
// subset that does not fit LDS) and the function stops at the reverse-NN step: per train row the running minimum
// (d2 << 32 | global slot) over the chunks seen so far is kept in tb_all[0 .. nt) (global memory; row cb0 + tid is
// read and written by thread tid only, chunk after chunk, so no other thread ever needs to see it before the caller's
// election) -- the cross-check is a per-train-row minimum over the query slots, and a minimum merges exactly.
// qbest is not touched; tie_guard must be 0 (the float32-root repair looks at slots below the elected one).
template <int SR, int NT = 256, bool MERGE = false>
__device__ __forceinline__ void x1_round_wsplit(gptr<const int8_t> q_rows8, gptr<const int32_t> q_norm,
                                                const int* q_rows, int nq,
                                                gptr<const int8_t> t_rows8, gptr<const int32_t> t_norm,
                                                int64_t t0, int nt, char* smem, unsigned long long* qbest,
                                                unsigned long long* tbest /* LDS [kTbestWords] */, int tie_guard,
                                                long long* pt = nullptr, long long* ts = nullptr,
                                                const char* cell0 = nullptr /* LDS: chunk 0 staged by x1_stage_cell */,
                                                gptr<unsigned long long> tb_all = nullptr, unsigned slot_base = 0)
{
#define X1_STAMP(k) do { if (pt && threadIdx.x == 0) { const long long _n = wall_clock64(); pt[k] += _n - *ts; *ts = _n; } } while (0)
    constexpr int NW = NT / 64;                   // waves
    constexpr int kGroups = NW / 4;               // 4 waves: every wave owns all four 32-column blocks;
    constexpr int NB = 4 / kGroups;               // 8 waves: two groups of waves own two blocks each
    static_assert(NW == 4 || NW == 8, "four or eight waves");
    constexpr int kPieces = (SR / 8) / NW;        // 1-KiB gather pieces (8 rows) per wave
    static_assert(kPieces >= 1 && (SR / 8) % NW == 0, "stage rows must spread evenly over the waves");
    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int h    = lane >> 5;
    const int sw = ((lane & 31) >> 1) & 7;
    int aoff[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) aoff[c] = (lane & 31) * kDim + 16 * ((2 * c + h) ^ sw);
    constexpr int kRowBytes = SR * kDim;
    const int xoff = kRowBytes + h * 64;
    const int nstages = (nq + SR - 1) / SR;
    const int blk0 = (wave % kGroups) * NB;       // this wave's first column block of the 128-column chunk
    const int tphase = wave / kGroups;            // ... and which tiles it takes (every fourth)

    for (int cb0 = 0; cb0 < nt; cb0 += 128) {
        if (tid < 128) tbest[tid] = ~0ull;
        v4i bf[NB][4];
        const bool staged = cell0 != nullptr && cb0 == 0;
        if (staged) {
            // staged in LDS while the caller was busy elsewhere; read below, behind the wait and the barrier of
            // the first query stage (the cell's DMA is older than that stage's: it has landed when they return)
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int c = 0; c < 4; ++c) bf[j][c] = v4i{0, 0, 0, 0};
        } else {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int n = cb0 + 32 * (blk0 + j) + (lane & 31);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (n < nt) bf[j][c] = *(gptr<const v4i>)(t_rows8 + (size_t)(t0 + n) * kDim + 32 * c + 16 * h);
                    else        bf[j][c] = v4i{0, 0, 0, 0};
                }
            }
        }
        TopTile top[NB];
        int tnorm[NB];                            // the merge's train-row norms: on their way from here
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            top[j].init();
            const int n = cb0 + 32 * (blk0 + j) + (lane & 31);
            tnorm[j] = t_norm[t0 + (n < nt ? n : nt - 1)];
        }

        for (int st = 0; st < nstages; ++st) {
            lds_barrier();                       // previous stage fully consumed
            X1_STAMP(8);
            // issue every global load of the step first (row norms, then the 16-byte row
            // pieces; unconditional, clamped past the subset), then do the LDS writes
            int nmv[(SR + NT - 1) / NT];
#pragma unroll
            for (int k = 0; k < (SR + NT - 1) / NT; ++k) {
                const int slot = st * SR + k * NT + tid;
                nmv[k] = q_norm[q_rows[slot < nq ? slot : nq - 1]];
            }
            // the rows themselves go global -> LDS by DMA (no registers, all SR/32 pieces of a wave in
            // flight together): a piece is 8 rows x 128 B, lane (row = lane >> 3, p = lane & 7) fetches
            // source chunk p ^ ((row >> 1) & 7), which lands at chunk position p.  Slots past the subset
            // fetch the last real row; their accumulator init (kPadCinit) keeps them below every real one.
#pragma unroll
            for (int i = 0; i < kPieces; ++i) {
                const int g = wave * kPieces + i;
                const int row = g * 8 + (lane >> 3);
                const int slot = st * SR + row;
                const int qi = q_rows[slot < nq ? slot : nq - 1];
                gptr<const int8_t> src = q_rows8 + (size_t)qi * kDim + 16 * ((lane & 7) ^ ((row >> 1) & 7));
                __builtin_amdgcn_global_load_lds((gptr<const void>)src,
                                                 (__attribute__((address_space(3))) void*)(smem + g * 1024), 16, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < (SR + NT - 1) / NT; ++k) {
                const int rr = k * NT + tid;
                if (rr < SR) {
                    const int slot = st * SR + rr;
                    const int tile = rr >> 5, mm = rr & 31;
                    const int hh = (mm >> 2) & 1, reg = (mm & 3) + 4 * (mm >> 3);
                    int cinit = -(nmv[k] >> 1), low = ((1 - (nmv[k] & 1)) << 4) | (15 - reg);
                    if (slot >= nq) { cinit = kPadCinit; low = 15 - reg; }
                    int* aux = (int*)(smem + kRowBytes) + tile * kAuxPerTile;
                    aux[16 * hh + reg] = cinit;
                    aux[32 + 16 * hh + reg] = low;
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
            if (staged && st == 0) {
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        bf[j][c] = *(const v4i*)(cell0 + (32 * (blk0 + j)) * kDim + aoff[c]);
            }
            X1_STAMP(9);
            const int ntiles = min(SR / kTileRows, (nq - st * SR + kTileRows - 1) / kTileRows);
            for (int tt = tphase; tt < ntiles; tt += 4) {         // this wave's tiles, ascending
                v4i af[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) af[c] = *(const v4i*)(smem + tt * (kTileRows * kDim) + aoff[c]);
                const v16i ci = lds_read16(smem + xoff + tt * (kAuxPerTile * 4));
                // the exact path's low words, unconditionally and with the operands: in a round's few tiles nearly every
                // block takes that path, and each visit otherwise waits for an LDS round trip of its own
                const v16i low = lds_read16(smem + xoff + tt * (kAuxPerTile * 4) + 128);
                v16i acc[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[0], bf[j][0], ci, 0, 0, 0);
#pragma unroll
                for (int c = 1; c < 4; ++c)
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[c], bf[j][c], acc[j], 0, 0, 0);
                // No threshold test in front of the update (the dense kernel's fast path): a wave sees a handful of
                // tiles per chunk, and one of its 64 lanes improves its best in nearly every one of them (the k-th tile
                // of a lane does with probability 1/k) -- the test only added its nine instructions to every visit.
#pragma unroll
                for (int j = 0; j < NB; ++j) top[j].update(acc[j], low, st * (SR / kTileRows) + tt);
            }
        }
        X1_STAMP(10);
        // lane halves, then the four waves through tbest
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int mh = top[j].hi();
            int mi = (top[j].tile >= 0) ? top[j].index(h) : -1;
            if (mi >= nq) mi = -1;                  // padding slot
            const int oh = __shfl_xor(mh, 32);
            const int oi = __shfl_xor(mi, 32);
            const bool mine = !better(oh, oi, mh, mi);
            const int rh = mine ? mh : oh;
            const int ri = mine ? mi : oi;
            const int n = cb0 + 32 * (blk0 + j) + (lane & 31);
            if (h == 0 && n < nt && ri >= 0) {
                const unsigned d2 = (unsigned)(tnorm[j] + 1 - rh);
                atomicMin(&tbest[32 * (blk0 + j) + (lane & 31)], ((unsigned long long)d2 << 32) | (unsigned)ri);
            }
        }
        lds_barrier();
        bool tied = false;                               // (rows whose election the float32 root may change)
        if (tid < 128 && cb0 + tid < nt) {
            const unsigned long long tb = tbest[tid];
            if (tb != ~0ull) {
                if constexpr (MERGE) {
                    const unsigned long long g = tb + slot_base;        // (the slot is the low word and stays below 2^31)
                    if (g < tb_all[cb0 + tid]) tb_all[cb0 + tid] = g;
                } else {
                    const unsigned d2 = (unsigned)(tb >> 32);
                    tied = tie_guard && d2 >= kSqrtTieMin && sqrt_ties_up(d2);
                    if (!tied)
                        atomicMin(&qbest[(unsigned)tb], ((unsigned long long)(tie_guard ? sqrt_bits(d2) : d2) << 32) | (unsigned)(cb0 + tid));
                }
            }
        }
        if (!MERGE && tie_guard) {
            // Cold path (never with SIFT-range descriptors): train row n's best d2 shares its float32
            // root with d2 + 1, so a query slot at d2 + 1 with a LOWER slot number is the one OpenCV
            // elects.  Exact rescan of the slots below the elected one, row by row.
            if (tid < 128) {
                const unsigned long long m = __builtin_amdgcn_ballot_w64(tied);
                if (lane == 0) tbest[128 + wave] = m;
            }
            lds_barrier();
            for (int w = 0; w < 2; ++w) {
                unsigned long long m = tbest[128 + w];          // (the same in every lane: kept scalar)
                m = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(m >> 32)) << 32) |
                    (unsigned)__builtin_amdgcn_readfirstlane((unsigned)m);
                while (m) {
                    const int f = 64 * w + (int)__builtin_ctzll(m);
                    m &= m - 1;
                    const unsigned long long tb = tbest[f];
                    const unsigned d2 = (unsigned)(tb >> 32), smin = (unsigned)tb;
                    unsigned* acc = (unsigned*)&tbest[130];
                    if (tid == 0) *acc = smin;
                    lds_barrier();
                    gptr<const int8_t> trow = t_rows8 + (size_t)(t0 + cb0 + f) * kDim;
                    const int tn = t_norm[t0 + cb0 + f];
                    for (unsigned s = tid; s < smin; s += NT) {
                        const int qi = q_rows[s];
                        if (exact_d2_i8(q_rows8 + (size_t)qi * kDim, q_norm[qi], trow, tn) == d2 + 1u) atomicMin(acc, s);
                    }
                    lds_barrier();
                    if (tid == 0) atomicMin(&qbest[*acc], ((unsigned long long)sqrt_bits(d2) << 32) | (unsigned)(cb0 + f));
                    lds_barrier();
                }
            }
        }
        lds_barrier();
        X1_STAMP(11);
    }
#undef X1_STAMP
}

}  // namespace fm
