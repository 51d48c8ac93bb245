// One expansion round's cross-checked 1-NN for descriptors that are
// NOT integer valued (RootSIFT-style float32 banks), as a workgroup-wide device function shared
// by round_f32_kernel (rounds.hip) and expand_kernel (expand.hip).  256 threads; contains
// barriers: every thread of the block must call it.
//
// Same result, bit for bit, as the dense float32 route (K5 / K8: OpenCV's float32 chain
// s = fmaf(v_k, v_k, s), k ascending, dist = sqrtf(s); reference call site fastmatch.pyx:161-162),
// found the way K8 (filter_f16.hip) finds it, at round size:
//   sweep 0 A(t,q) = |t|^2 + |q|^2 - 2 t.q on the matrix cores (v_mfma_f32_16x16x32_f16 on the
//           banks' fp16 planes, accumulator started at -|q|^2/2, so a LARGER acc is a SMALLER
//           distance): per train row t the best acc over the query slots gathered so far;
//   sweep 1 the same tiles again: every (t, slot) with acc >= best(t) - M(t) goes to a candidate
//           list in LDS together with its acc.  M is K8's margin (|A - D| <= M for the exact
//           chain value D, see filter_f16.hip), so the exact nearest slot of t -- and every slot
//           tied with it -- is on the list;
//   exact   one thread per candidate that is still inside the FINAL bound evaluates the float32
//           chain; per train row the minimum of
//           (distance bits, slot) = cv::batchDistance's reverse nearest neighbour, lowest slot
//           on ties; then the scatter-min into qbest[slot] of (distance bits << 32 | t).
// Train rows are taken 128 at a time (each wave owns 32 of them for all query tiles), query
// slots 256 at a time (gathered by index into the swizzled LDS image K8 uses).
#pragma once
#include "tile_ops.h"
#include <type_traits>

namespace fm {

typedef _Float16 r_v8h __attribute__((ext_vector_type(8)));
typedef float    r_v4f __attribute__((ext_vector_type(4)));

constexpr int kRF_SR        = 256;                       // query slots per gather stage
constexpr int kRF_RowBytes  = 256;                       // fp16 row
constexpr int kRF_StageBytes = kRF_SR * kRF_RowBytes + kRF_SR * 4;   // rows + accumulator inits

struct RoundF32 {             // as the host fills it (generic pointers)
    const char*  q_rowsh;     // query bank: fp16 plane (scaled by 2^kq), 256 B per row
    const float* q_auxf;      //             -|row|^2/2 of the scaled row
    const float* q_rowsf;     //             float32 rows [n_pad][128] (exact chain)
    const char*  t_rowsh;     // train bank (all cells back to back)
    const float* t_normf;     //             |row|^2 of the scaled row
    const float* t_rowsf;
    float eps_c, eps_nm, aux_mul;   // K8's margin terms and accumulator-init factor (launch_filter)
};

struct RoundF32G {            // the same with global-memory pointers (gptr, tile_ops.h), for the kernels
    gptr<const char>  q_rowsh;
    gptr<const float> q_auxf, q_rowsf;
    gptr<const char>  t_rowsh;
    gptr<const float> t_normf, t_rowsf;
    float eps_c, eps_nm, aux_mul;
    __device__ __forceinline__ explicit RoundF32G(const RoundF32& r)
        : q_rowsh((gptr<const char>)r.q_rowsh), q_auxf((gptr<const float>)r.q_auxf), q_rowsf((gptr<const float>)r.q_rowsf),
          t_rowsh((gptr<const char>)r.t_rowsh), t_normf((gptr<const float>)r.t_normf), t_rowsf((gptr<const float>)r.t_rowsf),
          eps_c(r.eps_c), eps_nm(r.eps_nm), aux_mul(r.aux_mul) {}
};

__device__ __forceinline__ float rf_max3(float a, float b, float c)
{
    return __builtin_elementwise_maximum(__builtin_elementwise_maximum(a, b), c);
}

// qbest[slot], slot in [0, nq): pre-filled with ~0 by the caller; on return (after the caller's
// next barrier) (float32 distance bits << 32 | local train index) of the cross-checked match.
// clist: LDS scratch for clist_cap candidates (u32 each: slot | column << 12); tbest: LDS u64[128];
// sh: LDS int[NT / 64].  NT = threads of the workgroup (256 in round_f32_kernel, 512 in expand_kernel).
// Returns false (uniformly) if the candidate list overflowed: the round's result is then invalid.
// MERGE (K7's chunked rounds, expand.hip): the query slots are one chunk (slots slot_base .. slot_base + nq of a radius
// subset that does not fit LDS); the function stops at the reverse-NN step and folds its per-train-row minimum
// (distance bits << 32 | global slot) -- exact within the chunk, margin and all -- into tb_all[0 .. nt) (global; row
// cb0 + tid belongs to thread tid, as in x1_round_wsplit).  qbest is not touched.
template <int NT = 256, bool MERGE = false>
__device__ __forceinline__ bool x1_round_f32(const RoundF32G& R, const int* q_rows, int nq, int64_t t0, int nt,
                                             char* smem, unsigned long long* qbest, unsigned* clist, int clist_cap,
                                             unsigned long long* tbest, int* sh,
                                             long long* pt = nullptr, long long* ts = nullptr,
                                             gptr<unsigned long long> tb_all = nullptr, unsigned slot_base = 0)
{
#define RF_STAMP(k) do { if (pt && threadIdx.x == 0) { const long long _n = wall_clock64(); pt[k] += _n - *ts; *ts = _n; } } while (0)
    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int g    = lane >> 4;
    const int c16  = lane & 15;
    int aoff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) aoff[s] = c16 * kRF_RowBytes + 16 * ((4 * s + g) ^ c16);
    const int xoff = kRF_SR * kRF_RowBytes + 16 * g;
    constexpr int NW = NT / 64;                          // waves of the workgroup
    constexpr int kRF_NC = 128 / (16 * NW);              // 16-column blocks per wave: the waves share a 128-column chunk
    constexpr int kPieces = (kRF_SR / 4) / NW;           // 1-KiB gather pieces (4 rows) per wave
    static_assert(kRF_NC >= 1 && kPieces >= 1, "4 or 8 waves");
    const int nstages = (nq + kRF_SR - 1) / kRF_SR;
    bool ok = true;

    // Gather stage st of the query subset into the swizzled image by LDS-DMA (no registers, all
    // 16 pieces of a wave in flight at once).  A piece = 4 rows x 256 B; lane (r = lane >> 4,
    // p = lane & 15) fetches source chunk p ^ (row & 15) of its row, which the DMA stores at chunk
    // position p.  Slots past the subset fetch the last real row; their accumulator init
    // (-3.4e38) keeps them below every bound.
    auto gather = [&](int st) {
        lds_barrier();                                  // previous image fully consumed
#pragma unroll
        for (int i = 0; i < kPieces; ++i) {
            const int piece = wave * kPieces + i;         // 64 pieces of 4 rows per stage
            const int row = piece * 4 + (lane >> 4);
            const int slot = st * kRF_SR + row;
            const int qi = q_rows[slot < nq ? slot : nq - 1];
            gptr<const char> src = R.q_rowsh + (size_t)qi * kRF_RowBytes + 16 * ((lane & 15) ^ (row & 15));
            __builtin_amdgcn_global_load_lds((gptr<const void>)src,
                                             (__attribute__((address_space(3))) void*)(smem + piece * 1024), 16, 0, 0);
        }
        if (tid < kRF_SR) {
            const int aslot = st * kRF_SR + tid;
            const float aux = aslot < nq ? R.q_auxf[q_rows[aslot]] * R.aux_mul : -3.4e38f;
            ((float*)(smem + kRF_SR * kRF_RowBytes))[tid] = aux;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
    };

    int loaded = -1;                                      // stage whose image is in LDS (uniform)
    for (int cb0 = 0; cb0 < nt; cb0 += 128) {
        if (tid < 128) tbest[tid] = ~0ull;
        const int wcap = clist_cap / NW;                   // every wave appends to its own part of the list
        unsigned* const wlist = clist + wave * wcap;
        int wcount = 0;                                    // wave uniform
        // this wave's 32 train rows (two 16-column blocks), stationary for the whole chunk
        r_v8h bh[kRF_NC][4];
        float marg[kRF_NC];
#pragma unroll
        for (int j = 0; j < kRF_NC; ++j) {
            const int n = cb0 + wave * (16 * kRF_NC) + 16 * j + c16;
            const bool in = n < nt;
            marg[j] = in ? fmaf(R.eps_c, R.t_normf[t0 + n], R.eps_nm) : 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                v4i h = v4i{0, 0, 0, 0};
                if (in) h = *(gptr<const v4i>)(R.t_rowsh + (size_t)(t0 + n) * kRF_RowBytes + (4 * s + g) * 16);
                bh[j][s] = __builtin_bit_cast(r_v8h, h);
            }
        }
        float best[kRF_NC];                    // column maximum over the query slots (all lane groups after sweep 0)
#pragma unroll
        for (int j = 0; j < kRF_NC; ++j) best[j] = -3.0e38f;

        // One MFMA sweep over the tiles of the stage whose image is in LDS.  PASS 0: column maxima.
        // PASS 1: one bit per accumulator value, "acc >= best - M" (two VALU per value, no branch,
        // no store: with one wave per SIMD every dependent scalar round trip is exposed latency,
        // and almost every tile holds a candidate somewhere in the wave).  The bits are turned into
        // list entries once per stage.  Fragments of tile k + 1 are read while tile k's MFMAs run.
        float thr[kRF_NC];
        unsigned long long bits[kRF_NC];
        auto sweep = [&](auto pass_tag, int st) __attribute__((always_inline)) {
            constexpr int PASS = decltype(pass_tag)::value;
            const int ntiles = min(kRF_SR / 16, (nq - st * kRF_SR + 15) / 16);
            // two 16-row tiles per step: four independent accumulator chains keep the MFMA pipe
            // fed from a single wave, and the next step's fragments are read while they run
            r_v8h fn[2][4];
            r_v4f cn[2];
            auto load_frag = [&](int k) __attribute__((always_inline)) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int kk = min(k + h, kRF_SR / 16 - 1);     // (a tile past the end is computed and ignored)
                    const char* rows = smem + kk * 16 * kRF_RowBytes;
                    cn[h] = *(const r_v4f*)(smem + xoff + kk * 64);
#pragma unroll
                    for (int s = 0; s < 4; ++s) fn[h][s] = __builtin_bit_cast(r_v8h, *(const v4i*)(rows + aoff[s]));
                }
            };
            load_frag(0);
            for (int k = 0; k < ntiles; k += 2) {
                r_v8h fs[2][4];
                r_v4f ci[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    ci[h] = cn[h];
#pragma unroll
                    for (int s = 0; s < 4; ++s) fs[h][s] = fn[h][s];
                }
                if (k + 2 < ntiles) load_frag(k + 2);
                r_v4f acc[2][kRF_NC];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int j = 0; j < kRF_NC; ++j) acc[h][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fs[h][0], bh[j][0], ci[h], 0, 0, 0);
#pragma unroll
                for (int s = 1; s < 4; ++s)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int j = 0; j < kRF_NC; ++j) acc[h][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fs[h][s], bh[j][s], acc[h][j], 0, 0, 0);
                const bool second = k + 1 < ntiles;                 // uniform
                if constexpr (PASS == 0) {
#pragma unroll
                    for (int j = 0; j < kRF_NC; ++j) {
                        best[j] = rf_max3(rf_max3(acc[0][j][0], acc[0][j][1], acc[0][j][2]), acc[0][j][3], best[j]);
                        const float m1 = rf_max3(rf_max3(acc[1][j][0], acc[1][j][1], acc[1][j][2]), acc[1][j][3], best[j]);
                        best[j] = second ? m1 : best[j];
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < kRF_NC; ++j) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            bits[j] = (bits[j] << 1) | (acc[0][j][r] >= thr[j] ? 1ull : 0ull);
                        if (second) {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                bits[j] = (bits[j] << 1) | (acc[1][j][r] >= thr[j] ? 1ull : 0ull);
                        }
                    }
                }
            }
            if constexpr (PASS == 1) {
                // value v = 4 k + r of this lane sits at bit 4 ntiles - 1 - v; slot = st * SR + 16 k + 4 g + r
                const int nv = 4 * ntiles;
#pragma unroll
                for (int j = 0; j < kRF_NC; ++j) {
                    const int tl = wave * (16 * kRF_NC) + 16 * j + c16;       // column within the 128-chunk
                    unsigned long long b = (cb0 + tl < nt) ? bits[j] : 0ull;
                    for (;;) {
                        const bool has = b != 0ull;
                        const unsigned long long mask = __builtin_amdgcn_ballot_w64(has);
                        if (mask == 0ull) break;
                        const int p = has ? 63 - __clzll((long long)b) : 0;
                        const int v = nv - 1 - p;
                        const int slot = st * kRF_SR + 16 * (v >> 2) + 4 * g + (v & 3);
                        const bool want = has && slot < nq;
                        const unsigned long long wmask = __builtin_amdgcn_ballot_w64(want);
                        if (want) {
                            const int pos = wcount + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(wmask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)wmask, 0u));
                            if (pos < wcap) wlist[pos] = (unsigned)slot | ((unsigned)tl << 12);
                        }
                        wcount += __popcll(wmask);
                        b &= ~(1ull << p);
                    }
                }
            }
        };

        // sweep 0 over every stage, then sweep 1 walking the stages backwards so that the image the
        // first pass left in LDS is reused: nstages + (nstages - 1) gathers per 128-column chunk
        RF_STAMP(8);
        for (int st = 0; st < nstages; ++st) {
            if (loaded != st) { gather(st); loaded = st; }
            RF_STAMP(9);
            sweep(std::integral_constant<int, 0>{}, st);
            RF_STAMP(6);
        }
        // the four lane groups hold disjoint slot subsets of the same columns
#pragma unroll
        for (int j = 0; j < kRF_NC; ++j) {
            float b = best[j];
            b = fmaxf(b, __shfl_xor(b, 16));
            b = fmaxf(b, __shfl_xor(b, 32));
            thr[j] = b - marg[j];
        }
        for (int st = nstages - 1; st >= 0; --st) {
            if (loaded != st) { gather(st); loaded = st; }
            RF_STAMP(9);
#pragma unroll
            for (int j = 0; j < kRF_NC; ++j) bits[j] = 0ull;
            sweep(std::integral_constant<int, 1>{}, st);
            RF_STAMP(7);
        }
        RF_STAMP(10);
        if (lane == 0) sh[wave] = wcount;
        lds_barrier();
        int cnt[NW], ncand = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const int c = sh[w];
            if (c > wcap) ok = false;
            cnt[w] = min(c, wcap);
            ncand += cnt[w];
        }
        // exact float32 chain per candidate (K5's order: k ascending, v = a - b, s = fma(v, v, s))
        for (int i = tid; i < ncand; i += NT) {
            int w = 0, o = i;
#pragma unroll
            for (int q = 0; q < NW - 1; ++q) { const bool past = w == q && o >= cnt[q]; o -= past ? cnt[q] : 0; w += past ? 1 : 0; }
            const unsigned e = clist[w * wcap + o];
            const int slot = (int)(e & 4095u), tl = (int)(e >> 12);
            gptr<const r_v4f> cp = (gptr<const r_v4f>)(R.t_rowsf + (size_t)(t0 + cb0 + tl) * kDim);
            gptr<const r_v4f> rp = (gptr<const r_v4f>)(R.q_rowsf + (size_t)q_rows[slot] * kDim);
            float sum = 0.f;
#pragma unroll 16
            for (int k4 = 0; k4 < kDim / 4; ++k4) {
                const r_v4f a = cp[k4];
                const r_v4f b = rp[k4];
                float v;
                v = a[0] - b[0]; sum = __builtin_fmaf(v, v, sum);
                v = a[1] - b[1]; sum = __builtin_fmaf(v, v, sum);
                v = a[2] - b[2]; sum = __builtin_fmaf(v, v, sum);
                v = a[3] - b[3]; sum = __builtin_fmaf(v, v, sum);
            }
            atomicMin(&tbest[tl], ((unsigned long long)__float_as_uint(sqrtf(sum)) << 32) | (unsigned)slot);
        }
        lds_barrier();
        RF_STAMP(11);
        if (tid < 128 && cb0 + tid < nt) {
            const unsigned long long tb = tbest[tid];
            if (tb != ~0ull) {
                if constexpr (MERGE) {
                    const unsigned long long g = tb + slot_base;
                    if (g < tb_all[cb0 + tid]) tb_all[cb0 + tid] = g;
                } else {
                    atomicMin(&qbest[(unsigned)tb], (tb & 0xffffffff00000000ull) | (unsigned)(cb0 + tid));
                }
            }
        }
        lds_barrier();
    }
#undef RF_STAMP
    return ok;
}

}  // namespace fm
