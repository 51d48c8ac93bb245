// Device helpers shared by the dense row-reduce kernel (rowreduce.hip) and the per-round
// kernel (rounds.hip): MFMA vector types, stage geometry, the in-lane max tree and the
// exact top-K update.
#pragma once
#include "fm_internal.h"

namespace fm {

typedef int v4i  __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int kStageRowBytes = kStageRows * kDim;           // 16384
constexpr int kStageAuxBytes = (kStageRows / kTileRows) * kAuxPerTile * 4;  // 1024
constexpr int kStageBytes    = kStageRowBytes + kStageAuxBytes;             // 17408

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

// Exact update of a lane's top-KTOP with the 16 candidates of one tile.
// hi = 2*acc + npar orders candidates by descending (d2 ascending); candidates arrive in
// ascending row index within a lane, so strict '>' keeps the lower index on ties
// (cv::batchDistance insertion rule).
template <int KTOP>
__device__ __forceinline__ void exact_update(const v16i& acc, const v16i& np, int idx_base, int nred,
                                             int (&bh)[KTOP], int (&bi)[KTOP])
{
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int hi  = (acc[r] << 1) | np[r];
        const int idx = idx_base + (r & 3) + 8 * (r >> 2);
        const bool ok = idx < nred;
        if constexpr (KTOP == 1) {
            if (ok && hi > bh[0]) { bh[0] = hi; bi[0] = idx; }
        } else {
            if (ok && hi > bh[1]) {
                if (hi > bh[0]) { bh[1] = bh[0]; bi[1] = bi[0]; bh[0] = hi; bi[0] = idx; }
                else            { bh[1] = hi;    bi[1] = idx; }
            }
        }
    }
}

// (hi, idx) a is better than b: larger hi, then lower index.  idx < 0 means "none".
__device__ __forceinline__ bool better(int ah, int ai, int bh_, int bi_)
{
    if (ai < 0) return false;
    if (bi_ < 0) return true;
    return ah > bh_ || (ah == bh_ && ai < bi_);
}


// Read the 16 accumulator-order words (cinit or npar) of one tile from LDS.
__device__ __forceinline__ v16i lds_read16(const char* p)
{
    const v4i* ax = (const v4i*)p;
    const v4i c0 = ax[0], c1 = ax[1], c2 = ax[2], c3 = ax[3];
    return v16i{c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3],
                c2[0], c2[1], c2[2], c2[3], c3[0], c3[1], c3[2], c3[3]};
}

}  // namespace fm
