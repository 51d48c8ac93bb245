// K9 -- exact k-NN lists for 3 <= k <= 8: cv2.BFMatcher(NORM_L2).knnMatch(dt1, dt2, k) for the k the reference's
// bf_match / flann_match signature admits (matchutil.py:39-43, 46-67: `k` is any int) but its own code never asks for
// (it calls k = 1 and k = 2 only: fastmatch.pyx:122-123, 161-162, cache.pyx:250-252, 271-273 -- those stay on the MFMA
// kernels).  Off the hot path: a plain vector-ALU kernel, one query row per thread, the train rows staged through LDS and
// read as broadcasts; 100k x 100k int8 rows take ~20 ms.
//   order of a list: (float32 bits of the distance, train index), strict, candidates in ascending train order -- OpenCV's
//   insertion; integer route: dist = sqrtf((float)d2) of the exact int32 d2, float32 route:
//   K5's chain (s = fmaf(v, v, s), k ascending; dist_f32.hip) -- the same bits fm_knn2 returns for its two columns.
// partial[(split * nq + q) * K + i] = (distance bits << 32) | train row, ascending, ~0 = none; knnk_merge_kernel merges
// the splits of the train range.
#include "tile_ops.h"

namespace fm {

constexpr int kKnnStage = 64;          // train rows per LDS stage

template <int K>
__device__ __forceinline__ void knnk_insert(unsigned long long (&keys)[K], unsigned long long key)
{
    // strict: an equal key cannot occur (indices differ); a later row never displaces an earlier one at equal distance
    // because its index is larger
#pragma unroll
    for (int i = K - 1; i >= 0; --i) {
        const unsigned long long prev = i > 0 ? keys[i - 1] : 0ull;
        const bool here = key < keys[i] && (i == 0 || !(key < prev));
        const bool shift = i > 0 && key < prev;
        keys[i] = shift ? prev : (here ? key : keys[i]);
    }
}

template <int K>
__global__ __launch_bounds__(256)
void knnk_i8_kernel(const int8_t* __restrict__ qrows, const int32_t* __restrict__ qnorm, int nq,
                    const int8_t* __restrict__ trows, const int32_t* __restrict__ tnorm, int nt, int rows_per_split,
                    unsigned long long* __restrict__ partial)
{
    __shared__ __attribute__((aligned(16))) int8_t srow[kKnnStage * kDim];
    __shared__ int snorm[kKnnStage];
    const int tid = threadIdx.x;
    const int q = blockIdx.x * 256 + tid;
    const bool live = q < nq;
    v4i qv[kDim / 16];
#pragma unroll
    for (int c = 0; c < kDim / 16; ++c) qv[c] = live ? *(const v4i*)(qrows + (size_t)q * kDim + 16 * c) : v4i{0, 0, 0, 0};
    const int qn = live ? qnorm[q] : 0;
    unsigned long long keys[K];
#pragma unroll
    for (int i = 0; i < K; ++i) keys[i] = ~0ull;
    unsigned wd2 = 0xffffffffu;                    // a candidate needs d2 <= wd2 (the K-th entry's d2 + 1: float32 roots tie in pairs)
    const int t0 = blockIdx.y * rows_per_split, t1 = min(nt, t0 + rows_per_split);
    for (int base = t0; base < t1; base += kKnnStage) {
        __syncthreads();
        {   // 64 rows x 128 B: two 16-byte pieces per thread (rows beyond the bank's padding are never read: n_pad % 128 == 0)
            const int piece = tid * 2;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = (piece + i) >> 3, c = (piece + i) & 7;
                *(v4i*)(srow + r * kDim + 16 * c) = *(const v4i*)(trows + (size_t)(base + r) * kDim + 16 * c);
            }
            if (tid < kKnnStage) snorm[tid] = tnorm[base + tid];
        }
        __syncthreads();
        const int rn = min(kKnnStage, t1 - base);
        for (int r = 0; r < rn; ++r) {
            int dot = 0;
#pragma unroll
            for (int c = 0; c < kDim / 16; ++c) {
                const v4i y = *(const v4i*)(srow + r * kDim + 16 * c);
#pragma unroll
                for (int w = 0; w < 4; ++w) dot = __builtin_amdgcn_sdot4(qv[c][w], y[w], dot, false);
            }
            const unsigned d2 = (unsigned)(qn + snorm[r] - 2 * dot);
            if (d2 <= wd2) {
                const unsigned long long key = ((unsigned long long)sqrt_bits(d2) << 32) | (unsigned)(base + r);
                if (key < keys[K - 1]) {
                    knnk_insert<K>(keys, key);
                    if (keys[K - 1] != ~0ull) {
                        // the largest d2 whose root can still be <= the K-th entry's: that entry's d2, + 1 if it ties upward
                        const float f = __uint_as_float((unsigned)(keys[K - 1] >> 32));
                        unsigned hi = (unsigned)(f * f) + 4u;          // (f * f is within 2 of the entry's d2: d2 < 2^24)
                        while (hi > 0u && sqrt_bits(hi) > (unsigned)(keys[K - 1] >> 32)) --hi;
                        wd2 = hi;
                    }
                }
            }
        }
    }
    if (live) {
#pragma unroll
        for (int i = 0; i < K; ++i) partial[((size_t)blockIdx.y * nq + q) * K + i] = keys[i];
    }
}

template <int K>
__global__ __launch_bounds__(256)
void knnk_f32_kernel(const float* __restrict__ qrows, int nq, const float* __restrict__ trows, int nt, int rows_per_split,
                     unsigned long long* __restrict__ partial)
{
    constexpr int kStage = 32;                      // 32 rows x 512 B
    __shared__ __attribute__((aligned(16))) float srow[kStage * kDim];
    const int tid = threadIdx.x;
    const int q = blockIdx.x * 256 + tid;
    const bool live = q < nq;
    float4 qv[kDim / 4];                            // the thread's query row: 128 registers
#pragma unroll
    for (int k4 = 0; k4 < kDim / 4; ++k4) qv[k4] = ((const float4*)(qrows + (size_t)(live ? q : 0) * kDim))[k4];
    unsigned long long keys[K];
#pragma unroll
    for (int i = 0; i < K; ++i) keys[i] = ~0ull;
    const int t0 = blockIdx.y * rows_per_split, t1 = min(nt, t0 + rows_per_split);
    for (int base = t0; base < t1; base += kStage) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {               // 32 x 32 float4: four per thread
            const int f4 = tid + 256 * i;
            const int r = f4 >> 5, c = f4 & 31;
            ((float4*)srow)[f4] = (base + r < t1) ? *(const float4*)(trows + (size_t)(base + r) * kDim + 4 * c) : float4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
        const int rn = min(kStage, t1 - base);
        for (int r = 0; r < rn; ++r) {
            float sum = 0.f;
#pragma unroll
            for (int k4 = 0; k4 < kDim / 4; ++k4) {
                const float4 a = qv[k4];
                const float4 b = ((const float4*)srow)[r * (kDim / 4) + k4];
                float v;
                v = a.x - b.x; sum = __builtin_fmaf(v, v, sum);
                v = a.y - b.y; sum = __builtin_fmaf(v, v, sum);
                v = a.z - b.z; sum = __builtin_fmaf(v, v, sum);
                v = a.w - b.w; sum = __builtin_fmaf(v, v, sum);
            }
            const unsigned long long key = ((unsigned long long)__float_as_uint(sqrtf(sum)) << 32) | (unsigned)(base + r);
            if (key < keys[K - 1]) knnk_insert<K>(keys, key);
        }
    }
    if (live) {
#pragma unroll
        for (int i = 0; i < K; ++i) partial[((size_t)blockIdx.y * nq + q) * K + i] = keys[i];
    }
}

// One thread per query row: the K smallest keys over the splits' lists (each ascending); missing neighbours -1 / +inf.
template <int K>
__global__ __launch_bounds__(256)
void knnk_merge_kernel(const unsigned long long* __restrict__ partial, int nsplit, int nq, int32_t* __restrict__ idx, float* __restrict__ dist)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    unsigned long long keys[K];
#pragma unroll
    for (int i = 0; i < K; ++i) keys[i] = ~0ull;
    for (int s = 0; s < nsplit; ++s)
        for (int i = 0; i < K; ++i) {
            const unsigned long long key = partial[((size_t)s * nq + q) * K + i];
            if (!(key < keys[K - 1])) break;            // (a split's list ascends)
            knnk_insert<K>(keys, key);
        }
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const bool has = keys[i] != ~0ull;
        idx[(size_t)q * K + i] = has ? (int32_t)(unsigned)keys[i] : -1;
        dist[(size_t)q * K + i] = has ? __uint_as_float((unsigned)(keys[i] >> 32)) : INFINITY;
    }
}

int knnk_splits(int64_t nq, int64_t nt)
{
    // ~2048 workgroups on the chip; a split of at least 512 train rows
    const int64_t qblocks = (nq + 255) / 256;
    int64_t s = (2048 + qblocks - 1) / qblocks;
    if (s > (nt + 511) / 512) s = (nt + 511) / 512;
    if (s < 1) s = 1;
    if (s > 65535) s = 65535;
    return (int)s;
}

size_t knnk_partial_bytes(int64_t nq, int64_t nt, int k) { return (size_t)knnk_splits(nq, nt) * (size_t)nq * (size_t)k * 8; }

hipError_t launch_knnk(const Bank& q, const Bank& t, int k, unsigned long long* partial, int32_t* d_idx, float* d_dist, hipStream_t stream)
{
    if (k < 1 || k > 8 || q.kind != t.kind || q.n <= 0) return hipErrorInvalidValue;
    const int nq = (int)q.n, nt = (int)t.n;
    const int nsplit = knnk_splits(q.n, t.n);
    int per = (nt + nsplit - 1) / nsplit;
    per = (per + kKnnStage - 1) / kKnnStage * kKnnStage;          // whole stages (both kernels' stage sizes divide it)
    if (per < kKnnStage) per = kKnnStage;
    const dim3 grid((unsigned)((nq + 255) / 256), (unsigned)nsplit), mgrid((unsigned)((nq + 255) / 256));
#define FM_KNNK(K_)                                                                                                       \
    case K_:                                                                                                               \
        if (q.kind == FM_BANK_F32)                                                                                         \
            hipLaunchKernelGGL((knnk_f32_kernel<K_>), grid, dim3(256), 0, stream, (const float*)q.rowsf, nq, (const float*)t.rowsf, nt, per, partial); \
        else                                                                                                               \
            hipLaunchKernelGGL((knnk_i8_kernel<K_>), grid, dim3(256), 0, stream, (const int8_t*)q.rows8, (const int32_t*)q.norm, nq,   \
                               (const int8_t*)t.rows8, (const int32_t*)t.norm, nt, per, partial);                         \
        hipLaunchKernelGGL((knnk_merge_kernel<K_>), mgrid, dim3(256), 0, stream, (const unsigned long long*)partial, nsplit, nq, d_idx, d_dist); \
        break;
    switch (k) {
        FM_KNNK(1) FM_KNNK(2) FM_KNNK(3) FM_KNNK(4) FM_KNNK(5) FM_KNNK(6) FM_KNNK(7) FM_KNNK(8)
        default: return hipErrorInvalidValue;
    }
#undef FM_KNNK
    return hipGetLastError();
}

}  // namespace fm
