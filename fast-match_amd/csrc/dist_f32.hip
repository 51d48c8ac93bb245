// K5 -- float32 route of the row-reduce for descriptors that are NOT integer valued.
//
// Same contract as rowreduce.hip (top-KTOP of (distance, index) over the rows of bank `red`
// for every row of bank `cols`, per split of the reduction range), but the distance is
// OpenCV's float32 L2 (cv::batchDistance):
//     d2 = sum_k (a_k - b_k)^2 accumulated in float32,   dist = sqrtf(d2).
// The accumulation order is FIXED and mirrored by the oracle (orc_*_f32, order 1):
//     s = fmaf(v_k, v_k, s), k = 0 .. 127 ascending, v_k = a_k - b_k in float32,
// so results are bit-comparable (OpenCV's own order is build dependent).  The GEMM form
// |a|^2 + |b|^2 - 2ab is NOT used here: in float32 it cancels catastrophically for near
// neighbours.  Candidates are compared on dist (the float32 sqrt), strictly, in ascending
// row order -- exactly what OpenCV's insertion does -- but sqrtf is evaluated only for
// candidates that can still enter a lane's top-K (d2 <= B*B*(1+2^-22), B = current K-th
// best distance), so the hot loop is 2 VALU ops per pair-dimension (v_sub + v_fma).
//
// Tiling: 256 threads compute a 64 (output rows) x 64 (reduced rows) block per step;
// thread (tn = tid & 15, tm = tid >> 4) owns a 4 x 4 patch; both operands are staged
// k-major in LDS ([k][row], 128 x 64 floats each) and read with ds_read_b128.
// Bound: fp32 VALU (157 TFLOP/s) = 3.1e11 pairs/s at 2 ops x 128 dims per pair.
#include "fm_internal.h"

namespace fm {

constexpr int kF32Tile = 64;                 // rows per block step, both sides
constexpr int kF32Ld   = kF32Tile + 4;       // padded leading dimension (floats) of the k-major images

struct F32Params {
    const float* col_rows;    // [ncols_pad][128]
    int          ncols_pad;
    const float* red_rows;    // [nred_pad][128]
    int          nred;        // real rows
    int          nsteps;      // ceil(nred_pad / 64)
    int          nred_pad;    // rows the reduced bank's arrays hold
    int          nsplit;
    int          steps_per_split;
    int          ncols_alloc;
    unsigned long long* partial;
    const int*   run_flag;    // null, or: skip the whole launch unless *run_flag != 0
    int          self;        // the banks are one bank: row n is not a candidate for output row n (fm_self_dist)
};

__device__ __forceinline__ void load_tile_kmajor(const float* __restrict__ rows, int row0, int row_end, float* __restrict__ img, int tid)
{
    // 64 rows x 128 floats: thread t reads float4 #(t + 256*i) of the tile (coalesced along k)
    // and scatters it into the k-major image img[k][row].  Rows from row_end on are not read (zeros): a VIEW into a
    // bank -- one cell's rows from an arbitrary first row, round_xcheck_dense -- is clipped to the bank's allocation,
    // so its padded size need not be a multiple of this tile (up to 63 rows past a hipMalloc'ed array).
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int f4 = tid + 256 * i;              // 0 .. 2047
        const int r = f4 >> 5, k4 = (f4 & 31) * 4;
        const float4 v = (row0 + r < row_end) ? *(const float4*)(rows + (size_t)(row0 + r) * kDim + k4) : float4{0.f, 0.f, 0.f, 0.f};
        img[(k4 + 0) * kF32Ld + r] = v.x;
        img[(k4 + 1) * kF32Ld + r] = v.y;
        img[(k4 + 2) * kF32Ld + r] = v.z;
        img[(k4 + 3) * kF32Ld + r] = v.w;
    }
}

template <int KTOP>
__global__ __launch_bounds__(256)
void rowreduce_f32_kernel(F32Params p)
{
    __shared__ __attribute__((aligned(16))) float colimg[kDim * kF32Ld];
    __shared__ __attribute__((aligned(16))) float redimg[kDim * kF32Ld];

    if (p.run_flag) {
        if (*p.run_flag == 0) return;                 // the fp16 filter's result stands
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd((int*)p.run_flag + 2, 1);
    }
    const int tid = threadIdx.x;
    const int tn = tid & 15, tm = tid >> 4;
    const int split = blockIdx.x % p.nsplit;
    const int chunk = blockIdx.x / p.nsplit;
    const int c0 = chunk * kF32Tile;

    if (c0 < p.ncols_pad) load_tile_kmajor(p.col_rows, c0, p.ncols_pad, colimg, tid);

    float bd[4][KTOP];      // best distances per owned column (ascending)
    int   bi[4][KTOP];
    float thr[4];           // d2 above this cannot enter
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        thr[i] = INFINITY;
#pragma unroll
        for (int k = 0; k < KTOP; ++k) { bd[i][k] = INFINITY; bi[i][k] = -1; }
    }

    const int s0 = split * p.steps_per_split;
    const int s1 = min(s0 + p.steps_per_split, p.nsteps);
    for (int st = s0; st < s1; ++st) {
        __syncthreads();                                   // previous redimg fully consumed
        load_tile_kmajor(p.red_rows, st * kF32Tile, p.nred_pad, redimg, tid);
        __syncthreads();
        float s[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) s[i][j] = 0.f;
#pragma unroll 8
        for (int k = 0; k < kDim; ++k) {
            const float4 a = *(const float4*)(colimg + k * kF32Ld + 4 * tn);
            const float4 b = *(const float4*)(redimg + k * kF32Ld + 4 * tm);
            const float av[4] = {a.x, a.y, a.z, a.w};
            const float bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = av[i] - bv[j];
                    s[i][j] = __builtin_fmaf(v, v, s[i][j]);
                }
        }
        const int m0 = st * kF32Tile + 4 * tm;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float mn = fminf(fminf(s[i][0], s[i][1]), fminf(s[i][2], s[i][3]));
            if (mn <= thr[i]) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int idx = m0 + j;
                    if (idx < p.nred && s[i][j] <= thr[i] && !(p.self && idx == c0 + 4 * tn + i)) {
                        const float d = sqrtf(s[i][j]);
                        if (d < bd[i][KTOP - 1]) {            // strict: earlier row wins ties
                            if constexpr (KTOP == 2) {
                                if (d < bd[i][0]) { bd[i][1] = bd[i][0]; bi[i][1] = bi[i][0]; bd[i][0] = d; bi[i][0] = idx; }
                                else              { bd[i][1] = d; bi[i][1] = idx; }
                            } else {
                                bd[i][0] = d; bi[i][0] = idx;
                            }
                            const float B = bd[i][KTOP - 1];
                            thr[i] = (B * B) * 1.00000024f;    // d2 > thr  =>  sqrtf(d2) >= B
                        }
                    }
                }
            }
        }
    }

    // merge the 16 threads (tm) that own the same column through LDS (reuse redimg)
    __syncthreads();
    float* md = redimg;                                        // [64 cols][16][KTOP]
    int* mi = (int*)(redimg + kF32Tile * 16 * KTOP);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < KTOP; ++k) {
            md[((4 * tn + i) * 16 + tm) * KTOP + k] = bd[i][k];
            mi[((4 * tn + i) * 16 + tm) * KTOP + k] = bi[i][k];
        }
    __syncthreads();
    if (tid < kF32Tile) {
        const int n = c0 + tid;
        float rd[KTOP]; int ri[KTOP];
#pragma unroll
        for (int k = 0; k < KTOP; ++k) { rd[k] = INFINITY; ri[k] = -1; }
        for (int t = 0; t < 16; ++t)
#pragma unroll
            for (int k = 0; k < KTOP; ++k) {
                const float d = md[(tid * 16 + t) * KTOP + k];
                const int ix = mi[(tid * 16 + t) * KTOP + k];
                if (ix < 0) continue;
                // (dist, index) lexicographic; the 16 threads hold interleaved row sets
                const bool b0 = ri[0] < 0 || d < rd[0] || (d == rd[0] && ix < ri[0]);
                if constexpr (KTOP == 2) {
                    const bool b1 = ri[1] < 0 || d < rd[1] || (d == rd[1] && ix < ri[1]);
                    if (b0) { rd[1] = rd[0]; ri[1] = ri[0]; rd[0] = d; ri[0] = ix; }
                    else if (b1) { rd[1] = d; ri[1] = ix; }
                } else {
                    if (b0) { rd[0] = d; ri[0] = ix; }
                }
            }
        if (n < p.ncols_alloc) {
            unsigned long long* out = p.partial + ((size_t)split * p.ncols_alloc + n) * KTOP;
#pragma unroll
            for (int k = 0; k < KTOP; ++k)
                out[k] = (ri[k] >= 0 && n < p.ncols_pad)
                    ? (((unsigned long long)__float_as_uint(rd[k]) << 32) | (unsigned)ri[k]) : ~0ull;
        }
    }
}

RowReducePlan plan_rowreduce_f32(int64_t ncols_pad, int64_t nred_pad, int force_nsplit)
{
    RowReducePlan pl;
    pl.nb = 0;
    pl.nchunks = (int)((ncols_pad + kF32Tile - 1) / kF32Tile);
    if (pl.nchunks < 1) pl.nchunks = 1;
    pl.ncols_alloc = pl.nchunks * kF32Tile;
    const int64_t nsteps = (nred_pad + kF32Tile - 1) / kF32Tile;
    int64_t nsplit = (4096 + pl.nchunks - 1) / pl.nchunks;
    if (nsplit > nsteps / 8) nsplit = nsteps / 8;
    if (nsplit < 1) nsplit = 1;
    if (force_nsplit > 0) nsplit = force_nsplit;
    if (nsplit > nsteps) nsplit = nsteps > 0 ? nsteps : 1;
    int64_t per = (nsteps + nsplit - 1) / nsplit;
    if (per < 1) per = 1;
    nsplit = (nsteps + per - 1) / per;
    if (nsplit < 1) nsplit = 1;
    pl.nsplit = (int)nsplit;
    pl.stages_per_split = (int)per;
    return pl;
}

hipError_t launch_rowreduce_f32(const Bank& cols, const Bank& red, int ktop, const RowReducePlan& plan,
                                unsigned long long* partial, const int* run_flag, hipStream_t stream, bool self)
{
    F32Params p;
    p.run_flag = run_flag;
    p.self = self ? 1 : 0;
    p.col_rows = cols.rowsf;
    p.ncols_pad = (int)cols.n_pad;
    p.red_rows = red.rowsf;
    p.nred = (int)red.n;
    p.nsteps = (int)((red.n_pad + kF32Tile - 1) / kF32Tile);
    p.nred_pad = (int)red.n_pad;
    p.nsplit = plan.nsplit;
    p.steps_per_split = plan.stages_per_split;
    p.ncols_alloc = plan.ncols_alloc;
    p.partial = partial;
    const int grid = plan.nchunks * plan.nsplit;
    if (ktop == 1) hipLaunchKernelGGL((rowreduce_f32_kernel<1>), dim3(grid), dim3(256), 0, stream, p);
    else           hipLaunchKernelGGL((rowreduce_f32_kernel<2>), dim3(grid), dim3(256), 0, stream, p);
    return hipGetLastError();
}

}  // namespace fm
