// Ablation harness for the row-reduce kernel structure (not part of the product).
// Build on the GPU box: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I fast-match_amd/csrc scripts/ablate/ablate.hip -o /tmp/ablate
#include "tile_ops.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
using namespace fm;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

struct P {
    const int8_t* col_rows; const int8_t* red_rows; const int32_t* red_aux;
    int ncols_pad, nstages, nsplit, stages_per_split, ncols_alloc; int* out;
};

__device__ __forceinline__ void issue_stage(const P& p, int stage, char* buf, int wave, int lane)
{
    const int8_t* src_rows = p.red_rows + (size_t)stage * kStageRowBytes;
    const int slot = lane & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int g = wave * 4 + i;
        const int row = g * 8 + (lane >> 3);
        const int8_t* src = src_rows + row * kDim + 16 * (slot ^ ((row >> 1) & 7));
        __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(buf + g * 1024), 16, 0, 0);
    }
    if (wave == 0) {
        const int32_t* src = p.red_aux + (size_t)stage * (kStageAuxBytes / 4) + lane * 4;
        __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(buf + kStageRowBytes), 16, 0, 0);
    }
}

// MODE bits: 1 = skip epilogue, 2 = hoist LDS reads out of the tile loop, 4 = no staging/barriers
template <int NB, int WPS, int MODE>
__global__ __launch_bounds__(256, WPS) void k_base(P p)
{
    __shared__ __attribute__((aligned(16))) char smem[2 * kStageBytes];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int split = blockIdx.x % p.nsplit, chunk = blockIdx.x / p.nsplit;
    const int cb = chunk * (128 * NB) + wave * (32 * NB);
    v4i bf[NB][4];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int n = cb + 32 * j + (lane & 31);
#pragma unroll
        for (int c = 0; c < 4; ++c) bf[j][c] = *(const v4i*)(p.col_rows + (size_t)(n % p.ncols_pad) * kDim + 32 * c + 16 * h);
    }
    TopK<1> top[NB]; int thr[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) { top[j].init(); thr[j] = INT32_MIN; }
    const int st0 = split * p.stages_per_split, st1 = min(st0 + p.stages_per_split, p.nstages);
    const int sw = ((lane & 31) >> 1) & 7;
    int aoff[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) aoff[c] = (lane & 31) * kDim + 16 * ((2 * c + h) ^ sw);
    const int xoff = kStageRowBytes + h * 64;
    if (st0 < st1) issue_stage(p, st0, smem, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    v4i haf[4]; v16i hci;
    if (MODE & 2) {
#pragma unroll
        for (int c = 0; c < 4; ++c) haf[c] = *(const v4i*)(smem + aoff[c]);
        hci = lds_read16(smem + xoff);
    }
    for (int st = st0; st < st1; ++st) {
        char* buf = smem + ((st - st0) & 1) * kStageBytes;
        if (!(MODE & 4)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (st + 1 < st1) issue_stage(p, st + 1, smem + ((st + 1 - st0) & 1) * kStageBytes, wave, lane);
        }
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            v4i af[4]; v16i ci;
            if (MODE & 2) {
#pragma unroll
                for (int c = 0; c < 4; ++c) { af[c] = haf[c];
#if defined(__HIP_DEVICE_COMPILE__)
                    asm volatile("" : "+v"(af[c]));
#endif
                }
                ci = hci;
#if defined(__HIP_DEVICE_COMPILE__)
                asm volatile("" : "+v"(ci));
#endif
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) af[c] = *(const v4i*)(buf + tt * 4096 + aoff[c]);
                ci = lds_read16(buf + xoff + tt * 256);
            }
            v16i acc[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[0], bf[j][0], ci, 0, 0, 0);
#pragma unroll
            for (int c = 1; c < 4; ++c)
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[c], bf[j][c], acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (MODE & 1) {
#if defined(__HIP_DEVICE_COMPILE__)
                    asm volatile("" :: "v"(acc[j]));
#endif
                } else {
                    const int tmax = max16(acc[j]);
                    if (__builtin_amdgcn_ballot_w64(tmax >= thr[j]) != 0ull) {
                        const v16i low = lds_read16(buf + xoff + tt * 256 + 128);
                        top[j].update(acc[j], low, st * 4 + tt);
                        thr[j] = top[j].own_threshold();
                    }
                }
            }
        }
    }
    int s = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) s += top[j].key[0] + top[j].tile[0];
    if (s == 0x7fffffff) p.out[blockIdx.x * 256 + tid] = s;
}

// Rotated, software-pipelined schedule: j-major MFMA chains; the epilogue of (tile, j) runs
// in the shadow of the next chain; A fragments / cinit of the next tile are prefetched.
template <int NB, int WPS>
__global__ __launch_bounds__(256, WPS) void k_pipe(P p)
{
    __shared__ __attribute__((aligned(16))) char smem[2 * kStageBytes];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int split = blockIdx.x % p.nsplit, chunk = blockIdx.x / p.nsplit;
    const int cb = chunk * (128 * NB) + wave * (32 * NB);
    v4i bf[NB][4];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int n = cb + 32 * j + (lane & 31);
#pragma unroll
        for (int c = 0; c < 4; ++c) bf[j][c] = *(const v4i*)(p.col_rows + (size_t)(n % p.ncols_pad) * kDim + 32 * c + 16 * h);
    }
    TopK<1> top[NB]; int thr[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) { top[j].init(); thr[j] = INT32_MIN; }
    const int st0 = split * p.stages_per_split, st1 = min(st0 + p.stages_per_split, p.nstages);
    const int sw = ((lane & 31) >> 1) & 7;
    int aoff[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) aoff[c] = (lane & 31) * kDim + 16 * ((2 * c + h) ^ sw);
    const int xoff = kStageRowBytes + h * 64;
    if (st0 < st1) issue_stage(p, st0, smem, wave, lane);

    v16i acc[NB];
    int ptile = -1; const char* pbuf = smem;     // tile whose last-chain epilogue is pending
    for (int st = st0; st < st1; ++st) {
        char* buf = smem + ((st - st0) & 1) * kStageBytes;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (st + 1 < st1) issue_stage(p, st + 1, smem + ((st + 1 - st0) & 1) * kStageBytes, wave, lane);
        v4i af[4]; v16i ci;
#pragma unroll
        for (int c = 0; c < 4; ++c) af[c] = *(const v4i*)(buf + aoff[c]);
        ci = lds_read16(buf + xoff);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            v4i naf[4]; v16i nci;
            if (tt < 3) {
#pragma unroll
                for (int c = 0; c < 4; ++c) naf[c] = *(const v4i*)(buf + (tt + 1) * 4096 + aoff[c]);
                nci = lds_read16(buf + xoff + (tt + 1) * 256);
            }
            const int tile = st * 4 + tt;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                // chain for (tile, j)
                v16i a = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[0], bf[j][0], ci, 0, 0, 0);
#pragma unroll
                for (int c = 1; c < 4; ++c) a = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[c], bf[j][c], a, 0, 0, 0);
                // epilogue of the previous chain in this chain's shadow
                const int pj = (j + NB - 1) % NB;
                const int pt = (j == 0) ? ptile : tile;
                const char* pb = (j == 0) ? pbuf : (const char*)buf;
                const int ptt = pt & 3;
                if (pt >= 0) {
                    const int tmax = max16(acc[pj]);
                    if (__builtin_amdgcn_ballot_w64(tmax >= thr[pj]) != 0ull) {
                        const v16i low = lds_read16(pb + xoff + ptt * 256 + 128);
                        top[pj].update(acc[pj], low, pt);
                        thr[pj] = top[pj].own_threshold();
                    }
                }
                acc[j] = a;
            }
            ptile = tile; pbuf = buf;
            if (tt < 3) {
#pragma unroll
                for (int c = 0; c < 4; ++c) af[c] = naf[c];
                ci = nci;
            }
        }
        // the pending epilogue reads `low` of this buffer: finish it before the buffer can be
        // overwritten (next-next stage), i.e. before the next barrier
        {
            const int pj = NB - 1;
            const int tmax = max16(acc[pj]);
            if (__builtin_amdgcn_ballot_w64(tmax >= thr[pj]) != 0ull) {
                const v16i low = lds_read16(pbuf + xoff + (ptile & 3) * 256 + 128);
                top[pj].update(acc[pj], low, ptile);
                thr[pj] = top[pj].own_threshold();
            }
            ptile = -1;
        }
    }
    int s = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) s += top[j].key[0] + top[j].tile[0];
    if (s == 0x7fffffff) p.out[blockIdx.x * 256 + tid] = s;
}

template __global__ void k_base<2, 4, 0>(P);
template __global__ void k_base<2, 4, 1>(P);
template __global__ void k_base<2, 4, 2>(P);
template __global__ void k_base<2, 4, 3>(P);
template __global__ void k_base<2, 4, 7>(P);
template __global__ void k_base<4, 2, 0>(P);
template __global__ void k_base<4, 2, 3>(P);
template __global__ void k_base<4, 2, 7>(P);

template <typename F>
static void bench(const char* name, F launch, int grid, double pairs)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) launch(grid);
    std::vector<float> ts;
    for (int i = 0; i < 7; ++i) {
        hipEventRecord(e0); launch(grid); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    hipError_t err = hipGetLastError();
    printf("%-34s grid %5d  min %.3f  med %.3f ms  -> %.2e pairs/s  (%.1f%% of 1.97e13) %s\n", name, grid, ts[0], ts[3],
           pairs / ts[0] * 1e3, 100.0 * pairs / ts[0] * 1e3 / 1.97e13, err == hipSuccess ? "" : hipGetErrorString(err));
}

int main(int argc, char** argv)
{
    const int n = 100096;                         // multiple of 128
    std::vector<int8_t> rows((size_t)n * 128);
    srand(1);
    for (auto& v : rows) v = (int8_t)(rand() % 61 - 30);
    std::vector<int32_t> aux((size_t)n / 32 * 64);
    for (size_t t = 0; t < (size_t)n / 32; ++t)
        for (int i = 0; i < 32; ++i) { aux[t * 64 + i] = -(rand() % 100000); aux[t * 64 + 32 + i] = ((rand() & 1) << 4) | (15 - (i & 15)); }
    int8_t *d_rows, *d_cols; int32_t* d_aux; int* d_out;
    hipMalloc(&d_rows, rows.size()); hipMalloc(&d_cols, rows.size()); hipMalloc(&d_aux, aux.size() * 4); hipMalloc(&d_out, 1 << 26);
    hipMemcpy(d_rows, rows.data(), rows.size(), hipMemcpyHostToDevice);
    for (auto& v : rows) v = (int8_t)(rand() % 61 - 30);
    hipMemcpy(d_cols, rows.data(), rows.size(), hipMemcpyHostToDevice);
    hipMemcpy(d_aux, aux.data(), aux.size() * 4, hipMemcpyHostToDevice);
    const double pairs = (double)n * n;
    auto mk = [&](int nb, int nsplit) {
        P p; p.col_rows = d_cols; p.red_rows = d_rows; p.red_aux = d_aux; p.ncols_pad = n; p.nstages = n / 128;
        p.nsplit = nsplit; p.stages_per_split = (p.nstages + nsplit - 1) / nsplit; p.nsplit = (p.nstages + p.stages_per_split - 1) / p.stages_per_split;
        p.ncols_alloc = ((n + 128 * nb - 1) / (128 * nb)) * 128 * nb; p.out = d_out; return p;
    };
#define RUN(NAME, KERNEL, NBV, NSPLIT) { P p = mk(NBV, NSPLIT); int grid = (p.ncols_alloc / (128 * NBV)) * p.nsplit; \
        bench(NAME, [&](int g) { hipLaunchKernelGGL(KERNEL, dim3(g), dim3(256), 0, 0, p); }, grid, pairs); }
    RUN("base NB2 wps4 split16", (k_base<2, 4, 0>), 2, 16);
    RUN("base NB2 wps4 noepi", (k_base<2, 4, 1>), 2, 16);
    RUN("base NB2 wps4 nolds", (k_base<2, 4, 2>), 2, 16);
    RUN("base NB2 wps4 noepi nolds", (k_base<2, 4, 3>), 2, 16);
    RUN("base NB2 wps4 mfma only", (k_base<2, 4, 7>), 2, 16);
    RUN("base NB4 wps2 split16", (k_base<4, 2, 0>), 4, 16);
    RUN("base NB4 wps2 noepi nolds", (k_base<4, 2, 3>), 4, 16);
    RUN("base NB4 wps2 mfma only", (k_base<4, 2, 7>), 4, 16);
    RUN("pipe NB2 wps3 split16", (k_pipe<2, 3>), 2, 16);
    RUN("pipe NB2 wps2 split16", (k_pipe<2, 2>), 2, 16);
    RUN("pipe NB4 wps2 split16", (k_pipe<4, 2>), 4, 16);
    RUN("pipe NB2 wps3 split8", (k_pipe<2, 3>), 2, 8);
    RUN("pipe NB4 wps2 split8", (k_pipe<4, 2>), 4, 8);
    RUN("pipe NB4 wps2 split24", (k_pipe<4, 2>), 4, 24);
    return 0;
}
