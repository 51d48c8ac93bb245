// Ablations of the 16x16x64 row-reduce structure (not part of the product).  MODE bits:
//  1 = exact path never taken, 2 = no epilogue at all, 4 = operands not re-read from LDS,
//  8 = no staging / barriers.
#include "tile_ops.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <type_traits>
using namespace fm;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
struct P { const int8_t* col_rows; const int8_t* red_rows; const int32_t* red_aux; int ncols_pad, nstages, nsplit, nchunks, stages_per_split; int* out; };

template <int NW>
__device__ __forceinline__ void issue_stage(const P& p, int stage, char* buf, int wave, int lane)
{
    const int8_t* src_rows = p.red_rows + (size_t)stage * kStageRowBytes;
    const int slot = lane & 7;
#pragma unroll
    for (int i = 0; i < 16 / NW; ++i) {
        const int g = wave * (16 / NW) + i, row = g * 8 + (lane >> 3);
        __builtin_amdgcn_global_load_lds(GLB_PTR(src_rows + row * kDim + 16 * (slot ^ ((row >> 1) & 7))), LDS_PTR(buf + g * 1024), 16, 0, 0);
    }
    if (wave == NW - 1)
        __builtin_amdgcn_global_load_lds(GLB_PTR(p.red_aux + (size_t)stage * 256 + lane * 4), LDS_PTR(buf + kStageRowBytes), 16, 0, 0);
}

template <int NC, int NW, int WPS, int MODE>
__global__ __launch_bounds__(64 * NW, WPS) void k(P p)
{
    __shared__ __attribute__((aligned(16))) char smem[2 * kStageBytes];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
    const int chunk = blockIdx.x % p.nchunks, split = blockIdx.x / p.nchunks;
    const int cb = chunk * (16 * NC * NW) + wave * (16 * NC);
    v4i bf[NC][2];
#pragma unroll
    for (int j = 0; j < NC; ++j)
#pragma unroll
        for (int c = 0; c < 2; ++c) bf[j][c] = *(const v4i*)(p.col_rows + (size_t)((cb + 16 * j + c16) % p.ncols_pad) * kDim + 64 * c + 16 * g);
    TopK8<1> top[NC]; int thr[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) { top[j].init(); thr[j] = (MODE & 1) ? INT32_MAX : INT32_MIN; }
    const int st0 = split * p.stages_per_split, st1 = min(st0 + p.stages_per_split, p.nstages);
    const int sw = (c16 >> 1) & 7;
    int aoff[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) aoff[c] = c16 * kDim + 16 * ((g + 4 * c) ^ sw);
    const int xoff = kStageRowBytes + 16 * g;
    if (st0 < st1) issue_stage<NW>(p, st0, smem, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    v4i haf0 = *(const v4i*)(smem + aoff[0]), haf1 = *(const v4i*)(smem + aoff[1]), hci = *(const v4i*)(smem + xoff);
    auto stage = [&](auto buf_tag, int st) {
        constexpr int BUF = decltype(buf_tag)::value;
        char* buf = smem + BUF * kStageBytes;
        if (!(MODE & 8)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (st + 1 < st1) issue_stage<NW>(p, st + 1, smem + (BUF ^ 1) * kStageBytes, wave, lane);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v4i acc[2][NC];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                v4i af0, af1, ci;
                if (MODE & 4) {
                    af0 = haf0; af1 = haf1; ci = hci;
#if defined(__HIP_DEVICE_COMPILE__)
                    asm volatile("" : "+v"(af0), "+v"(af1), "+v"(ci));
#endif
                } else {
                    const char* rows = buf + (32 * u + 16 * s) * kDim;
                    af0 = *(const v4i*)(rows + aoff[0]); af1 = *(const v4i*)(rows + aoff[1]);
                    ci = *(const v4i*)(buf + xoff + u * 256 + s * 128);
                }
#pragma unroll
                for (int j = 0; j < NC; ++j) acc[s][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af0, bf[j][0], ci, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NC; ++j) acc[s][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af1, bf[j][1], acc[s][j], 0, 0, 0);
            }
            if (MODE & 2) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
                for (int j = 0; j < NC; ++j) asm volatile("" :: "v"(acc[0][j]), "v"(acc[1][j]));
#endif
            } else {
                int tmax[NC]; bool any = false;
#pragma unroll
                for (int j = 0; j < NC; ++j) {
                    const int m0 = max(max(acc[0][j][0], acc[0][j][1]), acc[0][j][2]);
                    const int m1 = max(max(acc[0][j][3], acc[1][j][0]), acc[1][j][1]);
                    tmax[j] = max(max(max(acc[1][j][2], acc[1][j][3]), m0), m1);
                    any |= tmax[j] >= thr[j];
                }
                if (__builtin_amdgcn_ballot_w64(any) != 0ull) {
                    const v4i low0 = *(const v4i*)(buf + xoff + u * 256 + 64), low1 = *(const v4i*)(buf + xoff + u * 256 + 192);
#pragma unroll
                    for (int j = 0; j < NC; ++j)
                        if (__builtin_amdgcn_ballot_w64(tmax[j] >= thr[j]) != 0ull) {
                            top[j].update(acc[0][j], acc[1][j], low0, low1, st * 4 + u);
                            thr[j] = top[j].own_threshold();
                        }
                }
            }
        }
    };
    for (int st = st0; st < st1; st += 2) {
        stage(std::integral_constant<int, 0>{}, st);
        if (st + 1 < st1) stage(std::integral_constant<int, 1>{}, st + 1);
    }
    int s = 0;
#pragma unroll
    for (int j = 0; j < NC; ++j) s += top[j].key[0] + top[j].unit[0];
    if (s == 0x7fffffff) p.out[blockIdx.x * 64 * NW + tid] = s;
}

template <typename F> static void bench(const char* name, F launch, double pairs)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) launch();
    std::vector<float> ts;
    for (int i = 0; i < 7; ++i) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms); }
    std::sort(ts.begin(), ts.end());
    printf("%-40s min %.3f med %.3f ms -> %.1f%% of nominal int8 peak\n", name, ts[0], ts[3], 100.0 * pairs * 256 / (ts[0] * 1e-3) / 5e15);
}

int main()
{
    const int n = 100096;
    std::vector<int8_t> rows((size_t)n * 128); srand(1);
    for (auto& v : rows) v = (int8_t)(rand() % 61 - 30);
    std::vector<int32_t> aux((size_t)n / 32 * 64);
    for (size_t t = 0; t < (size_t)n / 32; ++t) for (int i = 0; i < 64; ++i) aux[t * 64 + i] = (i & 16) ? (((rand() & 1) << 4) | (15 - (i & 7))) : -(rand() % 100000);
    int8_t *d_rows, *d_cols; int32_t* d_aux; int* d_out;
    hipMalloc(&d_rows, rows.size()); hipMalloc(&d_cols, rows.size()); hipMalloc(&d_aux, aux.size() * 4); hipMalloc(&d_out, 1 << 26);
    hipMemcpy(d_rows, rows.data(), rows.size(), hipMemcpyHostToDevice);
    for (auto& v : rows) v = (int8_t)(rand() % 61 - 30);
    hipMemcpy(d_cols, rows.data(), rows.size(), hipMemcpyHostToDevice);
    hipMemcpy(d_aux, aux.data(), aux.size() * 4, hipMemcpyHostToDevice);
    const double pairs = (double)n * n;
#define RUN(NAME, NC, NW, WPS, MODE, NSPLIT) { P p; p.col_rows = d_cols; p.red_rows = d_rows; p.red_aux = d_aux; p.ncols_pad = n; p.nstages = n / 128; \
        p.stages_per_split = (p.nstages + NSPLIT - 1) / NSPLIT; p.nsplit = (p.nstages + p.stages_per_split - 1) / p.stages_per_split; \
        p.nchunks = (n + 16 * NC * NW - 1) / (16 * NC * NW); p.out = d_out; const int grid = p.nchunks * p.nsplit; \
        bench(NAME, [&]() { hipLaunchKernelGGL((k<NC, NW, WPS, MODE>), dim3(grid), dim3(64 * NW), 0, 0, p); }, pairs); }
    RUN("full (no bounds)        NC4 NW8", 4, 8, 4, 0, 16);
    RUN("exact path never taken  NC4 NW8", 4, 8, 4, 1, 16);
    RUN("no epilogue             NC4 NW8", 4, 8, 4, 2, 16);
    RUN("no LDS operand reads    NC4 NW8", 4, 8, 4, 4, 16);
    RUN("no epi, no LDS reads    NC4 NW8", 4, 8, 4, 6, 16);
    RUN("no staging/barrier      NC4 NW8", 4, 8, 4, 12, 16);
    RUN("MFMA only               NC4 NW8", 4, 8, 4, 14, 16);
    RUN("full (no bounds)        NC4 NW4", 4, 4, 4, 0, 16);
    RUN("exact never             NC4 NW4", 4, 4, 4, 1, 16);
    RUN("MFMA only               NC4 NW4", 4, 4, 4, 14, 16);
    return 0;
}
