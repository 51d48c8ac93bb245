// Build-up micro-benchmark: start from the free "dependent max-tree epilogue" loop and add
// the real kernel's features one by one (branch per tile, LDS operand reads, barrier,
// LDS-DMA staging) to see which one costs MFMA throughput.  NB=2-like: 8 MFMAs per tile.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ int max16(const v16i& a)
{
    int m0 = max(max(a[0], a[1]), a[2]), m1 = max(max(a[3], a[4]), a[5]), m2 = max(max(a[6], a[7]), a[8]);
    int m3 = max(max(a[9], a[10]), a[11]), m4 = max(max(a[12], a[13]), a[14]);
    return max(max(max(m0, m1), m2), max(max(m3, m4), a[15]));
}

// F bits: 1 branch per tile-j, 2 LDS operand reads per tile, 4 barrier per 4 tiles, 8 glds staging per 4 tiles
template <int F, int WPS, int NW>
__global__ __launch_bounds__(NW * 64, (WPS * NW + 3) / 4) void k(const int* in, int* out, int iters, int thr0)
{
    __shared__ __attribute__((aligned(16))) char smem[2 * 17408];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    for (int i = tid; i < 2 * 17408 / 4; i += NW * 64) ((int*)smem)[i] = in[i];
    __syncthreads();
    v4i bf[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) bf[j][c] = *(const v4i*)(in + 8192 + (tid * 8 + j * 4 + c) * 4);
    v4i af[4]; v16i ci;
#pragma unroll
    for (int c = 0; c < 4; ++c) af[c] = *(const v4i*)(in + 20000 + (tid * 4 + c) * 4);
#pragma unroll
    for (int r = 0; r < 16; ++r) ci[r] = in[r];
    const int sw = ((lane & 31) >> 1) & 7;
    int aoff[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) aoff[c] = (lane & 31) * 128 + 16 * ((2 * c + h) ^ sw);
    const int xoff = 16384 + h * 64;
    int thr[2] = {thr0, thr0}, m = 0, cnt = 0;
    for (int it = 0; it < iters; ++it) {
        char* buf = smem + (it & 1) * 17408;
        if (F & 4) {
            if (F & 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (F & 8) {
            char* nb = smem + ((it + 1) & 1) * 17408;
            const char* src = (const char*)in + ((it & 63) * 17408);
#pragma unroll
            for (int i = 0; i < 16 / NW; ++i) {
                const int g = wave * (16 / NW) + i;
                if (F & 16) {
                    *(v4i*)(nb + g * 1024 + lane * 16) = *(const v4i*)(src + g * 1024 + lane * 16);
                } else {
                    __builtin_amdgcn_global_load_lds(GLB_PTR(src + g * 1024 + lane * 16), LDS_PTR(nb + g * 1024), 16, 0, 0);
                }
            }
            if (wave == 0) __builtin_amdgcn_global_load_lds(GLB_PTR(src + 16384 + lane * 16), LDS_PTR(nb + 16384), 16, 0, 0);
        }
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            if (F & 2) {
#pragma unroll
                for (int c = 0; c < 4; ++c) af[c] = *(const v4i*)(buf + tt * 4096 + aoff[c]);
                const v4i* ax = (const v4i*)(buf + xoff + tt * 256);
                const v4i c0 = ax[0], c1 = ax[1], c2 = ax[2], c3 = ax[3];
                ci = v16i{c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3], c2[0], c2[1], c2[2], c2[3], c3[0], c3[1], c3[2], c3[3]};
            } else {
                asm volatile("" : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]));
                asm volatile("" : "+v"(ci));
            }
            v16i acc[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[0], bf[j][0], ci, 0, 0, 0);
#pragma unroll
            for (int c = 1; c < 4; ++c)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[c], bf[j][c], acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int tmax = max16(acc[j]);
                if (F & 1) {
                    if (__builtin_amdgcn_ballot_w64(tmax >= thr[j]) != 0ull) {   // never taken (thr huge)
                        cnt += acc[j][3] ^ acc[j][9];
                        thr[j] = max(thr[j], tmax + 1);
                    }
                } else {
                    m = max(m, tmax);
                }
            }
        }
    }
    if (m + cnt == 0x12345678) out[tid] = m;
}

template <int F, int WPS, int NW = 4>
void run(const char* name, int* in, int* out)
{
    const int iters = 500, grid = 256 * WPS;   // WPS blocks per CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<F, WPS, NW>), dim3(grid * 4 / NW), dim3(NW * 64), 0, 0, in, out, iters, 0x7fffffff);
    std::vector<float> ts;
    for (int i = 0; i < 5; ++i) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<F, WPS, NW>), dim3(grid * 4 / NW), dim3(NW * 64), 0, 0, in, out, iters, 0x7fffffff);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    const double mfma = (double)grid * 4 * iters * 32;
    const double cyc = 32.0 * mfma / 1024.0;
    printf("%-44s %.3f ms  -> %.1f%% of nominal 2.4 GHz MFMA rate\n", name, ts[0], 100.0 * cyc / (ts[0] * 1e-3) / 2.4e9);
}

int main()
{
    int *in, *out;
    hipMalloc(&in, 8 << 20); hipMalloc(&out, 1 << 20);
    std::vector<int> h(2 << 20);
    for (auto& x : h) x = rand() & 0x1f1f1f1f;
    hipMemcpy(in, h.data(), 8 << 20, hipMemcpyHostToDevice);
    run<0, 4>("max-tree only                      wps4", in, out);
    run<1, 4>("+branch                            wps4", in, out);
    run<2, 4>("+LDS operand reads                 wps4", in, out);
    run<3, 4>("+branch +LDS reads                 wps4", in, out);
    run<7, 4>("+branch +LDS reads +barrier        wps4", in, out);
    run<15, 4>("+branch +LDS +barrier +glds (full) wps4", in, out);
    run<14, 4>("no branch, LDS +barrier +glds      wps4", in, out);
    run<15, 3>("full                               wps3", in, out);
    run<15, 2>("full                               wps2", in, out);
    run<14, 2>("no branch, LDS +barrier +glds      wps2", in, out);
    run<15, 4, 8>("full, 8 waves/block (half staging)  wps4", in, out);
    run<15, 2, 8>("full, 8 waves/block (half staging)  wps2", in, out);
    run<15, 4, 16>("full, 16 waves/block (1/4 staging)  wps4", in, out);
    run<31, 4, 4>("full, register staging             wps4", in, out);
    run<31, 4, 8>("full, register staging 8 waves     wps4", in, out);
    return 0;
}
