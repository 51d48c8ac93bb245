// Micro-benchmark: how much independent VALU work hides under v_mfma_i32_32x32x32_i8?
// Per iteration each wave issues 16 MFMAs (4 independent accumulator chains x 4) and NV
// independent v_max3_i32; WPS waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int NV, int WPS, bool DEP>
__global__ __launch_bounds__(256, WPS) void k(const int* in, int* out, int iters)
{
    v4i a = *(const v4i*)(in + threadIdx.x * 4);
    v4i b = *(const v4i*)(in + 1024 + threadIdx.x * 4);
    v16i acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = in[j * 16 + r];
    int v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = in[100 + i + threadIdx.x];
    int m = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[j], 0, 0, 0);
                if (!DEP) {
#pragma unroll
                    for (int q = 0; q < NV / 16; ++q) {
                        const int i = (c * 4 + j + q) & 7;
                        v[i] = max(max(v[i], v[(i + 1) & 7]), v[(i + 3) & 7] + it);
                    }
                }
            }
        if (DEP) {
            // dependent epilogue: max tree over each accumulator (as the real kernel does)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int t0 = max(max(acc[j][0], acc[j][1]), acc[j][2]);
                int t1 = max(max(acc[j][3], acc[j][4]), acc[j][5]);
                int t2 = max(max(acc[j][6], acc[j][7]), acc[j][8]);
                int t3 = max(max(acc[j][9], acc[j][10]), acc[j][11]);
                int t4 = max(max(acc[j][12], acc[j][13]), acc[j][14]);
                m = max(m, max(max(max(t0, t1), t2), max(max(t3, t4), acc[j][15])));
            }
        }
    }
    int s = m;
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][7];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 0x12345678) out[threadIdx.x] = s;
}

template <int NV, int WPS, bool DEP>
void run(const char* name, int* in, int* out)
{
    const int iters = 2000, grid = 256 * WPS;     // WPS blocks of 4 waves per CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<NV, WPS, DEP>), dim3(grid), dim3(256), 0, 0, in, out, iters);
    std::vector<float> ts;
    for (int i = 0; i < 5; ++i) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NV, WPS, DEP>), dim3(grid), dim3(256), 0, 0, in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    const double mfma = (double)grid * 4 * iters * 16;
    const double cyc = 32.0 * mfma / 1024.0;      // MFMA pipe cycles per SIMD
    printf("%-28s %.3f ms  %.1f ns/iter/wave-slot  mfma-bound clock-equivalent %.2f GHz (%.1f%% of 2.4)\n", name, ts[0],
           ts[0] * 1e6 / iters / WPS, cyc / (ts[0] * 1e-3) / 1e9, 100.0 * cyc / (ts[0] * 1e-3) / 2.4e9);
}

int main()
{
    int *in, *out;
    hipMalloc(&in, 1 << 20); hipMalloc(&out, 1 << 20);
    std::vector<int> h(1 << 18);
    for (auto& x : h) x = rand();
    hipMemcpy(in, h.data(), 1 << 20, hipMemcpyHostToDevice);
    run<0, 1, false>("NV=0  wps1", in, out);
    run<0, 2, false>("NV=0  wps2", in, out);
    run<0, 4, false>("NV=0  wps4", in, out);
    run<16, 4, false>("NV=16 wps4 indep", in, out);
    run<32, 4, false>("NV=32 wps4 indep", in, out);
    run<48, 4, false>("NV=48 wps4 indep", in, out);
    run<64, 4, false>("NV=64 wps4 indep", in, out);
    run<96, 4, false>("NV=96 wps4 indep", in, out);
    run<32, 1, false>("NV=32 wps1 indep", in, out);
    run<64, 1, false>("NV=64 wps1 indep", in, out);
    run<32, 2, false>("NV=32 wps2 indep", in, out);
    run<0, 4, true>("dep epilogue wps4 (36 VALU)", in, out);
    run<0, 2, true>("dep epilogue wps2 (36 VALU)", in, out);
    run<0, 1, true>("dep epilogue wps1 (36 VALU)", in, out);
    return 0;
}
