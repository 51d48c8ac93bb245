// fp32 VALU issue-rate micro-benchmark: independent v_sub + v_fmac chains, WPS waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int WPS>
__global__ __launch_bounds__(256, WPS) void k(const float* in, float* out, int iters)
{
    float a[4], b[4], s[16];
    for (int i = 0; i < 4; ++i) { a[i] = in[threadIdx.x + i]; b[i] = in[threadIdx.x + 7 + i]; }
    for (int i = 0; i < 16; ++i) s[i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float v = a[i] - b[j]; s[i * 4 + j] = __builtin_fmaf(v, v, s[i * 4 + j]); }
            a[r & 3] += 1.0f;          // keep the subtractions from being hoisted
        }
    }
    float t = 0; for (int i = 0; i < 16; ++i) t += s[i];
    if (t == 12345.f) out[threadIdx.x] = t;
}
template <int WPS> void run(const float* in, float* out)
{
    const int iters = 4000, grid = 256 * WPS;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<WPS>), dim3(grid), dim3(256), 0, 0, in, out, iters);
    std::vector<float> ts;
    for (int i = 0; i < 5; ++i) { hipEventRecord(e0); hipLaunchKernelGGL((k<WPS>), dim3(grid), dim3(256), 0, 0, in, out, iters); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms); }
    std::sort(ts.begin(), ts.end());
    const double lane_ops = (double)grid * 256 * iters * 8 * 32;      // v_sub + v_fmac per pair
    printf("wps %d: %.3f ms -> %.2e lane-instr/s = %.1f%% of 7.86e13 (2.4 GHz nominal)\n", WPS, ts[0], lane_ops / (ts[0] * 1e-3), 100 * lane_ops / (ts[0] * 1e-3) / 7.86e13);
}
int main()
{
    float *in, *out; hipMalloc(&in, 1 << 16); hipMalloc(&out, 1 << 16); hipMemset(in, 0, 1 << 16);
    run<1>(in, out); run<2>(in, out); run<4>(in, out); run<8>(in, out);
    return 0;
}
