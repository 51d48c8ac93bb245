// K5's inner loop in isolation: 4x4 patch per thread, operands from LDS (k-major image), no
// staging.  Variants: V=0 as in the kernel (float4 LDS reads), V=1 operands kept in registers
// (no LDS reads, values made opaque), V=2 LDS reads but VALU reduced to fma only (no sub).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
constexpr int LD = 68;
template <int V, int WPS>
__global__ __launch_bounds__(256, WPS) void k(const float* in, float* out, int iters)
{
    __shared__ __attribute__((aligned(16))) float colimg[128 * LD];
    __shared__ __attribute__((aligned(16))) float redimg[128 * LD];
    const int tid = threadIdx.x, tn = tid & 15, tm = tid >> 4;
    for (int i = tid; i < 128 * LD; i += 256) { colimg[i] = in[i & 4095]; redimg[i] = in[(i * 7) & 4095]; }
    __syncthreads();
    float s[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s[i][j] = 0.f;
    float4 ra = *(const float4*)(colimg + 4 * tn), rb = *(const float4*)(redimg + 4 * tm);
    for (int it = 0; it < iters; ++it) {
#pragma unroll 8
        for (int kk = 0; kk < 128; ++kk) {
            float4 a, b;
            if (V == 1 || V == 4) {
                a = ra; b = rb;
#if defined(__HIP_DEVICE_COMPILE__)
                asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w));
                asm volatile("" : "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w));
#endif
            } else {
                a = *(const float4*)(colimg + kk * LD + 4 * tn);
                b = *(const float4*)(redimg + kk * LD + 4 * tm);
            }
            const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
            if (V == 3 || V == 4) {
                float v[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[i][j] = av[i] - bv[j];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) s[i][j] = __builtin_fmaf(v[i][j], v[i][j], s[i][j]);
                __builtin_amdgcn_sched_barrier(0);
                continue;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (V == 2) { s[i][j] = __builtin_fmaf(av[i], bv[j], s[i][j]); }
                    else { const float v = av[i] - bv[j]; s[i][j] = __builtin_fmaf(v, v, s[i][j]); }
                }
        }
    }
    float t = 0; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) t += s[i][j];
    if (t == 12345.f) out[tid] = t;
}
template <int V, int WPS> void run(const char* name, const float* in, float* out)
{
    const int iters = 60, grid = 256 * WPS;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<V, WPS>), dim3(grid), dim3(256), 0, 0, in, out, iters);
    std::vector<float> ts;
    for (int i = 0; i < 5; ++i) { hipEventRecord(e0); hipLaunchKernelGGL((k<V, WPS>), dim3(grid), dim3(256), 0, 0, in, out, iters); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms); }
    std::sort(ts.begin(), ts.end());
    const double pairdims = (double)grid * 256 * iters * 128 * 16;
    printf("%-28s wps%d: %.3f ms -> %.2e pair-dims/s = %.1f%% of the 2-VALU bound (3.93e13)\n", name, WPS, ts[0], pairdims / (ts[0] * 1e-3), 100 * pairdims / (ts[0] * 1e-3) / 3.93e13);
}
int main()
{
    float *in, *out; hipMalloc(&in, 1 << 16); hipMalloc(&out, 1 << 16);
    std::vector<float> h(1 << 14); for (auto& x : h) x = (float)(rand() % 255) + 0.25f;
    hipMemcpy(in, h.data(), 1 << 16, hipMemcpyHostToDevice);
    run<0, 2>("kernel loop (LDS float4)", in, out);
    run<1, 2>("operands in registers", in, out);
    run<2, 2>("LDS float4, fma only", in, out);
    run<3, 2>("LDS float4, subs|fmacs batched", in, out);
    run<4, 2>("registers,  subs|fmacs batched", in, out);
    run<3, 1>("LDS float4, subs|fmacs batched", in, out);
    run<0, 1>("kernel loop (LDS float4)", in, out);
    run<1, 1>("operands in registers", in, out);
    return 0;
}
