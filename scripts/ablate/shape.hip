// MFMA-only rate of the two int8 shapes (same MACs per wave-iteration), WPS waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int SHAPE, int WPS>
__global__ __launch_bounds__(256, WPS) void k(const int* in, int* out, int iters)
{
    v4i a[4], b[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { a[c] = *(const v4i*)(in + (threadIdx.x * 4 + c) * 4); b[c] = *(const v4i*)(in + 8192 + (threadIdx.x * 4 + c) * 4); }
    int s = 0;
    if (SHAPE == 32) {
        v16i acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = in[j * 16 + r];
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[c], b[(c + j) & 3], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][9];
    } else {
        v4i acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[j][r] = in[j * 4 + r];
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[(c + j) & 3], b[(c * 2 + j) & 3], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[j][0] + acc[j][3];
    }
    if (s == 0x12345678) out[threadIdx.x] = s;
}

template <int SHAPE, int WPS>
void run(const char* name, int* in, int* out)
{
    const int iters = 2000, grid = 256 * WPS;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<SHAPE, WPS>), dim3(grid), dim3(256), 0, 0, in, out, iters);
    std::vector<float> ts;
    for (int i = 0; i < 5; ++i) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<SHAPE, WPS>), dim3(grid), dim3(256), 0, 0, in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    // MACs per wave-iteration: 32x32x32 x16 = 524288; 16x16x64 x32 = 524288
    const double macs = (double)grid * 4 * iters * 524288.0;
    printf("%-24s %.3f ms -> %.2f POPS (%.1f%% of 5.0 nominal)\n", name, ts[0], 2 * macs / (ts[0] * 1e-3) / 1e15, 100 * 2 * macs / (ts[0] * 1e-3) / 5.0e15);
}

int main()
{
    int *in, *out;
    hipMalloc(&in, 1 << 20); hipMalloc(&out, 1 << 20);
    std::vector<int> h(1 << 18);
    for (auto& x : h) x = rand();
    hipMemcpy(in, h.data(), 1 << 20, hipMemcpyHostToDevice);
    run<32, 1>("32x32x32 wps1", in, out); run<16, 1>("16x16x64 wps1", in, out);
    run<32, 2>("32x32x32 wps2", in, out); run<16, 2>("16x16x64 wps2", in, out);
    run<32, 4>("32x32x32 wps4", in, out); run<16, 4>("16x16x64 wps4", in, out);
    return 0;
}
