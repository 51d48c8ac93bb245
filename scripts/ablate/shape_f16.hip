// MFMA-only rate of the fp16 shapes (register operands), WPS waves/SIMD: what K8's filter can hope for.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));

template <int SHAPE, int WPS>
__global__ __launch_bounds__(256, WPS) void k(const int* in, float* out, int iters)
{
    v8h a[4], b[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        a[c] = __builtin_bit_cast(v8h, *(const v4i*)(in + (threadIdx.x * 4 + c) * 4));
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j][c] = __builtin_bit_cast(v8h, *(const v4i*)(in + 8192 + (threadIdx.x * 16 + j * 4 + c) * 4));
    }
    float s = 0;
    if (SHAPE == 32) {
        v16f acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = (float)in[j * 16 + r];
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[c], b[j][c], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) s += acc[j][0] + acc[j][9];
    } else {
        v4f acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[j][r] = (float)in[j * 4 + r];
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[c], b[j][c], acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][3];
    }
    if (s == 1234.5f) out[threadIdx.x] = s;
}

template <int SHAPE, int WPS>
void run(const char* name, int* in, float* out)
{
    const int iters = 40000, grid = 256 * WPS;
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
    // MACs per wave-iteration: 16x16x32 x 16 = 131072; 32x32x16 x 8 = 131072
    const double macs = (double)grid * 4 * iters * 131072.0;
    printf("%-28s wps %d: %.3f ms  %.1f TFLOP/s (%.0f%% of 2500)\n", name, WPS, ts[2], 2 * macs / (ts[2] * 1e-3) / 1e12, 2 * macs / (ts[2] * 1e-3) / 2.5e15 * 100);
}

int main()
{
    int* in; float* out;
    hipMalloc(&in, 1 << 20);
    {   // random fp16 values in [-2, 2): the operand bits toggle as real data does (DVFS sees it)
        std::vector<unsigned short> h(1 << 19);
        unsigned x = 12345;
        for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(((x >> 16) & 0x83ff) | 0x3c00); }
        hipMemcpy(in, h.data(), 1 << 20, hipMemcpyHostToDevice);
    } hipMalloc(&out, 1 << 16);
    run<16, 1>("v_mfma_f32_16x16x32_f16", in, out);
    run<16, 2>("v_mfma_f32_16x16x32_f16", in, out);
    run<16, 4>("v_mfma_f32_16x16x32_f16", in, out);
    run<32, 1>("v_mfma_f32_32x32x16_f16", in, out);
    run<32, 2>("v_mfma_f32_32x32x16_f16", in, out);
    run<32, 4>("v_mfma_f32_32x32x16_f16", in, out);
    return 0;
}
