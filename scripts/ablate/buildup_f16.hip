// Build-up of K8's filter loop: register-only MFMA -> + LDS fragment reads -> + reduce epilogue -> + barrier.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float fmax3(float a, float b, float c) { return __builtin_elementwise_maximum(__builtin_elementwise_maximum(a, b), c); }

// LDS: 1 = frags from LDS (read right before use), 2 = prefetched one tile ahead; EPI: reduce epilogue per 32 rows;
// BAR: barrier every 8 tiles
template <int LDS, int EPI, int BAR, int WPS>
__global__ __launch_bounds__(256, WPS) void k(const int* in, float* out, int ntiles)
{
    __shared__ __attribute__((aligned(16))) char smem[2 * 33280];
    const int lane = threadIdx.x & 63, g = lane >> 4, c16 = lane & 15;
    for (int i = threadIdx.x; i < 2 * 33280 / 4; i += 256) ((int*)smem)[i] = in[i & 65535];
    __syncthreads();
    v8h b[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int s = 0; s < 4; ++s) b[j][s] = __builtin_bit_cast(v8h, *(const v4i*)(in + 65536 + (threadIdx.x * 16 + j * 4 + s) * 4));
    int aoff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) aoff[s] = c16 * 256 + 16 * ((4 * s + g) ^ c16);
    float thr[4] = {3e38f, 3e38f, 3e38f, 3e38f};
    v8h f[2][4];
    v4f ci[2];
    float keep = 0;
    auto load = [&](int set, int k) __attribute__((always_inline)) {
        const char* rows = smem + (k & 7) * 4096;
        ci[set] = *(const v4f*)(smem + 32768 + 16 * g + (k & 7) * 64);
#pragma unroll
        for (int s = 0; s < 4; ++s) f[set][s] = __builtin_bit_cast(v8h, *(const v4i*)(rows + aoff[s]));
    };
    if (LDS == 0) { load(0, 0); load(1, 1); }
    if (LDS == 2) load(0, 0);
    v4f acc[2][4];
    for (int k0 = 0; k0 < ntiles; k0 += 8) {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int k = k0 + kk, t = kk & 1;
            if (LDS == 1) load(t, kk);
            if (LDS == 2) { load((kk + 1) & 1, kk + 1); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[t][0], b[j][0], ci[t], 0, 0, 0);
#pragma unroll
            for (int s = 1; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[t][s], b[j][s], acc[t][j], 0, 0, 0);
            if (t == 1) {
                if (EPI) {
                    bool any = false;
                    float tm[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float m0 = fmax3(acc[0][j][0], acc[0][j][1], acc[0][j][2]);
                        const float m1 = fmax3(acc[0][j][3], acc[1][j][0], acc[1][j][1]);
                        tm[j] = fmax3(fmax3(acc[1][j][2], acc[1][j][3], m0), m1, m1);
                        any |= tm[j] >= thr[j];
                    }
                    if (__builtin_amdgcn_ballot_w64(any) != 0ull) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) { keep += tm[j]; thr[j] = tm[j] + 1.f; out[k & 1023] = keep; }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) keep += acc[0][j][0] + acc[1][j][3];
                }
            }
        }
        if (BAR) __syncthreads();
    }
    if (keep == 1234.5f) out[threadIdx.x] = keep;
}

template <int LDS, int EPI, int BAR, int WPS>
void run(const char* name, int* in, float* out)
{
    const int ntiles = 16000, grid = 256 * WPS;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<LDS, EPI, BAR, WPS>), dim3(grid), dim3(256), 0, 0, in, out, ntiles);
    std::vector<float> ts;
    for (int i = 0; i < 5; ++i) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<LDS, EPI, BAR, WPS>), dim3(grid), dim3(256), 0, 0, in, out, ntiles);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    const double macs = (double)grid * 4 * ntiles * 16 * 8192.0;
    printf("%-46s wps %d: %.3f ms  %.0f%% of 2500 TFLOP/s\n", name, WPS, ts[2], 2 * macs / (ts[2] * 1e-3) / 2.5e15 * 100);
}

int main()
{
    int* in; float* out;
    hipMalloc(&in, 1 << 21);
    {
        std::vector<unsigned short> h(1 << 20);
        unsigned x = 12345;
        for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned short)(((x >> 16) & 0x83ff) | 0x3c00); }
        hipMemcpy(in, h.data(), 1 << 21, hipMemcpyHostToDevice);
    }
    hipMalloc(&out, 1 << 16);
    run<0, 0, 0, 2>("registers only", in, out);
    run<1, 0, 0, 2>("+ LDS fragments, read at use", in, out);
    run<2, 0, 0, 2>("+ LDS fragments, one tile ahead", in, out);
    run<2, 1, 0, 2>("+ reduce epilogue", in, out);
    run<2, 1, 1, 2>("+ barrier per 8 tiles", in, out);
    run<1, 1, 1, 2>("same, fragments read at use", in, out);
    run<0, 1, 0, 2>("registers + reduce epilogue", in, out);
    run<1, 1, 1, 3>("fragments read at use", in, out);
    run<2, 1, 1, 3>("fragments one tile ahead", in, out);
    return 0;
}
