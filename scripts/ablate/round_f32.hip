// Micro-benchmark of x1_round_f32 (round_body_f32.h) on one workgroup: per-phase 100 MHz ticks.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I fast-match_amd/csrc scripts/ablate/round_f32.hip -o scripts/ablate/round_f32
#include "round_body_f32.h"
#include <cstdio>
#include <vector>
#include <cstdlib>
#include <cmath>
using namespace fm;

constexpr int kLds = kRF_StageBytes + 4096 * 8 + 4096 * 8 + 128 * 8 + 64 + 4096 * 4;

__global__ __launch_bounds__(256) void k(RoundF32 R, const int* q_rows_g, int nq, int nt, int reps, long long* out)
{
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    char* smem = dyn;
    unsigned long long* qbest = (unsigned long long*)(dyn + kRF_StageBytes);
    unsigned* clist = (unsigned*)(qbest + 4096);
    unsigned long long* tbest = (unsigned long long*)(clist + 8192);
    int* sh = (int*)(tbest + 128);
    int* q_rows = sh + 16;
    for (int i = threadIdx.x; i < nq; i += 256) q_rows[i] = q_rows_g[i];
    long long pt[12] = {0};
    long long ts = wall_clock64();
    long long t0 = ts;
    for (int r = 0; r < reps; ++r) {
        for (int i = threadIdx.x; i < nq; i += 256) qbest[i] = ~0ull;
        __syncthreads();
        x1_round_f32(RoundF32G(R), q_rows, nq, 0, nt, smem, qbest, clist, 8192, tbest, sh, pt, &ts);
        __syncthreads();
    }
    if (threadIdx.x == 0) { for (int i = 0; i < 12; ++i) out[i] = pt[i]; out[12] = wall_clock64() - t0; out[13] = sh[0]; }
}

int main(int argc, char** argv)
{
    const int n = 12544, nq = argc > 1 ? atoi(argv[1]) : 393, nt = argc > 2 ? atoi(argv[2]) : 125, reps = 200;
    std::vector<float> rows((size_t)n * 128);
    srand(3);
    for (auto& v : rows) v = (float)(rand() % 120) + 0.25f;
    // fp16 plane scaled by 2^6 (max ~120 -> 7680 in [2^12, 2^13)), norms, aux
    std::vector<_Float16> h((size_t)n * 128);
    std::vector<float> normf(n), auxf(n);
    for (int i = 0; i < n; ++i) {
        double ss = 0;
        for (int k = 0; k < 128; ++k) { float x = rows[(size_t)i * 128 + k] * 64.f; h[(size_t)i * 128 + k] = (_Float16)x; ss += (double)x * x; }
        normf[i] = (float)ss; auxf[i] = -0.5f * (float)ss;
    }
    float nm_max = 0; for (float v : normf) nm_max = fmaxf(nm_max, v);
    char *d_h; float *d_f, *d_n, *d_a; int* d_q; long long* d_out;
    hipMalloc(&d_h, h.size() * 2); hipMalloc(&d_f, rows.size() * 4); hipMalloc(&d_n, n * 4); hipMalloc(&d_a, n * 4);
    hipMalloc(&d_q, 4096 * 4); hipMalloc(&d_out, 256);
    hipMemcpy(d_h, h.data(), h.size() * 2, hipMemcpyHostToDevice); hipMemcpy(d_f, rows.data(), rows.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_n, normf.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(d_a, auxf.data(), n * 4, hipMemcpyHostToDevice);
    std::vector<int> q(4096); for (auto& v : q) v = rand() % n;
    hipMemcpy(d_q, q.data(), 4096 * 4, hipMemcpyHostToDevice);
    RoundF32 R; R.q_rowsh = d_h; R.q_auxf = d_a; R.q_rowsf = d_f; R.t_rowsh = d_h + (size_t)6000 * 256; R.t_normf = d_n + 6000; R.t_rowsf = d_f + (size_t)6000 * 128;
    const float eps = 1.1f / 1024.f; R.eps_c = eps; R.eps_nm = eps * nm_max; R.aux_mul = 1.f;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    for (int it = 0; it < 2; ++it) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), kLds, 0, R, d_q, nq, nt, reps, d_out);
        long long o[16]; hipMemcpy(o, d_out, sizeof(o), hipMemcpyDeviceToHost);
        printf("nq %d nt %d: total %.2f us/round | bfrag %.2f gather %.2f sweep0 %.2f sweep1 %.2f rest %.2f exact+merge %.2f | last ncand %lld\n", nq, nt,
               o[12] * 0.01 / reps, o[8] * 0.01 / reps, o[9] * 0.01 / reps, o[6] * 0.01 / reps, o[7] * 0.01 / reps, o[10] * 0.01 / reps, o[11] * 0.01 / reps, o[13]);
    }
    return 0;
}
