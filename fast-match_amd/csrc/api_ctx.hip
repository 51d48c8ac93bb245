// C-ABI of libfastmatch_hip.so (include/fastmatch_hip.h), part 1: the context (streams, workspaces, options,
// statistics, fm_mark / fm_wait), page-locked host memory and device-to-host staging, and the descriptor banks
// with K6, the bank upload kernel (bytes XOR 0x80, row norms, accumulator-order aux words).  The other
// entry points: api_match.hip (2-NN, cross-check, batches, rounds), api_expand.hip (K7), comm.hip (gather).
#include "ctx_internal.h"

using namespace fm;

// ---------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------
static std::mutex g_err_mu;
static std::string g_err;   // last context-less error

int fm::fail(fm_ctx* ctx, int code, const std::string& msg)
{
    if (ctx) ctx->err = msg;
    else { std::lock_guard<std::mutex> lk(g_err_mu); g_err = msg; }
    return code;
}

int fm::ws_ensure(fm_ctx* ctx, void** p, size_t* cap, size_t need)
{
    if (need <= *cap && *p) return FM_OK;
    if (*p) { HIP_TRY(ctx, hipFree(*p)); *p = nullptr; *cap = 0; }
    size_t sz = need + need / 4 + 4096;
    HIP_TRY(ctx, hipMalloc(p, sz));
    *cap = sz;
    return FM_OK;
}
// ---------------------------------------------------------------------------------------
// K6: bank preparation
// ---------------------------------------------------------------------------------------
// One 256-thread block per 32-row tile; thread (r = tid>>3, c = tid&7) owns the 16 bytes
// [16c, 16c+16) of tile row r.  SRC_F32: source rows are float32; values are converted to
// uint8 and nonint[0] is raised if any value is not an integer in [0,255]; nonint[1] = max over the
// rows of the squared norm of the uint8 row (Bank::usq_max).
#ifdef FM_ENC_AB
// Measurement builds only (scripts/gpu_k1_enc_ab.sh): the shift of the int8 encoding, 128 = the product's
// u ^ 0x80.  L2 is shift invariant as long as BOTH banks of a call use one shift and no byte leaves int8: shift 0 is exact
// for rows whose bytes are <= 127.  The question it answers: does the operand ENCODING move the clock the chip holds
// (MI355X_MICROARCH.md, DVFS give-back (1): zero-heavy operands clock 2.30 vs 1.9 GHz) -- SIFT's many near-zero bytes are
// -128 .. -100 under the product's encoding, 0 .. 27 under shift 0.
__device__ int g_enc_shift = 128;
extern "C" int fm_debug_enc_shift(int s) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_enc_shift), &s, sizeof(int)); }
#define FM_ENC_SHIFT g_enc_shift
#else
#define FM_ENC_SHIFT 128
#endif
template <bool SRC_F32>
__global__ __launch_bounds__(256)
void bank_prep_kernel(const void* __restrict__ src, int64_t n, int dim,
                      int8_t* __restrict__ rows8, int32_t* __restrict__ norm,
                      int32_t* __restrict__ aux, int* __restrict__ nonint, int64_t ntiles,
                      const int32_t* __restrict__ map = nullptr)
{
    // map (fm_bank_create_*_gather): bank row i is source row map[i] -- a Grid_Cache over pre-extracted features packs
    // a keypoint into up to four cells; its descriptor crosses PCIe once and is copied here
    const int tid = threadIdx.x;
    const int r = tid >> 3, c = tid & 7;
    // A workgroup walks tiles blockIdx.x, + gridDim.x, ...: a refill (fm_bank_refill_u8_async) runs beside the distance
    // kernels with a small grid of long-lived workgroups -- thousands of one-tile workgroups each took a CU slot a
    // distance-kernel workgroup was waiting for (r04: 0.27 ms per 100k-row bank beside K1, 0.18 ms per image pair lost).
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row = tile * kTileRows + r;
    unsigned w[4] = {0, 0, 0, 0};
    int sumsq = 0, usq = 0;
    bool bad = false;
    if (row < n) {
        const int64_t srow = map ? (int64_t)map[row] : row;
        // the thread's 16 values; full-width rows (dim = 128: SIFT) come in as 16-byte loads -- byte / float loads
        // one at a time ran the upload kernel at ~80 GB/s
        int uv[16];
        bool have[16];
        if (dim == kDim) {
            if constexpr (SRC_F32) {
                const float4* sp = (const float4*)((const float*)src + srow * kDim + 16 * c);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = sp[q];
                    const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const float fr = rintf(f[b]);
                        const bool ok = (f[b] == fr) && f[b] >= 0.f && f[b] <= 255.f;
                        bad |= !ok;
                        uv[4 * q + b] = ok ? (int)fr : 128;
                        have[4 * q + b] = true;
                    }
                }
            } else {
                const uint4 v = *(const uint4*)((const uint8_t*)src + srow * kDim + 16 * c);
                const unsigned ww[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int b = 0; b < 4; ++b) { uv[4 * q + b] = (int)((ww[q] >> (8 * b)) & 0xffu); have[4 * q + b] = true; }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = 16 * c + e;
                uv[e] = FM_ENC_SHIFT;                // padding beyond dim: 0 after the shift
                have[e] = k < dim;
                if (k < dim) {
                    if constexpr (SRC_F32) {
                        const float f = ((const float*)src)[srow * dim + k];
                        const float fr = rintf(f);
                        if (!(f == fr) || f < 0.f || f > 255.f) bad = true;
                        else uv[e] = (int)fr;
                    } else {
                        uv[e] = ((const uint8_t*)src)[srow * dim + k];
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned word = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int u = uv[4 * q + b];
                if (have[4 * q + b]) usq += u * u;
                const int s = u - FM_ENC_SHIFT;      // 128: == (int8)(u ^ 0x80)
                sumsq += s * s;
                word |= (unsigned)(s & 0xff) << (8 * b);
            }
            w[q] = word;
        }
    }
    *(uint4*)(rows8 + row * kDim + 16 * c) = make_uint4(w[0], w[1], w[2], w[3]);
    sumsq += __shfl_xor(sumsq, 1);
    sumsq += __shfl_xor(sumsq, 2);
    sumsq += __shfl_xor(sumsq, 4);
    usq += __shfl_xor(usq, 1);
    usq += __shfl_xor(usq, 2);
    usq += __shfl_xor(usq, 4);
    usq = max(usq, __shfl_xor(usq, 8));
    usq = max(usq, __shfl_xor(usq, 16));
    usq = max(usq, __shfl_xor(usq, 32));
    if ((tid & 63) == 0 && usq > 0) atomicMax(nonint + 1, usq);
    if (c == 0) {
        // aux words of the 32-row unit in the accumulator order of v_mfma_i32_16x16x64_i8
        // (two 16-row tiles; tile row rr sits in lane group rr >> 2, register rr & 3)
        const int sub = r >> 4, rr = r & 15;
        const int id = 4 * sub + (rr & 3);
        int32_t* a = aux + tile * kAuxPerTile + 32 * sub;
        if (row < n) {
            norm[row] = sumsq;
            a[rr]      = -(sumsq >> 1);
            a[16 + rr] = ((1 - (sumsq & 1)) << 4) | (15 - id);
        } else {
            norm[row] = 0;
            a[rr]      = kPadCinit;
            a[16 + rr] = 15 - id;
        }
    }
    if constexpr (SRC_F32) {
        // one atomic per wave at most, and none once the flag is up (a bank that is not integer
        // valued would otherwise send one atomic per element to the same address)
        if (__builtin_amdgcn_ballot_w64(bad) != 0ull && (tid & 63) == 0 && *(volatile int*)nonint == 0) atomicOr(nonint, 1);
    }
  }
}

// float32 bank for the general (non-integer) route: zero-padded copy [n_pad][128].
__global__ void bank_copy_f32_kernel(const float* __restrict__ src, int64_t n, int dim,
                                     float* __restrict__ dst, int64_t n_pad, const int32_t* __restrict__ map)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad * kDim) return;
    const int64_t row = i / kDim;
    const int k = (int)(i % kDim);
    dst[i] = (row < n && k < dim) ? src[(map ? (int64_t)map[row] : row) * dim + k] : 0.f;
}

// fp16 rows, norms and accumulator inits of a float32 bank for the fp16 filter (filter_f16.hip).
// Pass 1: stat[0] = max |value| (float bits), stat[1] |= 1 if a value is not finite.
__global__ __launch_bounds__(256)
void bank_absmax_kernel(const float* __restrict__ rowsf, int64_t total, int* __restrict__ stat)
{
    float m = 0.f;
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const float v = fabsf(rowsf[i]);
        bad |= !(v <= 3.0e38f);
        m = fmaxf(m, v);
    }
#pragma unroll
    for (int mask = 1; mask < 64; mask <<= 1) m = fmaxf(m, __shfl_xor(m, mask));
    if (__builtin_amdgcn_ballot_w64(bad) != 0ull) { if ((threadIdx.x & 63) == 0) atomicOr(stat + 1, 1); }
    else if ((threadIdx.x & 63) == 0) atomicMax(stat, (int)__float_as_uint(m));     // m >= 0: bit order = value order
}

// Pass 2: 16 lanes per row, 8 dims per lane; rows scaled by 2^k (exact) and rounded to fp16
// (nearest even); norms of the scaled rows in float64 -> float32.  stat[0] = max norm.
__global__ __launch_bounds__(256)
void bank_prep_f16_kernel(const float* __restrict__ rowsf, int64_t n, int64_t n_pad, int k,
                          uint16_t* __restrict__ rowsh, float* __restrict__ normf,
                          float* __restrict__ auxf, int* __restrict__ stat)
{
    const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int c = threadIdx.x & 15;
    if (row >= n_pad) return;
    const float4 v0 = *(const float4*)(rowsf + row * kDim + 8 * c);
    const float4 v1 = *(const float4*)(rowsf + row * kDim + 8 * c + 4);
    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    unsigned h[8];
    double ss = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float x = ldexpf(v[i], k);
        const _Float16 hx = (_Float16)x;
        h[i] = (unsigned)__builtin_bit_cast(unsigned short, hx);
        ss += (double)x * (double)x;
    }
    *(uint4*)(rowsh + row * kDim + 8 * c) =
        make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
    ss += __shfl_xor(ss, 1);
    ss += __shfl_xor(ss, 2);
    ss += __shfl_xor(ss, 4);
    ss += __shfl_xor(ss, 8);
    if (c == 0) {
        const float nm = (float)ss;
        if (row < n) {
            normf[row] = nm;
            auxf[row] = -0.5f * nm;
        } else {
            normf[row] = 0.f;
            auxf[row] = -3.4e38f;
        }
    }
    // max norm: one atomic per wave (4 rows), not per row
    float m = (c == 0 && row < n) ? (float)ss : 0.f;
#pragma unroll
    for (int mask = 16; mask < 64; mask <<= 1) m = fmaxf(m, __shfl_xor(m, mask));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(stat, (int)__float_as_uint(m));
}

// ---------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------
extern "C" const char* fm_last_error(const fm_ctx* ctx)
{
    if (ctx) return ctx->err.c_str();
    std::lock_guard<std::mutex> lk(g_err_mu);
    static thread_local std::string copy;
    copy = g_err;
    return copy.c_str();
}

// Options of a context by name (include/fastmatch_hip.h lists them).
struct OptionDef { const char* name; int Tuning::* field; int lo, hi; const char* env; };
static const OptionDef kOptions[] = {
    {"nb", &Tuning::nb, 0, 8, "FM_NB"}, {"nsplit", &Tuning::nsplit, 0, 1 << 20, "FM_NSPLIT"}, {"nw", &Tuning::nw, 0, 16, "FM_NW"},
    {"nbuf", &Tuning::nbuf, 0, 3, "FM_NBUF"}, {"prio", &Tuning::prio, 0, 1, "FM_PRIO"},
    {"glds", &Tuning::glds, 0, 1, "FM_GLDS"}, {"coop", &Tuning::coop, 0, 1, "FM_COOP"},
    {"f32_filter", &Tuning::f32_filter, 0, 2, "FM_F32_FILTER"}, {"f32_nw", &Tuning::f32_nw, 0, 8, "FM_F32_NW"},
    {"f32_bound_every", &Tuning::f32_bound_every, 1, 64, "FM_F32_BOUND_EVERY"},
    {"f32_nsplit", &Tuning::f32_nsplit, 0, 1 << 20, "FM_F32_NSPLIT"}, {"f32_fused", &Tuning::f32_fused, -1, 1, "FM_F32_FUSED"},
    {"f32_lpc", &Tuning::f32_lpc, 0, 64, "FM_F32_LPC"},
    {"batch_group", &Tuning::batch_group, 1, kRRBatchMax, "FM_BATCH_GROUP"}, {"batch_tail", &Tuning::batch_tail, 0, kRRBatchMax, "FM_BATCH_TAIL"},
    {"async_time_every", &Tuning::async_time_every, 0, 1 << 20, "FM_ASYNC_TIME_EVERY"},
    {"k1_order", &Tuning::k1_order, 0, 2, "FM_K1_ORDER"}, {"bound_every", &Tuning::bound_every, 1, 1024, "FM_BOUND_EVERY"}, {"refill_grid", &Tuning::refill_grid, 1, 1 << 20, "FM_REFILL_GRID"},
    {"self_tri", &Tuning::self_tri, 0, 2, "FM_SELF_TRI"}, {"tri_stages", &Tuning::tri_stages, 0, 4096, "FM_TRI_STAGES"},
    {"expand_big", &Tuning::expand_big, 0, 1, nullptr}, {"expand_huge", &Tuning::expand_huge, 0, 1, nullptr}, {"expand_delegate", &Tuning::expand_delegate, 0, 1 << 30, "FM_EXPAND_DELEGATE"}, {"expand_grow", &Tuning::expand_grow, 0, 4, nullptr}, {"expand_prof", &Tuning::expand_prof, 0, 1, "FM_EXPAND_PROF"},
    {"delegated_rounds", &Tuning::delegated_rounds, 0, 0, nullptr},       // a counter: set to 0, read
};

extern "C" int fm_ctx_set_option(fm_ctx* ctx, const char* name, int64_t value)
{
    if (!ctx || !name) return fail(ctx, FM_EINVAL, "fm_ctx_set_option: NULL argument");
    for (const OptionDef& o : kOptions) {
        if (strcmp(o.name, name) != 0) continue;
        if (value < o.lo || value > o.hi) return fail(ctx, FM_EINVAL, std::string("fm_ctx_set_option: value out of range for ") + name);
        if (o.field == &Tuning::f32_filter && value != 0 && !ctx->d_counters)
            return fail(ctx, FM_EDEVICE, "fm_ctx_set_option: the fp16 filter's counters could not be allocated on this context");
        ctx->tune.*(o.field) = (int)value;
        return FM_OK;
    }
    return fail(ctx, FM_EINVAL, std::string("fm_ctx_set_option: unknown option ") + name);
}

extern "C" int fm_ctx_get_option(fm_ctx* ctx, const char* name, int64_t* value)
{
    if (!ctx || !name || !value) return fail(ctx, FM_EINVAL, "fm_ctx_get_option: NULL argument");
    for (const OptionDef& o : kOptions)
        if (strcmp(o.name, name) == 0) { *value = ctx->tune.*(o.field); return FM_OK; }
    return fail(ctx, FM_EINVAL, std::string("fm_ctx_get_option: unknown option ") + name);
}

extern "C" int fm_ctx_destroy(fm_ctx* ctx);

extern "C" int fm_ctx_create(int device_id, fm_ctx** out)
{
    if (!out) return fail(nullptr, FM_EINVAL, "fm_ctx_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return fail(nullptr, FM_EDEVICE,
                    std::string("fm_ctx_create: no HIP device available (") +
                    (e != hipSuccess ? hipGetErrorString(e) : "device count 0") +
                    "); libfastmatch_hip has no CPU fallback");
    }
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, FM_EINVAL, "fm_ctx_create: bad device id");
    fm_ctx* ctx = new (std::nothrow) fm_ctx();
    if (!ctx) return fail(nullptr, FM_ENOMEM, "fm_ctx_create: out of host memory");
    ctx->device = device_id;
    hipDeviceProp_t prop;
    if ((e = hipSetDevice(device_id)) != hipSuccess || (e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess) {
        delete ctx;
        return fail(nullptr, FM_EDEVICE, std::string("fm_ctx_create: ") + hipGetErrorString(e));
    }
    ctx->devname = std::string(prop.gcnArchName) + " " + prop.name;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        std::string m = "fm_ctx_create: device is " + ctx->devname + "; this library is built for gfx950 only";
        delete ctx;
        return fail(nullptr, FM_EUNSUPPORTED, m);
    }
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&ctx->ev_call0)) != hipSuccess || (e = hipEventCreate(&ctx->ev_call1)) != hipSuccess ||
        (e = hipEventCreate(&ctx->ev_k0)) != hipSuccess || (e = hipEventCreate(&ctx->ev_k1)) != hipSuccess) {
        delete ctx;
        return fail(nullptr, FM_EDEVICE, std::string("fm_ctx_create: ") + hipGetErrorString(e));
    }
    {
        // the tail stream gets the highest priority: its workgroups are few and short and should not
        // queue behind the thousands of workgroups of the K1 they overlap
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); least = greatest = 0; }
        if ((e = hipStreamCreateWithPriority(&ctx->tails[0], hipStreamNonBlocking, greatest)) != hipSuccess ||
            (e = hipStreamCreateWithPriority(&ctx->tails[1], hipStreamNonBlocking, greatest)) != hipSuccess ||
            (e = hipStreamCreateWithPriority(&ctx->tails[2], hipStreamNonBlocking, greatest)) != hipSuccess ||
            ((ctx->stream_tail = ctx->tails[0]), false) ||
            (e = hipEventCreateWithFlags(&ctx->aslot[0].tail_done, hipEventDisableTiming)) != hipSuccess ||
            (e = hipEventCreateWithFlags(&ctx->aslot[1].tail_done, hipEventDisableTiming)) != hipSuccess ||
            (e = hipEventCreateWithFlags(&ctx->ev_consumer, hipEventDisableTiming)) != hipSuccess ||
            (e = hipEventCreateWithFlags(&ctx->ev_tail_end[0], hipEventDisableTiming)) != hipSuccess ||
            (e = hipEventCreateWithFlags(&ctx->ev_tail_end[1], hipEventDisableTiming)) != hipSuccess ||
            (e = hipEventCreateWithFlags(&ctx->ev_tail_end[2], hipEventDisableTiming)) != hipSuccess ||
            (e = hipEventCreateWithFlags(&ctx->aslot[0].k_done, hipEventDisableTiming)) != hipSuccess ||
            (e = hipEventCreateWithFlags(&ctx->aslot[1].k_done, hipEventDisableTiming)) != hipSuccess) {
            fm_ctx_destroy(ctx);
            return fail(nullptr, FM_EDEVICE, std::string("fm_ctx_create: ") + hipGetErrorString(e));
        }
    }
    if (hipHostMalloc((void**)&ctx->h_scratch, 64, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); ctx->h_scratch = nullptr; }
    // defaults from the environment (experiments; a value out of range is ignored), then per context
    // through fm_ctx_set_option
    for (const OptionDef& o : kOptions)
        if (o.env) if (const char* s = getenv(o.env)) { const long v = atol(s); if (v >= o.lo && v <= o.hi) ctx->tune.*(o.field) = (int)v; }
    if (getenv("FM_EXPAND_NO_BIG")) ctx->tune.expand_big = 0;
    ctx->dbg_f32 = getenv("FM_F32_DEBUG") != nullptr;
    ctx->dbg_expand = getenv("FM_EXPAND_DEBUG") != nullptr;
    ctx->dbg_park = ctx->dbg_expand && getenv("FM_PARK_PROF") != nullptr;
    if (hipMalloc((void**)&ctx->d_counters, filter_flag_bytes()) != hipSuccess || hipMemset(ctx->d_counters, 0, filter_flag_bytes()) != hipSuccess) {
        (void)hipGetLastError();
        ctx->d_counters = nullptr;
        ctx->tune.f32_filter = 0;
    }
    *out = ctx;
    return FM_OK;
}

extern "C" int fm_ctx_destroy(fm_ctx* ctx)
{
    if (!ctx) return FM_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (hipStream_t ts : ctx->tails) if (ts) (void)hipStreamSynchronize(ts);
    auto free_slot = [](fm_ctx::AsyncSlot& sl) {
        if (sl.ws) (void)hipFree(sl.ws);
        if (sl.tail_done) (void)hipEventDestroy(sl.tail_done);
        if (sl.k_done) (void)hipEventDestroy(sl.k_done);
    };
    for (auto& m : ctx->marks) for (hipEvent_t ev : m.ev) if (ev) (void)hipEventDestroy(ev);
    for (auto& sl : ctx->aslot) free_slot(sl);
    for (auto& sl : ctx->bslot) free_slot(sl);
    if (ctx->upload) { (void)hipStreamSynchronize(ctx->upload); (void)hipStreamDestroy(ctx->upload); }
    if (ctx->ev_upload) (void)hipEventDestroy(ctx->ev_upload);
    if (ctx->ev_consumer) (void)hipEventDestroy(ctx->ev_consumer);
    for (hipEvent_t ev : ctx->ev_tail_end) if (ev) (void)hipEventDestroy(ev);
    if (ctx->comm) { comm_destroy(ctx->comm); ctx->comm = nullptr; }      // (before the streams it was used on)
    for (hipStream_t ts : ctx->tails) if (ts) (void)hipStreamDestroy(ts);
    for (auto* v : {&ctx->pending, &ctx->timer_pool})
        for (auto& t : *v) { (void)hipEventDestroy(t.c0); (void)hipEventDestroy(t.c1); (void)hipEventDestroy(t.k0); (void)hipEventDestroy(t.k1); }
    if (ctx->ws_partial) (void)hipFree(ctx->ws_partial);
    if (ctx->ws_out) (void)hipFree(ctx->ws_out);
    if (ctx->ws_in) (void)hipFree(ctx->ws_in);
    if (ctx->h_scratch) (void)hipHostFree(ctx->h_scratch);
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    if (ctx->d_counters) (void)hipFree(ctx->d_counters);
    if (ctx->ev_call0) (void)hipEventDestroy(ctx->ev_call0);
    if (ctx->ev_call1) (void)hipEventDestroy(ctx->ev_call1);
    if (ctx->ev_k0) (void)hipEventDestroy(ctx->ev_k0);
    if (ctx->ev_k1) (void)hipEventDestroy(ctx->ev_k1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return FM_OK;
}

// Account the calls that were enqueued without a synchronisation; the stream must be idle.
int fm::drain_pending(fm_ctx* ctx)
{
    for (auto& t : ctx->pending) {
        float ms = 0.f;
        if (t.call_timed) {      // enqueue-to-results latency of the call (overlapped calls: not additive)
            if (hipEventElapsedTime(&ms, t.k0, t.c1) == hipSuccess) { ctx->stats.total_ms += ms; ctx->stats.calls += 1; }
            else (void)hipGetLastError();
        }
        if (t.timed) {
            if (hipEventElapsedTime(&ms, t.k0, t.k1) == hipSuccess) {
                ctx->stats.kernel_ms += ms;
                ctx->stats.kernel_launches += 1;
                ctx->stats.pairs += t.pairs;
                ctx->stats_bytes += t.bytes;
            } else (void)hipGetLastError();
        }
        ctx->timer_pool.push_back(t);
    }
    ctx->pending.clear();
    return FM_OK;
}

extern "C" int fm_sync(fm_ctx* ctx)
{
    if (!ctx) return fail(nullptr, FM_EINVAL, "fm_sync: ctx is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->upload) HIP_TRY(ctx, hipStreamSynchronize(ctx->upload));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (hipStream_t ts : ctx->tails) HIP_TRY(ctx, hipStreamSynchronize(ts));
    return drain_pending(ctx);
}

// fm_mark: remember "everything enqueued on this context so far"; fm_wait: block until that point is
// reached.  Work enqueued after the mark keeps running: a consumer can read the results of batch i while
// batch i + 1 is already on the device (double-buffered outputs) -- fm_sync would drain both.
extern "C" int fm_mark(fm_ctx* ctx, int64_t* ticket)
{
    if (!ctx || !ticket) return fail(ctx, FM_EINVAL, "fm_mark: NULL argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    fm_ctx::Mark& m = ctx->marks[ctx->next_mark % fm_ctx::kMarks];
    for (int u = 0; u < 1 + fm_ctx::kTails; ++u) {
        if (!m.ev[u]) HIP_TRY(ctx, hipEventCreateWithFlags(&m.ev[u], hipEventDisableTiming));
        HIP_TRY(ctx, hipEventRecord(m.ev[u], u == 0 ? ctx->stream : ctx->tails[u - 1]));
    }
    m.id = ctx->next_mark;
    *ticket = ctx->next_mark++;
    return FM_OK;
}

extern "C" int fm_wait(fm_ctx* ctx, int64_t ticket)
{
    if (!ctx) return fail(nullptr, FM_EINVAL, "fm_wait: ctx is NULL");
    if (ticket < 0 || ticket >= ctx->next_mark) return fail(ctx, FM_EINVAL, "fm_wait: unknown ticket");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    fm_ctx::Mark& m = ctx->marks[ticket % fm_ctx::kMarks];
    if (m.id != ticket) {            // the slot has been re-used by a later mark: everything older is covered by it
        if (m.id < ticket) return fail(ctx, FM_EINVAL, "fm_wait: unknown ticket");
    }
    for (hipEvent_t ev : m.ev) if (ev) HIP_TRY(ctx, hipEventSynchronize(ev));
    return FM_OK;
}

extern "C" int fm_get_stats(fm_ctx* ctx, fm_stats* out)
{
    if (!ctx || !out) return fail(ctx, FM_EINVAL, "fm_get_stats: NULL argument");
    if (!ctx->pending.empty()) {
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (hipStream_t ts : ctx->tails) HIP_TRY(ctx, hipStreamSynchronize(ts));
        drain_pending(ctx);
    }
    *out = ctx->stats;
    return FM_OK;
}

extern "C" int fm_get_stats_ex(fm_ctx* ctx, fm_stats_ex* out, int64_t out_bytes)
{
    if (!ctx || !out || out_bytes < 8) return fail(ctx, FM_EINVAL, "fm_get_stats_ex: bad argument");
    fm_stats st;
    int rc = fm_get_stats(ctx, &st);
    if (rc != FM_OK) return rc;
    fm_stats_ex ex{};
    ex.struct_bytes = (int64_t)sizeof(fm_stats_ex);
    ex.kernel_ms = st.kernel_ms; ex.total_ms = st.total_ms; ex.kernel_launches = st.kernel_launches;
    ex.pairs = st.pairs; ex.calls = st.calls;
    ex.bytes_moved = ctx->stats_bytes;
    memcpy(out, &ex, (size_t)(out_bytes < (int64_t)sizeof(ex) ? out_bytes : (int64_t)sizeof(ex)));
    return FM_OK;
}

extern "C" int fm_abi_version(void) { return FM_ABI_VERSION; }

extern "C" int fm_reset_stats(fm_ctx* ctx)
{
    if (!ctx) return fail(nullptr, FM_EINVAL, "fm_reset_stats: ctx is NULL");
    ctx->stats = fm_stats{};
    ctx->stats_bytes = 0;
    return FM_OK;
}

extern "C" int fm_f32_filter_stats(fm_ctx* ctx, int64_t* launches, int64_t* fallbacks)
{
    if (!ctx) return fail(nullptr, FM_EINVAL, "fm_f32_filter_stats: ctx is NULL");
    int c[4] = {0, 0, 0, 0};
    if (ctx->d_counters) {
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipMemcpy(c, ctx->d_counters, 16, hipMemcpyDeviceToHost));
    }
    if (ctx->dbg_f32) fprintf(stderr, "[fm] filter: launches %lld, redone by K5 %d, output rows rescanned %d\n",
                                        (long long)ctx->filter_launches, c[2], c[3]);
    if (launches) *launches = ctx->filter_launches;
    if (fallbacks) *fallbacks = c[2];
    return FM_OK;
}

extern "C" int fm_device_name(fm_ctx* ctx, char* buf, int buflen)
{
    if (!ctx || !buf || buflen <= 0) return fail(ctx, FM_EINVAL, "fm_device_name: bad argument");
    snprintf(buf, (size_t)buflen, "%s", ctx->devname.c_str());
    return FM_OK;
}

// Page-locked allocations made through fm_host_alloc, with their device-side aliases: the async entry
// points look up to fifty output pointers per call, and a runtime query per pointer (microseconds each)
// would sit in front of the first launch of a batch.
struct PinnedRange { size_t bytes; char* dev; };
static std::mutex g_pinned_mu;
static std::map<uintptr_t, PinnedRange> g_pinned;

extern "C" int fm_host_alloc(fm_ctx* ctx, int64_t bytes, void** out)
{
    if (!ctx || !out || bytes < 0) return fail(ctx, FM_EINVAL, "fm_host_alloc: bad argument");
    *out = nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t sz = (size_t)(bytes > 0 ? bytes : 1);
    HIP_TRY(ctx, hipHostMalloc(out, sz, hipHostMallocDefault));
    void* dev = nullptr;
    if (hipHostGetDevicePointer(&dev, *out, 0) == hipSuccess && dev) {
        std::lock_guard<std::mutex> lk(g_pinned_mu);
        g_pinned[(uintptr_t)*out] = PinnedRange{sz, (char*)dev};
    } else (void)hipGetLastError();
    return FM_OK;
}

extern "C" int fm_host_free(fm_ctx* ctx, void* p)
{
    if (!p) return FM_OK;
    if (ctx) (void)hipSetDevice(ctx->device);
    { std::lock_guard<std::mutex> lk(g_pinned_mu); g_pinned.erase((uintptr_t)p); }
    hipError_t e = hipHostFree(p);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(ctx, FM_EDEVICE, std::string("fm_host_free: ") + hipGetErrorString(e)); }
    return FM_OK;
}

// Device-side alias of a page-locked host buffer (fm_host_alloc / hipHostMalloc), or NULL for
// ordinary pageable memory: kernels can then write results straight into the caller's buffer.
void* fm::pinned_device_alias(const void* host)
{
    if (!host) return nullptr;
    {
        std::lock_guard<std::mutex> lk(g_pinned_mu);
        auto it = g_pinned.upper_bound((uintptr_t)host);
        if (it != g_pinned.begin()) {
            --it;
            const size_t off = (uintptr_t)host - it->first;
            if (off < it->second.bytes) return it->second.dev + off;
        }
    }
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, host) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (at.type != hipMemoryTypeHost) return nullptr;
    return at.devicePointer;
}

// Copy through the kernel's own stores (dst is the device alias of page-locked host memory).
__global__ void copy_out_kernel(unsigned char* __restrict__ dst, const unsigned char* __restrict__ src, size_t bytes)
{
    const size_t words = bytes / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += stride)
        ((unsigned*)dst)[i] = ((const unsigned*)src)[i];
    if (blockIdx.x == 0 && threadIdx.x < (bytes & 3)) dst[words * 4 + threadIdx.x] = src[words * 4 + threadIdx.x];
}

// Device -> caller memory on the context's stream.  A copy into pageable memory makes the
// runtime pin the destination pages for the transfer (milliseconds for results of ~100 KB and
// up), so such results land in the context's own page-locked staging buffer and are moved to
// the caller by CallScope::finish() after the call's single synchronisation.
static hipError_t copy_out(fm_ctx* ctx, unsigned char* dst_alias, const void* src, size_t bytes)
{
    const unsigned grid = (unsigned)((bytes / 4 + 255) / 256 < 1024 ? (bytes / 4 + 255) / 256 + 1 : 1024);
    hipLaunchKernelGGL(copy_out_kernel, dim3(grid), dim3(256), 0, ctx->stream, dst_alias, (const unsigned char*)src, bytes);
    return hipGetLastError();
}

hipError_t fm::d2h(fm_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    if (bytes == 0) return hipSuccess;
    // A copy kernel rather than hipMemcpyAsync for anything but tiny results: the runtime hands
    // device-to-host copies of 64 KiB and more to a DMA queue behind a host-side wait for the
    // stream, which was seen to add 1-7 ms of idle time after multi-millisecond kernels.
    const bool kernel_ok = bytes >= 4096 && bytes <= ((size_t)256 << 20) && ((uintptr_t)src & 3) == 0;
    if (kernel_ok) {
        if (unsigned char* direct = (unsigned char*)pinned_device_alias(dst))       // page-locked destination
            return copy_out(ctx, direct, src, bytes);
        size_t off = (ctx->h_stage_used + 63) & ~(size_t)63;
        if (off + bytes > ctx->h_stage_bytes && ctx->staged.empty()) {
            if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
            ctx->h_stage = nullptr;
            ctx->h_stage_bytes = 0;
            const size_t want = bytes * 3 + (1 << 20);
            if (hipHostMalloc((void**)&ctx->h_stage, want, hipHostMallocDefault) == hipSuccess) ctx->h_stage_bytes = want;
            else { (void)hipGetLastError(); ctx->h_stage = nullptr; }
            off = 0;
        }
        if (ctx->h_stage && off + bytes <= ctx->h_stage_bytes) {
            if (unsigned char* alias = (unsigned char*)pinned_device_alias(ctx->h_stage + off)) {
                hipError_t e = copy_out(ctx, alias, src, bytes);
                if (e != hipSuccess) return e;
                ctx->staged.push_back({dst, off, bytes});
                ctx->h_stage_used = off + bytes;
                return hipSuccess;
            }
        }
    }
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream);
}
// ---------------------------------------------------------------------------------------
// banks
// ---------------------------------------------------------------------------------------
static void bank_free(Bank* b)
{
    if (b->rows8) (void)hipFree(b->rows8);
    if (b->norm) (void)hipFree(b->norm);
    if (b->aux) (void)hipFree(b->aux);
    if (b->rowsf) (void)hipFree(b->rowsf);
    if (b->rowsh) (void)hipFree(b->rowsh);
    if (b->normf) (void)hipFree(b->normf);
    if (b->auxf) (void)hipFree(b->auxf);
    if (b->selfdist) (void)hipFree(b->selfdist);
    if (b->stage) (void)hipFree(b->stage);
    b->stage = nullptr;
    b->rows8 = nullptr; b->norm = nullptr; b->aux = nullptr; b->rowsf = nullptr; b->selfdist = nullptr;
    b->rowsh = nullptr; b->normf = nullptr; b->auxf = nullptr;
}

// map != nullptr: the bank's row i is rows[map[i]] of the n_src source rows (0 <= map[i] < n_src, checked here).
static int bank_create(fm_ctx* ctx, const void* rows, int64_t n, int dim, bool f32, fm_bank** out, bool keep_f32 = false,
                       int64_t capacity = 0, const int32_t* map = nullptr, int64_t n_src = 0)
{
    if (!ctx) return fail(nullptr, FM_EINVAL, "fm_bank_create: ctx is NULL");
    if (!out) return fail(ctx, FM_EINVAL, "fm_bank_create: bank out pointer is NULL");
    *out = nullptr;
    if (n < 0 || dim < 1 || (n > 0 && !rows)) return fail(ctx, FM_EINVAL, "fm_bank_create: bad rows/n/dim");
    if (map) {
        if (n_src < 0 || n_src > INT32_MAX) return fail(ctx, FM_EINVAL, "fm_bank_create_*_gather: bad n_src");
        for (int64_t i = 0; i < n; ++i)
            if (map[i] < 0 || map[i] >= n_src) return fail(ctx, FM_EINVAL, "fm_bank_create_*_gather: src_row entry outside [0, n_src)");
    }
    if (dim > kDim) return fail(ctx, FM_EUNSUPPORTED, "fm_bank_create: dim > 128 is not supported");
    if (n > (int64_t)INT32_MAX - 2 * kStageRows) return fail(ctx, FM_EUNSUPPORTED, "fm_bank_create: n too large");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    fm_bank* b = new (std::nothrow) fm_bank();
    if (!b) return fail(ctx, FM_ENOMEM, "fm_bank_create: out of host memory");
    b->n = n;
    b->dim = dim;
    b->n_pad = ((n + kStageRows - 1) / kStageRows) * kStageRows;
    if (b->n_pad == 0) b->n_pad = kStageRows;
    b->cap_pad = b->n_pad;
    if (capacity > n) {
        // room to grow (fm_bank_append_u8): the arrays are sized, and every tile prepared as padding, for `capacity` rows
        if (capacity > (int64_t)INT32_MAX - 2 * kStageRows) { delete b; return fail(ctx, FM_EUNSUPPORTED, "fm_bank_create: capacity too large"); }
        b->cap_pad = ((capacity + kStageRows - 1) / kStageRows) * kStageRows;
    }
    b->kind = FM_BANK_I8;
    const size_t elt = f32 ? 4 : 1;
    const size_t src_bytes = (size_t)(map ? n_src : n) * dim * elt;
    int rc = FM_OK;
    auto bail = [&](int code) { bank_free(b); delete b; return code; };

    const size_t flag_off = (src_bytes + 15) & ~(size_t)15;
    if ((rc = ws_ensure(ctx, &ctx->ws_in, &ctx->ws_in_bytes, flag_off + 32 + (map ? (size_t)n * 4 : 0))) != FM_OK) return bail(rc);
    int* d_flag = (int*)((char*)ctx->ws_in + flag_off);
    const int32_t* d_map = map ? (const int32_t*)((char*)ctx->ws_in + flag_off + 32) : nullptr;
#define BTRY(expr)                                                                               \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            (void)hipGetLastError();                                                             \
            return bail(fail(ctx, _e == hipErrorOutOfMemory ? FM_ENOMEM : FM_EDEVICE,            \
                             std::string(#expr " failed: ") + hipGetErrorString(_e)));           \
        }                                                                                        \
    } while (0)
    BTRY(hipMalloc((void**)&b->rows8, (size_t)b->cap_pad * kDim));
    BTRY(hipMalloc((void**)&b->norm, (size_t)b->cap_pad * 4));
    BTRY(hipMalloc((void**)&b->aux, (size_t)(b->cap_pad / kTileRows) * kAuxPerTile * 4));
    if (src_bytes) BTRY(hipMemcpyAsync(ctx->ws_in, rows, src_bytes, hipMemcpyHostToDevice, ctx->stream));
    if (map && n > 0) BTRY(hipMemcpyAsync((void*)d_map, map, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    BTRY(hipMemsetAsync(d_flag, 0, 8, ctx->stream));
    const int ntiles = (int)(b->cap_pad / kTileRows);
    if (f32)
        hipLaunchKernelGGL(bank_prep_kernel<true>, dim3(ntiles), dim3(256), 0, ctx->stream,
                           (const void*)ctx->ws_in, n, dim, b->rows8, b->norm, b->aux, d_flag, (int64_t)ntiles, d_map);
    else
        hipLaunchKernelGGL(bank_prep_kernel<false>, dim3(ntiles), dim3(256), 0, ctx->stream,
                           (const void*)ctx->ws_in, n, dim, b->rows8, b->norm, b->aux, d_flag, (int64_t)ntiles, d_map);
    BTRY(hipGetLastError());
    int flags[2] = {0, 0};
    BTRY(hipMemcpyAsync(flags, d_flag, 8, hipMemcpyDeviceToHost, ctx->stream));
    BTRY(hipStreamSynchronize(ctx->stream));
    const int flag = flags[0];
    b->usq_max = flags[1];
    if (f32 && (flag || (keep_f32 && n > 0))) {
        // not integer-valued (or the caller wants the float32 route): keep a float32 bank for the fma-chain route
        b->kind = FM_BANK_F32;
        (void)hipFree(b->rows8); b->rows8 = nullptr;
        (void)hipFree(b->aux); b->aux = nullptr;
        BTRY(hipMalloc((void**)&b->rowsf, (size_t)b->n_pad * kDim * 4));
        const int64_t tot = b->n_pad * kDim;
        hipLaunchKernelGGL(bank_copy_f32_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const float*)ctx->ws_in, n, dim, b->rowsf, b->n_pad, d_map);
        BTRY(hipGetLastError());
        // rows for the fp16 filter, scaled by the power of two that puts the largest magnitude
        // of the bank in [2^13, 2^14)
        BTRY(hipMalloc((void**)&b->rowsh, (size_t)b->n_pad * kDim * 2));
        BTRY(hipMalloc((void**)&b->normf, (size_t)b->n_pad * 4));
        BTRY(hipMalloc((void**)&b->auxf, (size_t)b->n_pad * 4));
        BTRY(hipMemsetAsync(d_flag, 0, 8, ctx->stream));
        hipLaunchKernelGGL(bank_absmax_kernel, dim3(1024), dim3(256), 0, ctx->stream, (const float*)b->rowsf, tot, d_flag);
        BTRY(hipGetLastError());
        int stat[2] = {0, 0};
        BTRY(hipMemcpyAsync(stat, d_flag, 8, hipMemcpyDeviceToHost, ctx->stream));
        BTRY(hipStreamSynchronize(ctx->stream));
        float vmax = 0.f;
        memcpy(&vmax, &stat[0], 4);
        b->filt_ok = stat[1] == 0;
        if (b->filt_ok) {
            int ex = 0;
            if (vmax > 0.f) (void)frexpf(vmax, &ex);         // vmax = m 2^ex, m in [0.5, 1)
            b->kscale = vmax > 0.f ? 14 - ex : 0;
            BTRY(hipMemsetAsync(d_flag, 0, 8, ctx->stream));
            hipLaunchKernelGGL(bank_prep_f16_kernel, dim3((unsigned)(b->n_pad / 16)), dim3(256), 0, ctx->stream,
                               (const float*)b->rowsf, n, b->n_pad, b->kscale, b->rowsh, b->normf, b->auxf, d_flag);
            BTRY(hipGetLastError());
            BTRY(hipMemcpyAsync(stat, d_flag, 8, hipMemcpyDeviceToHost, ctx->stream));
            BTRY(hipStreamSynchronize(ctx->stream));
            memcpy(&b->nm_max, &stat[0], 4);
        }
    }
#undef BTRY
    *out = b;
    return FM_OK;
}

extern "C" int fm_bank_create_u8(fm_ctx* ctx, const uint8_t* rows, int64_t n, int dim, fm_bank** bank)
{
    return bank_create(ctx, rows, n, dim, false, bank);
}

// Banks whose rows are a gather of the caller's rows (a keypoint in several Grid_Cache cells: uploaded once).
extern "C" int fm_bank_create_u8_gather(fm_ctx* ctx, const uint8_t* rows, int64_t n_src, int dim, const int32_t* src_row, int64_t n,
                                        fm_bank** bank)
{
    if (n > 0 && !src_row) return fail(ctx, FM_EINVAL, "fm_bank_create_u8_gather: src_row is NULL");
    if (!src_row) return bank_create(ctx, rows, 0, dim, false, bank);
    return bank_create(ctx, rows, n, dim, false, bank, false, 0, src_row, n_src);
}

extern "C" int fm_bank_create_f32_gather(fm_ctx* ctx, const float* rows, int64_t n_src, int dim, int float_route, const int32_t* src_row,
                                         int64_t n, fm_bank** bank)
{
    if (n > 0 && !src_row) return fail(ctx, FM_EINVAL, "fm_bank_create_f32_gather: src_row is NULL");
    if (!src_row) return bank_create(ctx, rows, 0, dim, true, bank, float_route != 0);
    return bank_create(ctx, rows, n, dim, true, bank, float_route != 0, 0, src_row, n_src);
}

extern "C" int fm_bank_create_u8_cap(fm_ctx* ctx, const uint8_t* rows, int64_t n, int dim, int64_t capacity, fm_bank** bank)
{
    if (capacity < n) return fail(ctx, FM_EINVAL, "fm_bank_create_u8_cap: capacity < n");
    return bank_create(ctx, rows, n, dim, false, bank, false, capacity);
}

// Rows appended to a bank created with room for them, at the next multiple of 32 rows (whole MFMA tiles: the rows
// of the tile an earlier append ended in stay what they are, padding).  Synchronous.
extern "C" int fm_bank_append_u8(fm_ctx* ctx, fm_bank* bank, const uint8_t* rows, int64_t n, int64_t* first_row)
{
    if (!ctx || !bank) return fail(ctx, FM_EINVAL, "fm_bank_append_u8: NULL argument");
    if (bank->kind != FM_BANK_I8 || !bank->rows8) return fail(ctx, FM_EINVAL, "fm_bank_append_u8: not an integer-route bank");
    if (n < 0 || (n > 0 && !rows)) return fail(ctx, FM_EINVAL, "fm_bank_append_u8: bad rows / n");
    const int64_t off = ((bank->n + kTileRows - 1) / kTileRows) * kTileRows;
    if (first_row) *first_row = off;
    if (n == 0) return FM_OK;
    if (off + n > bank->cap_pad) return fail(ctx, FM_EINVAL, "fm_bank_append_u8: the bank's capacity is used up");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t src_bytes = (size_t)n * bank->dim, flag_off = (src_bytes + 15) & ~(size_t)15;
    int rc = ws_ensure(ctx, &ctx->ws_in, &ctx->ws_in_bytes, flag_off + 32);
    if (rc != FM_OK) return rc;
    int* d_flag = (int*)((char*)ctx->ws_in + flag_off);
    HIP_TRY(ctx, hipMemcpyAsync(ctx->ws_in, rows, src_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(d_flag, 0, 8, ctx->stream));
    const int64_t ntiles = (n + kTileRows - 1) / kTileRows;
    hipLaunchKernelGGL(bank_prep_kernel<false>, dim3((unsigned)ntiles), dim3(256), 0, ctx->stream, (const void*)ctx->ws_in, n, bank->dim,
                       bank->rows8 + (size_t)off * kDim, bank->norm + off, bank->aux + (off / kTileRows) * kAuxPerTile, d_flag, ntiles);
    HIP_TRY(ctx, hipGetLastError());
    int flags[2] = {0, 0};
    HIP_TRY(ctx, hipMemcpyAsync(flags, d_flag, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (flags[1] > bank->usq_max) bank->usq_max = flags[1];
    bank->n = off + n;
    bank->n_pad = ((bank->n + kStageRows - 1) / kStageRows) * kStageRows;
    return FM_OK;
}

// ---- a float32-route bank that grows (r05: lazy targets with descriptors that are not integer valued) -----------------
// Rows of the new range: largest magnitude (float bits) in stat[0], stat[1] |= 1 for a value that is not finite.
__global__ __launch_bounds__(256)
void bank_append_f32_kernel(const float* __restrict__ src, int64_t n, int dim, float* __restrict__ dst, int* __restrict__ stat)
{
    float m = 0.f;
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n * kDim; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / kDim;
        const int k = (int)(i % kDim);
        const float v = k < dim ? src[row * dim + k] : 0.f;
        dst[i] = v;
        bad |= !(fabsf(v) <= 3.0e38f);
        m = fmaxf(m, fabsf(v));
    }
#pragma unroll
    for (int mask = 1; mask < 64; mask <<= 1) m = fmaxf(m, __shfl_xor(m, mask));
    if (__builtin_amdgcn_ballot_w64(bad) != 0ull) { if ((threadIdx.x & 63) == 0) atomicOr(stat + 1, 1); }
    else if ((threadIdx.x & 63) == 0) atomicMax(stat, (int)__float_as_uint(m));
}

// An EMPTY float32-route bank with room for `capacity` rows.  The fp16 planes of a bank are scaled by ONE power of two
// chosen from the bank's largest magnitude (filter_f16.hip); a bank that grows cannot know its own, so it takes the
// scale of `scale_like` -- the bank it will be matched against (the query image's: same extractor, same value range).
// Every row is prepared as a padding row; fm_bank_append_f32 turns ranges of them into real ones.
extern "C" int fm_bank_create_f32_cap(fm_ctx* ctx, int dim, int64_t capacity, const fm_bank* scale_like, fm_bank** out)
{
    if (!ctx) return fail(nullptr, FM_EINVAL, "fm_bank_create_f32_cap: ctx is NULL");
    if (!out) return fail(ctx, FM_EINVAL, "fm_bank_create_f32_cap: bank out pointer is NULL");
    *out = nullptr;
    if (dim < 1 || dim > kDim || capacity < 1 || capacity > (int64_t)INT32_MAX - 2 * kStageRows)
        return fail(ctx, FM_EINVAL, "fm_bank_create_f32_cap: bad dim / capacity");
    if (!scale_like || scale_like->kind != FM_BANK_F32 || !scale_like->filt_ok)
        return fail(ctx, FM_EINVAL, "fm_bank_create_f32_cap: scale_like must be a float32-route bank of finite values");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    fm_bank* b = new (std::nothrow) fm_bank();
    if (!b) return fail(ctx, FM_ENOMEM, "fm_bank_create_f32_cap: out of host memory");
    b->kind = FM_BANK_F32;
    b->n = 0;
    b->dim = dim;
    b->n_pad = kStageRows;
    b->cap_pad = ((capacity + kStageRows - 1) / kStageRows) * kStageRows;
    b->kscale = scale_like->kscale;
    b->filt_ok = true;
    b->nm_max = 0.f;
    auto bail = [&](hipError_t e, const char* what) {
        (void)hipGetLastError();
        bank_free(b); delete b;
        return fail(ctx, e == hipErrorOutOfMemory ? FM_ENOMEM : FM_EDEVICE, std::string("fm_bank_create_f32_cap: ") + what + ": " + hipGetErrorString(e));
    };
    hipError_t e;
    if ((e = hipMalloc((void**)&b->rowsf, (size_t)b->cap_pad * kDim * 4)) != hipSuccess) return bail(e, "rows");
    if ((e = hipMalloc((void**)&b->rowsh, (size_t)b->cap_pad * kDim * 2)) != hipSuccess) return bail(e, "fp16 rows");
    if ((e = hipMalloc((void**)&b->normf, (size_t)b->cap_pad * 4)) != hipSuccess) return bail(e, "norms");
    if ((e = hipMalloc((void**)&b->auxf, (size_t)b->cap_pad * 4)) != hipSuccess) return bail(e, "aux");
    if ((e = hipMemsetAsync(b->rowsf, 0, (size_t)b->cap_pad * kDim * 4, ctx->stream)) != hipSuccess) return bail(e, "fill");
    int rc = ws_ensure(ctx, &ctx->ws_in, &ctx->ws_in_bytes, 64);
    if (rc != FM_OK) { bank_free(b); delete b; return rc; }
    // (n = 0: every row comes out as a padding row -- zero fp16 row, norm 0, accumulator init -3.4e38)
    hipLaunchKernelGGL(bank_prep_f16_kernel, dim3((unsigned)(b->cap_pad / 16)), dim3(256), 0, ctx->stream,
                       (const float*)b->rowsf, (int64_t)0, b->cap_pad, b->kscale, b->rowsh, b->normf, b->auxf, (int*)ctx->ws_in);
    if ((e = hipGetLastError()) != hipSuccess || (e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return bail(e, "prepare");
    *out = b;
    return FM_OK;
}

// n more rows at the next multiple of 32 rows, as fm_bank_append_u8.  FM_EUNSUPPORTED (the bank unchanged, its rows past
// bank->n rewritten as padding) when a value is not finite or, scaled by the bank's fixed power of two, leaves fp16's range.
extern "C" int fm_bank_append_f32(fm_ctx* ctx, fm_bank* bank, const float* rows, int64_t n, int64_t* first_row)
{
    if (!ctx || !bank) return fail(ctx, FM_EINVAL, "fm_bank_append_f32: NULL argument");
    if (bank->kind != FM_BANK_F32 || !bank->rowsf || !bank->rowsh || bank->cap_pad <= 0)
        return fail(ctx, FM_EINVAL, "fm_bank_append_f32: not a float32-route bank with capacity (fm_bank_create_f32_cap)");
    if (n < 0 || (n > 0 && !rows)) return fail(ctx, FM_EINVAL, "fm_bank_append_f32: bad rows / n");
    const int64_t off = ((bank->n + kTileRows - 1) / kTileRows) * kTileRows;
    if (first_row) *first_row = off;
    if (n == 0) return FM_OK;
    if (off + n > bank->cap_pad) return fail(ctx, FM_EINVAL, "fm_bank_append_f32: the bank's capacity is used up");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t src_bytes = (size_t)n * bank->dim * 4, flag_off = (src_bytes + 15) & ~(size_t)15;
    int rc = ws_ensure(ctx, &ctx->ws_in, &ctx->ws_in_bytes, flag_off + 32);
    if (rc != FM_OK) return rc;
    int* d_flag = (int*)((char*)ctx->ws_in + flag_off);
    HIP_TRY(ctx, hipMemcpyAsync(ctx->ws_in, rows, src_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(d_flag, 0, 16, ctx->stream));
    float* dst = bank->rowsf + (size_t)off * kDim;
    hipLaunchKernelGGL(bank_append_f32_kernel, dim3((unsigned)std::min<int64_t>(1024, (n * kDim + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const float*)ctx->ws_in, n, bank->dim, dst, d_flag);
    HIP_TRY(ctx, hipGetLastError());
    int stat[2] = {0, 0};
    HIP_TRY(ctx, hipMemcpyAsync(stat, d_flag, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    float vmax = 0.f;
    memcpy(&vmax, &stat[0], 4);
    const int64_t n16 = ((n + 15) / 16) * 16;                 // (the range's rows up to a multiple of 16: padding again)
    const bool ok = stat[1] == 0 && ldexpf(vmax, bank->kscale) < 60000.f;
    HIP_TRY(ctx, hipMemsetAsync(d_flag + 2, 0, 8, ctx->stream));
    if (!ok) HIP_TRY(ctx, hipMemsetAsync(dst, 0, (size_t)n * kDim * 4, ctx->stream));
    hipLaunchKernelGGL(bank_prep_f16_kernel, dim3((unsigned)(n16 / 16)), dim3(256), 0, ctx->stream,
                       (const float*)dst, ok ? n : (int64_t)0, n16, bank->kscale, bank->rowsh + (size_t)off * kDim, bank->normf + off,
                       bank->auxf + off, d_flag + 2);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(stat, d_flag + 2, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (!ok)
        return fail(ctx, FM_EUNSUPPORTED, "fm_bank_append_f32: a value is not finite or leaves the fp16 range under the bank's scale (2^" +
                                          std::to_string(bank->kscale) + "): the growing bank cannot take these rows");
    float nmx = 0.f;
    memcpy(&nmx, &stat[0], 4);
    if (nmx > bank->nm_max) bank->nm_max = nmx;
    bank->n = off + n;
    bank->n_pad = ((bank->n + kStageRows - 1) / kStageRows) * kStageRows;
    return FM_OK;
}

extern "C" int fm_bank_create_f32(fm_ctx* ctx, const float* rows, int64_t n, int dim, fm_bank** bank)
{
    return bank_create(ctx, rows, n, dim, true, bank);
}

extern "C" int fm_bank_create_f32_route(fm_ctx* ctx, const float* rows, int64_t n, int dim, fm_bank** bank)
{
    return bank_create(ctx, rows, n, dim, true, bank, true);
}

// A new image's descriptors into an existing bank: no allocation, no host synchronisation.  The copy (DMA engine)
// and the preparation kernel run on the context's upload stream, beside whatever the other streams compute.
extern "C" int fm_bank_refill_u8_async(fm_ctx* ctx, fm_bank* bank, const uint8_t* rows, int64_t n)
{
    if (!ctx || !bank) return fail(ctx, FM_EINVAL, "fm_bank_refill_u8_async: NULL argument");
    if (bank->kind != FM_BANK_I8 || !bank->rows8) return fail(ctx, FM_EINVAL, "fm_bank_refill_u8_async: not an integer-route bank");
    if (n < 0 || (n > 0 && !rows)) return fail(ctx, FM_EINVAL, "fm_bank_refill_u8_async: bad rows / n");
    int64_t n_pad = ((n + kStageRows - 1) / kStageRows) * kStageRows;
    if (n_pad == 0) n_pad = kStageRows;
    if (n_pad > bank->cap_pad) return fail(ctx, FM_EINVAL, "fm_bank_refill_u8_async: more rows than the bank was created with");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->upload) {
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->upload, hipStreamNonBlocking));
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_upload, hipEventDisableTiming));
    }
    if (!bank->stage) HIP_TRY(ctx, hipMalloc(&bank->stage, (size_t)bank->cap_pad * kDim + 64));
    int* d_flag = (int*)((char*)bank->stage + (size_t)bank->cap_pad * kDim);
    if (n > 0) HIP_TRY(ctx, hipMemcpyAsync(bank->stage, rows, (size_t)n * bank->dim, hipMemcpyHostToDevice, ctx->upload));
    // (the flag words are scratch here: the largest row norm is not read back, see below -- no fill in front of the kernel)
    const int64_t ntiles = n_pad / kTileRows;
    const int64_t grid = ntiles < ctx->tune.refill_grid ? ntiles : ctx->tune.refill_grid;
    hipLaunchKernelGGL(bank_prep_kernel<false>, dim3((unsigned)grid), dim3(256), 0, ctx->upload,
                       (const void*)bank->stage, n, bank->dim, bank->rows8, bank->norm, bank->aux, d_flag, ntiles);
    HIP_TRY(ctx, hipGetLastError());
    bank->n = n;
    bank->n_pad = n_pad;
    // the largest row norm of the new rows is known on the device only: assume the float32-root guard is needed
    // (an election then launches one extra kernel that finds an empty list, ~5 us beside the next distance kernel)
    bank->usq_max = INT32_MAX / 2;
    return FM_OK;
}

extern "C" int fm_upload_fence(fm_ctx* ctx)
{
    if (!ctx) return fail(nullptr, FM_EINVAL, "fm_upload_fence: ctx is NULL");
    if (!ctx->upload) return FM_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_upload, ctx->upload));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_upload, 0));
    return FM_OK;
}

// Everything enqueued on the context -- its own stream and the tail streams the async entry points use.
void fm::sync_all_streams(fm_ctx* ctx)
{
    (void)hipSetDevice(ctx->device);
    if (ctx->upload) (void)hipStreamSynchronize(ctx->upload);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (hipStream_t ts : ctx->tails) if (ts) (void)hipStreamSynchronize(ts);
}

extern "C" int fm_bank_destroy(fm_ctx* ctx, fm_bank* bank)
{
    if (!bank) return FM_OK;
    if (ctx) sync_all_streams(ctx);          // (tail kernels of async calls read the bank's self distances)
    bank_free(bank);
    delete bank;
    return FM_OK;
}

extern "C" int fm_bank_info(const fm_bank* bank, int64_t* n, int* dim, int* kind)
{
    if (!bank) return fail(nullptr, FM_EINVAL, "fm_bank_info: bank is NULL");
    if (n) *n = bank->n;
    if (dim) *dim = bank->dim;
    if (kind) *kind = bank->kind;
    return FM_OK;
}

extern "C" int fm_bank_set_selfdist(fm_ctx* ctx, fm_bank* bank, const double* selfdist)
{
    if (!ctx || !bank) return fail(ctx, FM_EINVAL, "fm_bank_set_selfdist: NULL argument");
    if (bank->n > 0 && !selfdist) return fail(ctx, FM_EINVAL, "fm_bank_set_selfdist: selfdist is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!bank->selfdist) HIP_TRY(ctx, hipMalloc((void**)&bank->selfdist, (size_t)(bank->cap_pad > 0 ? bank->cap_pad : 1) * 8));
    if (bank->n > 0) {
        HIP_TRY(ctx, hipMemcpyAsync(bank->selfdist, selfdist, (size_t)bank->n * 8, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return FM_OK;
}

int fm::check_pair(fm_ctx* ctx, const fm_bank* q, const fm_bank* t, const char* who)
{
    if (!ctx) return fail(nullptr, FM_EINVAL, std::string(who) + ": ctx is NULL");
    if (!q || !t) return fail(ctx, FM_EINVAL, std::string(who) + ": bank is NULL");
    if (q->dim != t->dim) return fail(ctx, FM_EINVAL, std::string(who) + ": query/train dim mismatch");
    if (q->kind != t->kind && q->n > 0 && t->n > 0)      // (an empty bank has no kind of its own)
        return fail(ctx, FM_EINVAL, std::string(who) + ": query/train kind mismatch (one bank is integer-valued, the other is not)");
    return FM_OK;
}
