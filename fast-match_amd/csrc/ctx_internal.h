// What the translation units behind the C-ABI share: the context object, error plumbing, workspace and
// device-to-host helpers.  api_ctx.hip (context, options, banks) defines the functions declared here;
// api_match.hip (2-NN, cross-check, batches, rounds), api_expand.hip (K7 glue) and comm.hip (result gather)
// use them.  Internal: nothing here is part of include/fastmatch_hip.h.
#pragma once
#include "fm_internal.h"
#include "expand_pair.h"
#include "round_body_f32.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <vector>
#include <map>
#include <string>

struct fm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_call0 = nullptr, ev_call1 = nullptr, ev_k0 = nullptr, ev_k1 = nullptr;
    std::string err;
    std::string devname;
    // diagnostics on stderr, read from the environment ONCE when the context is created (FM_F32_DEBUG, FM_EXPAND_DEBUG,
    // FM_PARK_PROF); they print, they change nothing that is computed
    bool dbg_f32 = false, dbg_expand = false, dbg_park = false;
    // growable device workspaces
    void*  ws_partial = nullptr; size_t ws_partial_bytes = 0;
    void*  ws_out = nullptr;     size_t ws_out_bytes = 0;
    void*  ws_in = nullptr;      size_t ws_in_bytes = 0;
    // configuration: fm_ctx_set_option (the FM_* environment variables seed it at creation)
    fm::Tuning tune;
    int* d_counters = nullptr;   // device words of the fp16 filter (layout: fm_internal.h, launch_filter)
    int64_t filter_launches = 0;
    unsigned long long* h_scratch = nullptr;   // pinned host words the kernels can write (counts)
    // page-locked staging for results that go to pageable caller memory (d2h below)
    char*  h_stage = nullptr; size_t h_stage_bytes = 0, h_stage_used = 0;
    struct StagedCopy { void* dst; size_t off, bytes; };
    std::vector<StagedCopy> staged;
    // calls enqueued without a synchronisation (fm_match_accepted_async): their events, read at fm_sync
    struct PendingTimer { hipEvent_t c0, c1, k0, k1; bool timed; int64_t pairs; bool call_timed = true; int64_t bytes = 0; };
    int64_t async_calls = 0;
    std::vector<PendingTimer> pending;       // in flight
    std::vector<PendingTimer> timer_pool;    // idle event sets
    // fm_match_accepted_async: K1 launches follow each other on `stream`; the small kernels behind a
    // K1 (election, decode + ratio, compaction) run on `stream_tail` and overlap the NEXT call's K1.
    // Two workspace slots alternate; a slot's tail kernels leave its bound[] and qbest[] arrays in
    // the state the next K1 / election expects, so no fill operations sit between two K1 launches.
    hipStream_t stream_tail = nullptr;      // = tails[0]
    static constexpr int kTails = 3;
    hipStream_t tails[kTails] = {nullptr, nullptr, nullptr};   // fm_match_accepted_batch spreads the pairs' tails over these
    hipStream_t rows_stream = nullptr;      // stream that produced the last device-resident rows (fm_gather_matches follows it)
    hipEvent_t ev_consumer = nullptr;
    hipEvent_t ev_tail_end[3] = {nullptr, nullptr, nullptr};   // one per tail stream (an event re-recorded on another stream
                                                               // before its waiters ran is not a safe handshake)
    struct AsyncSlot {
        void* ws = nullptr; size_t bytes = 0;
        int64_t nq = -1, ncols_alloc = -1, partial_bytes = -1;   // layout the arrays were initialised for
        hipEvent_t tail_done = nullptr, k_done = nullptr;
        bool in_use = false;
    } aslot[2];
    int aslot_next = 0;
    std::vector<AsyncSlot> bslot;           // fm_match_accepted_batch: a ring of kBatchSlots workspaces
    static constexpr int kBatchSlots = 32;  // (two launches of up to 16 pairs in flight; a slot is re-used behind its tail's event)
    int64_t bslot_next = 0;
    // fm_mark / fm_wait: points in the enqueued work a caller can wait for without draining what follows
    static constexpr int kMarks = 8;
    struct Mark { hipEvent_t ev[1 + kTails] = {nullptr, nullptr, nullptr, nullptr}; int64_t id = -1; } marks[kMarks];
    int64_t next_mark = 0;
    void* comm = nullptr;        // RCCL communicator of the result gather (fm_comm_init)
    int   comm_ranks = 0;
    fm_stats stats{};
    int64_t stats_bytes = 0;     // fm_stats_ex::bytes_moved
    bool kernel_timed = false;
    int64_t pending_pairs = 0;
    int64_t pending_bytes = 0;   // algorithmic bytes of the launches in pending_pairs: bank rows read once
    // fm_bank_refill_u8_async / fm_upload_fence: uploads run on a stream of their own beside the kernels
    hipStream_t upload = nullptr;
    hipEvent_t ev_upload = nullptr;
    // fm_self_dist: plans of the triangular sweep by (padded rows, stages per workgroup) -- numbers only, no device memory
    std::map<std::pair<int64_t, int>, fm::TriPlan> tri_plans;
};

namespace fm {
// Algorithmic bytes of a bank in a distance-kernel launch: every row read once (128 B int8, 512 B float32).
static inline int64_t bank_bytes(const fm::Bank* b) { return b ? b->n * (b->kind == FM_BANK_F32 ? 512 : 128) : 0; }
}

namespace fm {      // (internal helpers live in the library's namespace: a host program may have a `fail` of its own)
int fail(fm_ctx* ctx, int code, const std::string& msg);
}

#define HIP_TRY(ctx, expr)                                                                  \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            char _b[512];                                                                   \
            snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                     __FILE__, __LINE__);                                                   \
            (void)hipGetLastError();                                                        \
            return fm::fail(ctx, _e == hipErrorOutOfMemory ? FM_ENOMEM : FM_EDEVICE, _b);       \
        }                                                                                   \
    } while (0)

// Ablation builds only (-DFM_ABLATE, scripts/ablate): FM_ABLATE_KEEP_BOUNDS leaves the bounds of a finished
// run in place (measures what the exact path costs).  The product build always resets them.
static inline bool ablate_keep_bounds()
{
#ifdef FM_ABLATE
    return getenv("FM_ABLATE_KEEP_BOUNDS") != nullptr;
#else
    return false;
#endif
}

namespace fm {
int ws_ensure(fm_ctx* ctx, void** p, size_t* cap, size_t need);
// Device-side alias of a page-locked host buffer (fm_host_alloc / hipHostMalloc), or NULL for pageable memory.
void* pinned_device_alias(const void* host);
// Device -> caller memory on the context's stream (through a copy kernel and, for pageable destinations, the
// context's page-locked staging buffer: moved to the caller by CallScope::finish()).
hipError_t d2h(fm_ctx* ctx, void* dst, const void* src, size_t bytes);
// Account the calls that were enqueued without a synchronisation; the streams must be idle.
int drain_pending(fm_ctx* ctx);
// Everything enqueued on the context -- its own stream and the tail streams the async entry points use.
void sync_all_streams(fm_ctx* ctx);
int check_pair(fm_ctx* ctx, const fm_bank* q, const fm_bank* t, const char* who);
// Planes and scale terms of a (query = reduced, train = output rows) pair of float32 banks for x1_round_f32.
void fill_round_f32(fm::RoundF32* r, const fm::Bank& q, const fm::Bank& t);
// One expansion round's cross-checked 1-NN on the whole GPU (K7's delegated cross-check, api_match.hip).
int round_xcheck_dense(fm_ctx* ctx, const fm::Bank& q, const int32_t* d_rows, int64_t nq, const fm::Bank& t, int64_t t0, int64_t nt,
                       unsigned long long* d_qbest);
}

// Brackets one API call: events for total time, stats accounting after the final sync.
struct CallScope {
    fm_ctx* ctx;
    ~CallScope() { ctx->staged.clear(); ctx->h_stage_used = 0; }
    explicit CallScope(fm_ctx* c) : ctx(c)
    {
        // entries left behind by a call that failed half way point at host memory that is gone
        ctx->staged.clear();
        ctx->h_stage_used = 0;
        ctx->kernel_timed = false;
        ctx->pending_pairs = 0;
        ctx->pending_bytes = 0;
        (void)hipEventRecord(ctx->ev_call0, ctx->stream);
    }
    int finish()
    {
        HIP_TRY(ctx, hipEventRecord(ctx->ev_call1, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        // (async calls still in flight finish on the tail stream: a synchronous call completes them too,
        // as the header promises; they are accounted at the next fm_sync / fm_get_stats)
        if (!ctx->pending.empty()) for (hipStream_t ts : ctx->tails) HIP_TRY(ctx, hipStreamSynchronize(ts));
        for (const auto& c : ctx->staged) memcpy(c.dst, ctx->h_stage + c.off, c.bytes);
        ctx->staged.clear();
        ctx->h_stage_used = 0;
        float ms = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_call0, ctx->ev_call1));
        ctx->stats.total_ms += ms;
        ctx->stats.calls += 1;
        if (ctx->kernel_timed) {
            float kms = 0.f;
            HIP_TRY(ctx, hipEventElapsedTime(&kms, ctx->ev_k0, ctx->ev_k1));
            ctx->stats.kernel_ms += kms;
            ctx->stats.kernel_launches += 1;
            ctx->stats.pairs += ctx->pending_pairs;
            ctx->stats_bytes += ctx->pending_bytes;
        }
        return FM_OK;
    }
};
