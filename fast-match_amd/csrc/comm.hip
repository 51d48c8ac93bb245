// Result gather over RCCL: the one exchange the hot path needs when independent
// image pairs are sharded over the GPUs of a node -- an all-gather of every rank's accepted
// matches (12-byte rows left on the device by fm_match_accepted_dev) plus their counts.  The
// reference has no distributed code (turntable.py:59 maps the matcher over pairs one after the
// other); this is the data-parallel axis it implies.
//
// RCCL is bound at run time (dlopen "librccl.so.1"): a process that already carries an RCCL --
// PyTorch-ROCm bundles one under the same SONAME -- shares it instead of loading a second copy,
// and a process that never gathers needs no RCCL at all.
#include "ctx_internal.h"
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>
#include <mutex>
#include <string>

namespace fm {

struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId)    GetUniqueId = nullptr;
    decltype(&ncclCommInitRank)   CommInitRank = nullptr;
    decltype(&ncclCommDestroy)    CommDestroy = nullptr;
    decltype(&ncclAllGather)      AllGather = nullptr;
    decltype(&ncclGroupStart)     GroupStart = nullptr;
    decltype(&ncclGroupEnd)       GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

static RcclApi* rccl_api()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
        }
        if (!api.handle) { api.error = std::string("cannot load RCCL: ") + (dlerror() ? dlerror() : "?"); return; }
#define FM_SYM(field, name) api.field = (decltype(api.field))dlsym(api.handle, name); if (!api.field) { api.error = std::string("RCCL symbol missing: ") + name; return; }
        FM_SYM(GetUniqueId, "ncclGetUniqueId")
        FM_SYM(CommInitRank, "ncclCommInitRank")
        FM_SYM(CommDestroy, "ncclCommDestroy")
        FM_SYM(AllGather, "ncclAllGather")
        FM_SYM(GroupStart, "ncclGroupStart")
        FM_SYM(GroupEnd, "ncclGroupEnd")
        FM_SYM(GetErrorString, "ncclGetErrorString")
#undef FM_SYM
    });
    return &api;
}

int comm_unique_id(void* id128, std::string* err)
{
    RcclApi* a = rccl_api();
    if (!a->error.empty()) { *err = a->error; return FM_EUNSUPPORTED; }
    static_assert(sizeof(ncclUniqueId) == 128, "fm_comm_unique_id hands out 128 bytes");
    ncclUniqueId id;
    const ncclResult_t r = a->GetUniqueId(&id);
    if (r != ncclSuccess) { *err = std::string("ncclGetUniqueId: ") + a->GetErrorString(r); return FM_EDEVICE; }
    memcpy(id128, &id, 128);
    return FM_OK;
}

int comm_init(int device, int nranks, int rank, const void* id128, void** comm_out, std::string* err)
{
    RcclApi* a = rccl_api();
    if (!a->error.empty()) { *err = a->error; return FM_EUNSUPPORTED; }
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); *err = "hipSetDevice failed"; return FM_EDEVICE; }
    ncclComm_t comm = nullptr;
    const ncclResult_t r = a->CommInitRank(&comm, nranks, id, rank);
    if (r != ncclSuccess) { *err = std::string("ncclCommInitRank: ") + a->GetErrorString(r); return FM_EDEVICE; }
    *comm_out = (void*)comm;
    return FM_OK;
}

int comm_destroy(void* comm)
{
    RcclApi* a = rccl_api();
    if (comm && a->CommDestroy) (void)a->CommDestroy((ncclComm_t)comm);
    return FM_OK;
}

// One grouped pair of all-gathers on `stream`: counts (1 x int64 per rank; skipped when d_count is null) and
// rows (cap x 3 x int32 per rank; skipped when cap == 0).
int comm_gather(void* comm, const int32_t* d_rows, const int64_t* d_count, int64_t cap,
                int32_t* d_all_rows, int64_t* d_all_counts, hipStream_t stream, std::string* err)
{
    RcclApi* a = rccl_api();
    if (!a->error.empty()) { *err = a->error; return FM_EUNSUPPORTED; }
    ncclResult_t r = a->GroupStart();
    if (r == ncclSuccess && d_count) r = a->AllGather(d_count, d_all_counts, 1, ncclInt64, (ncclComm_t)comm, stream);
    if (r == ncclSuccess && cap > 0) r = a->AllGather(d_rows, d_all_rows, (size_t)cap * 3, ncclInt32, (ncclComm_t)comm, stream);
    const ncclResult_t e = a->GroupEnd();
    if (r == ncclSuccess) r = e;
    if (r != ncclSuccess) { *err = std::string("ncclAllGather: ") + a->GetErrorString(r); return FM_EDEVICE; }
    return FM_OK;
}

}  // namespace fm

using namespace fm;

// ---------------------------------------------------------------------------------------
// result gather over RCCL (comm.hip)
// ---------------------------------------------------------------------------------------
extern "C" int fm_comm_unique_id(void* id128)
{
    if (!id128) return fail(nullptr, FM_EINVAL, "fm_comm_unique_id: NULL buffer");
    std::string err;
    const int rc = comm_unique_id(id128, &err);
    return rc == FM_OK ? FM_OK : fail(nullptr, rc, "fm_comm_unique_id: " + err);
}

extern "C" int fm_comm_init(fm_ctx* ctx, int nranks, int rank, const void* id128)
{
    if (!ctx) return fail(nullptr, FM_EINVAL, "fm_comm_init: ctx is NULL");
    if (!id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(ctx, FM_EINVAL, "fm_comm_init: bad argument");
    if (ctx->comm) return fail(ctx, FM_EINVAL, "fm_comm_init: the context already has a communicator (fm_comm_destroy first)");
    std::string err;
    void* comm = nullptr;
    const int rc = comm_init(ctx->device, nranks, rank, id128, &comm, &err);
    if (rc != FM_OK) return fail(ctx, rc, "fm_comm_init: " + err);
    ctx->comm = comm;
    ctx->comm_ranks = nranks;
    return FM_OK;
}

extern "C" int fm_comm_destroy(fm_ctx* ctx)
{
    if (!ctx) return FM_OK;
    if (ctx->comm) {
        sync_all_streams(ctx);               // (a gather behind an async fill runs on a tail stream)
        comm_destroy(ctx->comm);
        ctx->comm = nullptr;
        ctx->comm_ranks = 0;
    }
    return FM_OK;
}

// Two-phase form: the counts first, then only as many rows per rank as the fullest rank holds (the padded
// form ships cap rows per rank whatever they hold).  Costs a host synchronisation between the phases.
extern "C" int fm_gather_matches_counted(fm_ctx* ctx, const int32_t* d_rows, const int64_t* d_count, int64_t cap,
                                         int32_t* d_all_rows, int64_t* d_all_counts, int64_t* rows_per_rank)
{
    if (!ctx) return fail(nullptr, FM_EINVAL, "fm_gather_matches_counted: ctx is NULL");
    if (!ctx->comm) return fail(ctx, FM_EINVAL, "fm_gather_matches_counted: no communicator (fm_comm_init)");
    if (cap < 0 || !d_count || !d_all_counts || !rows_per_rank || (cap > 0 && (!d_rows || !d_all_rows)))
        return fail(ctx, FM_EINVAL, "fm_gather_matches_counted: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::string err;
    hipStream_t gs = ctx->rows_stream ? ctx->rows_stream : ctx->stream;
    int rc = comm_gather(ctx->comm, nullptr, d_count, 0, nullptr, d_all_counts, gs, &err);
    if (rc != FM_OK) return fail(ctx, rc, "fm_gather_matches_counted: " + err);
    std::vector<int64_t> counts((size_t)ctx->comm_ranks);
    HIP_TRY(ctx, hipMemcpyAsync(counts.data(), d_all_counts, counts.size() * 8, hipMemcpyDeviceToHost, gs));
    HIP_TRY(ctx, hipStreamSynchronize(gs));
    int64_t m = 0;
    for (int64_t c : counts) m = c > m ? c : m;
    if (m > cap) m = cap;
    *rows_per_rank = m;
    if (m > 0) {
        rc = comm_gather(ctx->comm, d_rows, nullptr, m, d_all_rows, nullptr, gs, &err);
        if (rc != FM_OK) return fail(ctx, rc, "fm_gather_matches_counted: " + err);
        HIP_TRY(ctx, hipStreamSynchronize(gs));
    }
    return FM_OK;
}

extern "C" int fm_gather_matches(fm_ctx* ctx, const int32_t* d_rows, const int64_t* d_count, int64_t cap,
                                 int32_t* d_all_rows, int64_t* d_all_counts, int wait)
{
    if (!ctx) return fail(nullptr, FM_EINVAL, "fm_gather_matches: ctx is NULL");
    if (!ctx->comm) return fail(ctx, FM_EINVAL, "fm_gather_matches: no communicator (fm_comm_init)");
    if (cap < 0 || !d_count || !d_all_counts || (cap > 0 && (!d_rows || !d_all_rows)))
        return fail(ctx, FM_EINVAL, "fm_gather_matches: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::string err;
    // on the stream that filled d_rows: behind fm_match_accepted_dev_async that is the tail stream, so
    // the collective does not sit in front of the next pair's K1
    hipStream_t gs = ctx->rows_stream ? ctx->rows_stream : ctx->stream;
    const int rc = comm_gather(ctx->comm, d_rows, d_count, cap, d_all_rows, d_all_counts, gs, &err);
    if (rc != FM_OK) return fail(ctx, rc, "fm_gather_matches: " + err);
    if (wait) HIP_TRY(ctx, hipStreamSynchronize(gs));
    return FM_OK;
}

