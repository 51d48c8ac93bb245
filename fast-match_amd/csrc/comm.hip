// Result gather over RCCL: the one exchange the hot path needs when independent
// image pairs are sharded over the GPUs of a node -- an all-gather of every rank's accepted
// matches (12-byte rows left on the device by fm_match_accepted_dev) plus their counts.  The
// reference has no distributed code (turntable.py:59 maps the matcher over pairs one after the
// other); this is the data-parallel axis it implies.
//
// RCCL is bound at run time (dlopen "librccl.so.1"): a process that already carries an RCCL --
// PyTorch-ROCm bundles one under the same SONAME -- shares it instead of loading a second copy,
// and a process that never gathers needs no RCCL at all.
#include "fm_internal.h"
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
