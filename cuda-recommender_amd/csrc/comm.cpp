#include "comm.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

namespace mfx {
namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};

RcclApi& api() {
    static RcclApi a;
    static std::once_flag once;
    std::call_once(once, [] {
        // A process that already imported torch has its bundled librccl.so.1 mapped; asking for
        // the SONAME reuses that copy instead of mapping a second RCCL next to it.
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            a.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (a.handle) break;
        }
        if (!a.handle) {
            a.why = std::string("dlopen(librccl) failed: ") + dlerror();
            return;
        }
        auto sym = [&](const char* n) {
            void* p = dlsym(a.handle, n);
            if (!p && a.why.empty()) a.why = std::string("missing RCCL symbol ") + n;
            return p;
        };
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
        a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(sym("ncclAllReduce"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return a;
}

int check_api() {
    RcclApi& a = api();
    if (!a.why.empty()) return fail(MFX_ERR_COMM, "RCCL unavailable: %s", a.why.c_str());
    return MFX_OK;
}

#define MFX_NCCL(call)                                                                              \
    do {                                                                                            \
        ncclResult_t _r = (call);                                                                   \
        if (_r != ncclSuccess)                                                                      \
            return fail(MFX_ERR_COMM, "%s failed: %s", #call, api().GetErrorString ? api().GetErrorString(_r) : "?"); \
    } while (0)

}  // namespace

static_assert(sizeof(ncclUniqueId) == MFX_COMM_ID_BYTES, "unique id size");

int comm_unique_id(void* id_out) {
    MFX_TRY(check_api());
    ncclUniqueId id;
    MFX_NCCL(api().GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return MFX_OK;
}

int comm_create(mfx_comm_s** out, const void* id_bytes, int rank, int nranks, int device) {
    MFX_REQUIRE(out && id_bytes && nranks >= 1 && rank >= 0 && rank < nranks, "comm_create: bad arguments");
    MFX_TRY(check_api());
    MFX_TRY(use_device(device));
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof(id));
    ncclComm_t c = nullptr;
    MFX_NCCL(api().CommInitRank(&c, nranks, id, rank));
    mfx_comm_s* h = new mfx_comm_s;
    h->nccl = c;
    h->rank = rank;
    h->nranks = nranks;
    h->device = device;
    *out = h;
    return MFX_OK;
}

int comm_destroy(mfx_comm_s* c) {
    if (!c) return MFX_OK;
    if (c->nccl && api().CommDestroy) (void) api().CommDestroy(static_cast<ncclComm_t>(c->nccl));
    delete c;
    return MFX_OK;
}

int comm_allreduce_f32(mfx_comm_s* c, float* buf, size_t count, hipStream_t st) {
    MFX_NCCL(api().AllReduce(buf, buf, count, ncclFloat32, ncclSum, static_cast<ncclComm_t>(c->nccl), st));
    return MFX_OK;
}

int comm_allreduce_f64(mfx_comm_s* c, double* buf, size_t count, hipStream_t st) {
    MFX_NCCL(api().AllReduce(buf, buf, count, ncclFloat64, ncclSum, static_cast<ncclComm_t>(c->nccl), st));
    return MFX_OK;
}

}  // namespace mfx
