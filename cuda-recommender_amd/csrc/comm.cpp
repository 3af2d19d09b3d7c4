#include "comm.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace mfx {
namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
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
        a.Broadcast = reinterpret_cast<decltype(a.Broadcast)>(sym("ncclBroadcast"));
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

// ---------------------------------------------------------------------------------------------
// Loopback group: the ranks are THREADS of one process (each with its own solver, on the same or on
// different devices).  All-reduce = stream sync, D2H into the rank's slot, rendezvous, sum in rank
// order (deterministic), H2D.  Host-staged and slow by design: it exists so that the sharded solver
// path (local partials -> all-reduce -> global counts) can be run through the real kernels on a
// single-GPU box, where RCCL refuses two ranks on one device.
// ---------------------------------------------------------------------------------------------
struct LocalGroup {
    std::mutex m;
    std::condition_variable cv;
    int nranks = 0;
    int arrived = 0;
    uint64_t generation = 0;
    int members = 0;
    std::vector<std::vector<unsigned char>> slots;
    std::vector<unsigned char> result;
};

namespace {
std::mutex g_groups_mutex;
std::map<int, std::shared_ptr<LocalGroup>> g_groups;

// Blocks until all ranks arrived; the last arriver runs `on_last` while the others still wait.
template <typename F>
void rendezvous(LocalGroup* g, F on_last) {
    std::unique_lock<std::mutex> lk(g->m);
    const uint64_t gen = g->generation;
    if (++g->arrived == g->nranks) {
        on_last();
        g->arrived = 0;
        ++g->generation;
        g->cv.notify_all();
    } else {
        g->cv.wait(lk, [&] { return g->generation != gen; });
    }
}

template <typename T>
int local_allreduce(mfx_comm_s* c, T* buf, size_t count, hipStream_t st) {
    LocalGroup* g = c->local;
    const size_t bytes = count * sizeof(T);
    MFX_HIP(hipStreamSynchronize(st));
    g->slots[c->rank].resize(bytes);
    MFX_HIP(hipMemcpy(g->slots[c->rank].data(), buf, bytes, hipMemcpyDeviceToHost));
    rendezvous(g, [&] {
        g->result.assign(bytes, 0);
        T* acc = reinterpret_cast<T*>(g->result.data());
        for (int r = 0; r < g->nranks; ++r) {  // rank order: reproducible
            const T* src = reinterpret_cast<const T*>(g->slots[r].data());
            for (size_t i = 0; i < count; ++i) acc[i] += src[i];
        }
    });
    MFX_HIP(hipMemcpy(buf, g->result.data(), bytes, hipMemcpyHostToDevice));
    rendezvous(g, [] {});  // nobody overwrites a slot before everyone has read the result
    return MFX_OK;
}
}  // namespace

int comm_create_local(mfx_comm_s** out, int group, int rank, int nranks, int device) {
    MFX_REQUIRE(out && nranks >= 1 && rank >= 0 && rank < nranks, "comm_create_local: bad arguments");
    MFX_TRY(use_device(device));
    std::shared_ptr<LocalGroup> g;
    {
        std::lock_guard<std::mutex> lk(g_groups_mutex);
        std::shared_ptr<LocalGroup>& slot = g_groups[group];
        if (!slot) {
            slot = std::make_shared<LocalGroup>();
            slot->nranks = nranks;
            slot->slots.resize(nranks);
        }
        MFX_REQUIRE(slot->nranks == nranks, "loopback group %d was created with %d ranks", group, slot->nranks);
        ++slot->members;
        g = slot;
    }
    mfx_comm_s* h = new mfx_comm_s;
    h->local = g.get();
    h->rank = rank;
    h->nranks = nranks;
    h->device = device;
    *out = h;
    return MFX_OK;
}

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
    if (c->local) {
        std::lock_guard<std::mutex> lk(g_groups_mutex);
        for (auto it = g_groups.begin(); it != g_groups.end(); ++it) {
            if (it->second.get() == c->local) {
                if (--it->second->members == 0) g_groups.erase(it);
                break;
            }
        }
        delete c;
        return MFX_OK;
    }
    if (c->nccl && api().CommDestroy) (void) api().CommDestroy(static_cast<ncclComm_t>(c->nccl));
    delete c;
    return MFX_OK;
}

int comm_allreduce_f32(mfx_comm_s* c, float* buf, size_t count, hipStream_t st) {
    if (c->local) return local_allreduce<float>(c, buf, count, st);
    MFX_NCCL(api().AllReduce(buf, buf, count, ncclFloat32, ncclSum, static_cast<ncclComm_t>(c->nccl), st));
    return MFX_OK;
}

int comm_allreduce_f64(mfx_comm_s* c, double* buf, size_t count, hipStream_t st) {
    if (c->local) return local_allreduce<double>(c, buf, count, st);
    MFX_NCCL(api().AllReduce(buf, buf, count, ncclFloat64, ncclSum, static_cast<ncclComm_t>(c->nccl), st));
    return MFX_OK;
}

}  // namespace mfx

namespace mfx {
int comm_broadcast_f32(mfx_comm_s* c, float* buf, size_t count, int root, hipStream_t st) {
    if (count == 0) return MFX_OK;
    if (c->local) {
        LocalGroup* g = c->local;
        const size_t bytes = count * sizeof(float);
        MFX_HIP(hipStreamSynchronize(st));
        if (c->rank == root) {
            g->result.resize(bytes);
            MFX_HIP(hipMemcpy(g->result.data(), buf, bytes, hipMemcpyDeviceToHost));
        }
        rendezvous(g, [] {});  // the root's data is in place
        if (c->rank != root) MFX_HIP(hipMemcpy(buf, g->result.data(), bytes, hipMemcpyHostToDevice));
        rendezvous(g, [] {});  // everyone has read it
        return MFX_OK;
    }
    MFX_NCCL(api().Broadcast(buf, buf, count, ncclFloat32, root, static_cast<ncclComm_t>(c->nccl), st));
    return MFX_OK;
}
}  // namespace mfx
