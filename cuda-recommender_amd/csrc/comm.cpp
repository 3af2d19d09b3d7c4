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
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};

RcclApi& api() {
    static RcclApi a;
    static std::once_flag once;
    std::call_once(once, [] {
        // One RCCL per process, like one HIP runtime (mfx/_lib.py): PyTorch-ROCm bundles its own
        // librccl under /opt/rocm's SONAME.  A process that already imported torch has that copy mapped
        // and asking for the SONAME reuses it; a process that will import torch LATER must get the same
        // file now -- its path arrives in MFX_RCCL_PATH (set by the Python binding when torch is installed)
        // -- or torch would map a second RCCL next to ours (seen: heap corruption at interpreter exit).
        // RTLD_LOCAL: the symbols are looked up with dlsym, nothing else in the process should bind to them.
        const char* hint = getenv("MFX_RCCL_PATH");
        for (const char* name : {hint, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (!name || !*name) continue;
            a.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
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
        a.CommAbort = reinterpret_cast<decltype(a.CommAbort)>(dlsym(a.handle, "ncclCommAbort"));  // optional
        a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(sym("ncclAllReduce"));
        a.Broadcast = reinterpret_cast<decltype(a.Broadcast)>(sym("ncclBroadcast"));
        a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(sym("ncclGroupStart"));
        a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(sym("ncclGroupEnd"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return a;
}

int check_api() {
    RcclApi& a = api();
    if (!a.why.empty()) return fail(MFX_ERR_COMM, "RCCL unavailable: %s", a.why.c_str());
    return MFX_OK;
}

// RCCL probes peers / devices with HIP calls that are allowed to fail and leaves their error code in the
// runtime's sticky per-thread slot ("invalid device ordinal" while enumerating peers).  That slot is
// what hipGetLastError() after OUR next kernel launch would report, so it is dropped HERE, right after
// the RCCL call that succeeded -- and nowhere else: a launch failure of libmfx's own is never swallowed.
#define MFX_NCCL(call)                                                                              \
    do {                                                                                            \
        ncclResult_t _r = (call);                                                                   \
        if (_r != ncclSuccess)                                                                      \
            return fail(MFX_ERR_COMM, "%s failed: %s", #call, api().GetErrorString ? api().GetErrorString(_r) : "?"); \
        (void) hipGetLastError();                                                                   \
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
    bool aborted = false;  // set by a rank that failed: every present and future waiter gives up
    std::vector<std::vector<unsigned char>> slots;
    std::vector<unsigned char> result;
};

namespace {
std::mutex g_groups_mutex;
std::map<int, std::shared_ptr<LocalGroup>> g_groups;

// RCCL communicators of this process, by the unique id they were created from: mfx_comm_abort on one
// of them aborts every sibling too, which is what releases their collectives' kernels on the GPU.
struct RcclPeer { std::string id; mfx_comm_s* comm; };
std::mutex g_peers_mutex;
std::vector<RcclPeer> g_peers;

// Blocks until all ranks arrived; the last arriver runs `on_last` while the others still wait.
// False when the group was aborted (mfx_comm_abort by a rank that failed): the collective did not
// happen and the caller must unwind.
template <typename F>
bool rendezvous(LocalGroup* g, F on_last) {
    std::unique_lock<std::mutex> lk(g->m);
    if (g->aborted) return false;
    const uint64_t gen = g->generation;
    if (++g->arrived == g->nranks) {
        on_last();
        g->arrived = 0;
        ++g->generation;
        g->cv.notify_all();
    } else {
        g->cv.wait(lk, [&] { return g->generation != gen || g->aborted; });
        if (g->generation == gen) return false;  // woken by the abort, not by the last arriver
    }
    return true;
}

void abort_group(LocalGroup* g) {
    std::lock_guard<std::mutex> lk(g->m);
    g->aborted = true;
    g->cv.notify_all();
}

int aborted_error() { return fail(MFX_ERR_COMM, "loopback communicator was aborted by a failing rank"); }

// A local failure inside a loopback collective must not leave the other ranks waiting for this one.
#define MFX_HIP_OR_ABORT(g, call)                                                            \
    do {                                                                                     \
        hipError_t _e = (call);                                                              \
        if (_e != hipSuccess) {                                                              \
            abort_group(g);                                                                  \
            return ::mfx::fail(MFX_ERR_HIP, "%s failed: %s (%s:%d)", #call,                  \
                               hipGetErrorString(_e), __FILE__, __LINE__);                   \
        }                                                                                    \
    } while (0)

template <typename T>
int local_allreduce(mfx_comm_s* c, T* buf, size_t count, hipStream_t st) {
    LocalGroup* g = c->local;
    const size_t bytes = count * sizeof(T);
    MFX_HIP_OR_ABORT(g, hipStreamSynchronize(st));
    g->slots[c->rank].resize(bytes);
    MFX_HIP_OR_ABORT(g, hipMemcpy(g->slots[c->rank].data(), buf, bytes, hipMemcpyDeviceToHost));
    if (!rendezvous(g, [&] {
            g->result.assign(bytes, 0);
            T* acc = reinterpret_cast<T*>(g->result.data());
            for (int r = 0; r < g->nranks; ++r) {  // rank order: reproducible
                const T* src = reinterpret_cast<const T*>(g->slots[r].data());
                for (size_t i = 0; i < count; ++i) acc[i] += src[i];
            }
        }))
        return aborted_error();
    MFX_HIP_OR_ABORT(g, hipMemcpy(buf, g->result.data(), bytes, hipMemcpyHostToDevice));
    if (!rendezvous(g, [] {})) return aborted_error();  // nobody overwrites a slot before everyone has read the result
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
    {
        std::lock_guard<std::mutex> lk(g_peers_mutex);
        g_peers.push_back({std::string(static_cast<const char*>(id_bytes), sizeof(id)), h});
    }
    *out = h;
    return MFX_OK;
}

// Worst status over all ranks (statuses are <= 0, so the minimum).  Every rank calls this once after
// its own setup, successful or not, BEFORE the first data-path collective: a rank whose shard
// extraction / solver creation failed tells the others instead of leaving them inside an all-reduce
// that it will never join.  On RCCL it doubles as the warm-up collective (RCCL sets up its
// connections lazily, inside the first one).
int comm_agree(mfx_comm_s* c, int local_status, int* global_status) {
    MFX_REQUIRE(c && global_status, "comm_agree: null argument");
    *global_status = local_status;
    if (c->local) {
        LocalGroup* g = c->local;
        {
            std::lock_guard<std::mutex> lk(g->m);
            if (g->slots[c->rank].size() < sizeof(int)) g->slots[c->rank].resize(sizeof(int));
            memcpy(g->slots[c->rank].data(), &local_status, sizeof(int));
        }
        int worst = 0;
        if (!rendezvous(g, [&] {
                g->result.assign(sizeof(int), 0);
                int w = 0;
                for (int r = 0; r < g->nranks; ++r) {
                    int s = 0;
                    memcpy(&s, g->slots[r].data(), sizeof(int));
                    if (s < w) w = s;
                }
                memcpy(g->result.data(), &w, sizeof(int));
            }))
            return aborted_error();
        memcpy(&worst, g->result.data(), sizeof(int));
        if (!rendezvous(g, [] {})) return aborted_error();
        *global_status = worst;
        return MFX_OK;
    }
    MFX_TRY(use_device(c->device));
    DevBuf<int> d;
    MFX_TRY(d.alloc(1));
    MFX_HIP(hipMemcpy(d.get(), &local_status, sizeof(int), hipMemcpyHostToDevice));
    MFX_NCCL(api().AllReduce(d.get(), d.get(), 1, ncclInt32, ncclMin, static_cast<ncclComm_t>(c->nccl), nullptr));
    MFX_HIP(hipStreamSynchronize(nullptr));
    MFX_HIP(hipMemcpy(global_status, d.get(), sizeof(int), hipMemcpyDeviceToHost));
    return MFX_OK;
}

// Called by a rank that failed after the collectives started: releases the ranks waiting for it
// (loopback: wakes the rendezvous; RCCL: ncclCommAbort on every communicator of the same unique id
// in this process, which ends their collectives' kernels).  The communicator is unusable afterwards.
int comm_abort(mfx_comm_s* c) {
    if (!c) return MFX_OK;
    if (c->local) { abort_group(c->local); return MFX_OK; }
    if (!api().CommAbort) return fail(MFX_ERR_COMM, "this RCCL has no ncclCommAbort");
    std::vector<mfx_comm_s*> group;
    {
        std::lock_guard<std::mutex> lk(g_peers_mutex);
        std::string id;
        for (const RcclPeer& p : g_peers) if (p.comm == c) id = p.id;
        for (RcclPeer& p : g_peers) if (p.id == id && p.comm->nccl) group.push_back(p.comm);
    }
    for (mfx_comm_s* p : group) {
        (void) api().CommAbort(static_cast<ncclComm_t>(p->nccl));
        p->nccl = nullptr;  // aborted communicators are already released
    }
    (void) hipGetLastError();
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
    {
        std::lock_guard<std::mutex> lk(g_peers_mutex);
        for (size_t i = 0; i < g_peers.size(); ++i)
            if (g_peers[i].comm == c) { g_peers.erase(g_peers.begin() + (long) i); break; }
    }
    if (c->nccl && api().CommDestroy) {
        (void) api().CommDestroy(static_cast<ncclComm_t>(c->nccl));
        (void) hipGetLastError();
    }
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
        MFX_HIP_OR_ABORT(g, hipStreamSynchronize(st));
        if (c->rank == root) {
            g->result.resize(bytes);
            MFX_HIP_OR_ABORT(g, hipMemcpy(g->result.data(), buf, bytes, hipMemcpyDeviceToHost));
        }
        if (!rendezvous(g, [] {})) return aborted_error();  // the root's data is in place
        if (c->rank != root) MFX_HIP_OR_ABORT(g, hipMemcpy(buf, g->result.data(), bytes, hipMemcpyHostToDevice));
        if (!rendezvous(g, [] {})) return aborted_error();  // everyone has read it
        return MFX_OK;
    }
    MFX_NCCL(api().Broadcast(buf, buf, count, ncclFloat32, root, static_cast<ncclComm_t>(c->nccl), st));
    return MFX_OK;
}

// Every rank r owns elements [bounds[r], bounds[r+1]) of the replicated buffer `buf` and holds fresh values
// there; afterwards every rank holds all blocks.  RCCL: the nranks broadcasts are ONE group call (one launch,
// all roots progress together -- not nranks serial collectives); blocks may have different sizes, which is why
// this is not an ncclAllGather.  Loopback: every rank stages its block, one rendezvous, everyone reads.
int comm_allgather_blocks_f32(mfx_comm_s* c, float* buf, const int64_t* bounds, hipStream_t st) {
    if (c->local) {
        LocalGroup* g = c->local;
        MFX_HIP_OR_ABORT(g, hipStreamSynchronize(st));
        const size_t mine = (size_t) (bounds[c->rank + 1] - bounds[c->rank]) * sizeof(float);
        g->slots[c->rank].resize(mine);
        if (mine) MFX_HIP_OR_ABORT(g, hipMemcpy(g->slots[c->rank].data(), buf + bounds[c->rank], mine, hipMemcpyDeviceToHost));
        if (!rendezvous(g, [] {})) return aborted_error();  // every block is staged
        for (int r = 0; r < g->nranks; ++r) {
            const size_t bytes = (size_t) (bounds[r + 1] - bounds[r]) * sizeof(float);
            if (r == c->rank || bytes == 0) continue;
            if (g->slots[r].size() != bytes) { abort_group(g); return fail(MFX_ERR_COMM, "loopback all-gather: rank %d staged %zu bytes, expected %zu", r, g->slots[r].size(), bytes); }
            MFX_HIP_OR_ABORT(g, hipMemcpy(buf + bounds[r], g->slots[r].data(), bytes, hipMemcpyHostToDevice));
        }
        if (!rendezvous(g, [] {})) return aborted_error();  // nobody restages before everyone has read
        return MFX_OK;
    }
    MFX_NCCL(api().GroupStart());
    for (int r = 0; r < c->nranks; ++r) {
        const size_t cnt = (size_t) (bounds[r + 1] - bounds[r]);
        if (cnt == 0) continue;
        const ncclResult_t rc = api().Broadcast(buf + bounds[r], buf + bounds[r], cnt, ncclFloat32, r, static_cast<ncclComm_t>(c->nccl), st);
        if (rc != ncclSuccess) {
            (void) api().GroupEnd();
            return fail(MFX_ERR_COMM, "ncclBroadcast (grouped, root %d) failed: %s", r, api().GetErrorString ? api().GetErrorString(rc) : "?");
        }
    }
    MFX_NCCL(api().GroupEnd());
    return MFX_OK;
}
}  // namespace mfx
