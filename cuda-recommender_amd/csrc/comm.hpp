// comm.hpp -- thin communicator over RCCL (librccl is dlopen'ed on first use so that a
// single-GPU process never loads it).  One process per GPU; collectives are enqueued on the
// solver's stream, no host synchronisation on the data path.
#pragma once

#include "common.hpp"

namespace mfx { struct LocalGroup; }

struct mfx_comm_s {
    void* nccl = nullptr;  // ncclComm_t (RCCL communicator), or
    mfx::LocalGroup* local = nullptr;  // in-process loopback group (threads of one process, host-staged)
    int rank = 0;
    int nranks = 1;
    int device = 0;
};

namespace mfx {
int comm_unique_id(void* id_out);
int comm_create(mfx_comm_s** out, const void* id, int rank, int nranks, int device);
int comm_create_local(mfx_comm_s** out, int group, int rank, int nranks, int device);
int comm_destroy(mfx_comm_s* c);
// Collective: worst (minimum) status over all ranks; see comm.cpp.
int comm_agree(mfx_comm_s* c, int local_status, int* global_status);
// Releases the ranks waiting for this one after a local failure; the communicator is dead afterwards.
int comm_abort(mfx_comm_s* c);
// In-place sum all-reduce on `st`.
int comm_allreduce_f32(mfx_comm_s* c, float* buf, size_t count, hipStream_t st);
int comm_allreduce_f64(mfx_comm_s* c, double* buf, size_t count, hipStream_t st);
// In-place broadcast of `count` floats from rank `root`.
int comm_broadcast_f32(mfx_comm_s* c, float* buf, size_t count, int root, hipStream_t st);
// Rank r holds fresh values of elements [bounds[r], bounds[r+1]) of the replicated `buf`; afterwards everyone holds
// every block.  One grouped call on RCCL (blocks may differ in size).
int comm_allgather_blocks_f32(mfx_comm_s* c, float* buf, const int64_t* bounds, hipStream_t st);
}  // namespace mfx
