// comm.h -- the one collective of the render path: an all-reduce(max) of the per-project peak table across the ranks of a job
// (BASELINE config 5).  The reference has no counterpart: State::render (state.rs:563-575) renders one project per
// process; this is what a batch driver running that loop on every GPU of a node ends with.
//
// RCCL is reached through dlopen -- librccl.so.1 as the process already holds it (a host that also runs torch shares ONE RCCL)
// or from the ROCm install -- so the library loads and renders on a machine without it; only td_comm_init fails there.
#pragma once
#include <stddef.h>
#include <hip/hip_runtime.h>

struct td_comm {
    void* nccl = nullptr;        // ncclComm_t (kind 0)
    int rank = 0, world = 1, device = 0;
    int kind = 0;                // 0: RCCL on device memory; 1: the host's own all-reduce on host memory (td_comm_init_host)
    int (*host_allreduce_max)(void* ctx, float* table, size_t n) = nullptr;
    void* host_ctx = nullptr;
};

namespace tde {
int rccl_unique_id(void* out128);                                             // ncclGetUniqueId
int rccl_init(td_comm* c, const void* id128, int rank, int world);            // ncclCommInitRank on the current device
void rccl_destroy(td_comm* c);
int rccl_allreduce_max_f32(td_comm* c, float* d_table, size_t n, hipStream_t s);   // in place, enqueued on s
const char* rccl_library_path();                                              // what dlopen resolved ("" before the first use)
}
