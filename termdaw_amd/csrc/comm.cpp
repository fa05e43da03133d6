// comm.cpp -- RCCL behind six symbols (comm.h).
#include "comm.h"

#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>

#include "engine.h"

namespace tde {
namespace {
// rccl.h's ABI for the calls used (NCCL 2.x: stable across the releases ROCm ships)
struct UniqueId { char internal[128]; };
typedef int (*GetUniqueId_t)(UniqueId*);
typedef int (*CommInitRank_t)(void**, int, UniqueId, int);
typedef int (*CommDestroy_t)(void*);
typedef int (*AllReduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*GetErrorString_t)(int);
constexpr int kNcclFloat32 = 7, kNcclMax = 2;   // ncclDataType_t / ncclRedOp_t (rccl.h:448-466)

struct Rccl {
    void* h = nullptr;
    GetUniqueId_t get_id = nullptr;
    CommInitRank_t init = nullptr;
    CommDestroy_t destroy = nullptr;
    AllReduce_t allreduce = nullptr;
    GetErrorString_t errstr = nullptr;
    std::string path, why;
};
Rccl& lib() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        if (getenv("TD_RCCL_DISABLE")) { r.why = "switched off by TD_RCCL_DISABLE"; return; }   // (tests: the callers' behaviour without RCCL)
        const char* env = getenv("TD_RCCL_LIB");
        const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            r.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.h) { r.path = n; break; }
            r.why = dlerror();
        }
        if (!r.h) return;
        r.get_id = (GetUniqueId_t)dlsym(r.h, "ncclGetUniqueId");
        r.init = (CommInitRank_t)dlsym(r.h, "ncclCommInitRank");
        r.destroy = (CommDestroy_t)dlsym(r.h, "ncclCommDestroy");
        r.allreduce = (AllReduce_t)dlsym(r.h, "ncclAllReduce");
        r.errstr = (GetErrorString_t)dlsym(r.h, "ncclGetErrorString");
        if (!(r.get_id && r.init && r.destroy && r.allreduce)) {
            r.why = "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce";
            dlclose(r.h);
            r.h = nullptr;
        }
    });
    return r;
}
int need() {
    Rccl& r = lib();
    if (!r.h) return fail("termdaw_amd: RCCL is not available (" + r.why + "); set TD_RCCL_LIB, or use td_comm_init_host");
    return 1;
}
int check(int rc, const char* what) {
    if (rc == 0) return 1;
    Rccl& r = lib();
    return fail(std::string("RCCL error: ") + (r.errstr ? r.errstr(rc) : "code " + std::to_string(rc)) + " at " + what);
}
}  // namespace

const char* rccl_library_path() { return lib().path.c_str(); }
int rccl_unique_id(void* out128) {
    if (!need()) return 0;
    UniqueId id;
    memset(&id, 0, sizeof id);
    if (!check(lib().get_id(&id), "ncclGetUniqueId")) return 0;
    memcpy(out128, &id, sizeof id);
    return 1;
}
int rccl_init(td_comm* c, const void* id128, int rank, int world) {
    if (!need()) return 0;
    UniqueId id;
    memcpy(&id, id128, sizeof id);
    return check(lib().init(&c->nccl, world, id, rank), "ncclCommInitRank");
}
void rccl_destroy(td_comm* c) {
    if (c->nccl && lib().h) (void)lib().destroy(c->nccl);
    c->nccl = nullptr;
}
int rccl_allreduce_max_f32(td_comm* c, float* d_table, size_t n, hipStream_t s) {
    if (!need()) return 0;
    return check(lib().allreduce(d_table, d_table, n, kNcclFloat32, kNcclMax, c->nccl, s), "ncclAllReduce");
}
}  // namespace tde
