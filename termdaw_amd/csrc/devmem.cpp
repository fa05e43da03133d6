// devmem.cpp -- the per-process memory cache and the debugging aids behind devmem.h.
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "kernels.h"

// ------------------------------------------------------------------------------------------------
// Device and page-locked memory of the engine comes from a per-process cache: a block given back goes on a free list (by
// device, kind and size) and is handed out again to the next request it fits; nothing goes back to the driver until the cache
// is over its budget (TD_ALLOC_CACHE_MB, default 2048; 0: every free is a real free) or td_trim_memory() is called.
// Why: (1) a project's set-up and tear-down cost dozens of hipMalloc / hipFree calls, each a driver round trip; (2) on this
// platform, with more than 8 processes on one GPU, a device or page-locked buffer allocated where another was just freed is
// sometimes read or written THROUGH THE OLD MAPPING by one of the 8 XCDs (tools/ubench/h2d_check.hip shows it with nothing of
// this library linked: an upload after which one eighth of the words are not there; DESIGN.md 7 "One process of 40") --
// addresses whose mapping never changes cannot be stale.  The semantics of the calls stay the driver's: a free waits for the
// device first (hipFree does), so a block is never handed out while work that uses it is in flight.
// TD_DEBUG_SYNC & 8 (a debugging aid, tools/cls_run.sh): every block handed out starts as 0xFF bytes -- what memory last used by
// ANOTHER process may hold -- with 64 KB of the same behind it, so that a kernel reading a byte nobody wrote shows on an idle GPU.
namespace {
struct MemCache {
    struct Block { void* p; size_t size; int device; unsigned kind; };   // kind: 0 device, 1 + flags page-locked
    std::mutex mu;
    std::map<void*, Block> live;
    std::multimap<std::tuple<int, unsigned, size_t>, void*> free_list;
    std::atomic<size_t> cached_bytes{0};
    size_t budget() {
        static const size_t b = (size_t)(getenv("TD_ALLOC_CACHE_MB") ? atoll(getenv("TD_ALLOC_CACHE_MB")) : 2048) << 20;
        return b;
    }
    static bool poison() {
        static const bool on = getenv("TD_DEBUG_SYNC") && (atoi(getenv("TD_DEBUG_SYNC")) & 8);
        return on;
    }
    static size_t round_up(size_t n) {   // few distinct sizes: 4 KB steps below 64 KB, 64 KB steps below 16 MB, 1 MB steps above
        n = std::max<size_t>(n, 1);
        const size_t step = n < ((size_t)64 << 10) ? 4096 : n < ((size_t)16 << 20) ? ((size_t)64 << 10) : ((size_t)1 << 20);
        return (n + step - 1) / step * step;
    }
    hipError_t raw_alloc(void** p, size_t n, unsigned kind) {
        return kind == 0u ? (hipMalloc)(p, n) : (hipHostMalloc)(p, n, kind - 1u);
    }
    void raw_free(void* p, unsigned kind) {
        if (kind == 0u) (void)(hipFree)(p);
        else (void)(hipHostFree)(p);
    }
    hipError_t get(void** out, size_t n, unsigned kind) {
        int device = 0;
        (void)hipGetDevice(&device);
        const size_t guard = (poison() && kind == 0u) ? (size_t)65536 : 0;
        const size_t want = budget() ? round_up(n + guard) : std::max<size_t>(n + guard, 1);   // (cache off: the driver's own sizes)
        void* p = nullptr;
        size_t size = 0;
        {
            std::lock_guard<std::mutex> lk(mu);
            auto it = free_list.lower_bound(std::make_tuple(device, kind, want));
            // (a block up to 1.5 x the request: a 23 MB edge buffer does not end up holding a 16 MB slab's place for good)
            if (it != free_list.end() && std::get<0>(it->first) == device && std::get<1>(it->first) == kind && std::get<2>(it->first) <= want + want / 2) {
                p = it->second;
                size = std::get<2>(it->first);
                free_list.erase(it);
                cached_bytes -= size;
            }
        }
        if (!p) {
            hipError_t e = raw_alloc(&p, want, kind);
            if (e != hipSuccess) {   // out of memory with blocks on the free list: give them back and try once more
                trim(0);
                (void)hipGetLastError();
                e = raw_alloc(&p, want, kind);
            }
            if (e != hipSuccess) return e;
            size = want;
        }
        if (poison() && kind == 0u) { (void)hipMemset(p, 0xFF, size); (void)hipDeviceSynchronize(); }
        std::lock_guard<std::mutex> lk(mu);
        live[p] = Block{p, size, device, kind};
        *out = p;
        return hipSuccess;
    }
    hipError_t put(void* p) {
        if (!p) return hipSuccess;
        Block b{};
        bool ours = false;
        {
            std::lock_guard<std::mutex> lk(mu);
            auto it = live.find(p);
            if (it != live.end()) {
                b = it->second;
                live.erase(it);
                ours = true;
            }
        }
        if (!ours) {   // (not from here: the driver's own calls)
            if ((hipFree)(p) == hipSuccess) return hipSuccess;
            (void)hipGetLastError();
            return (hipHostFree)(p);
        }
        // what hipFree does before it lets go of memory -- on the block's own device: nothing that uses it is still running
        int cur = 0;
        (void)hipGetDevice(&cur);
        if (cur != b.device) (void)hipSetDevice(b.device);
        const hipError_t e = hipDeviceSynchronize();
        if (b.size > budget() || e != hipSuccess) {
            raw_free(p, b.kind);
        } else {
            std::lock_guard<std::mutex> lk(mu);
            free_list.emplace(std::make_tuple(b.device, b.kind, b.size), p);
            cached_bytes += b.size;
        }
        if (cur != b.device) (void)hipSetDevice(cur);
        if (cached_bytes > budget()) trim(budget() / 2);
        return e;
    }
    void trim(size_t keep) {   // largest blocks first
        std::vector<std::pair<void*, std::tuple<int, unsigned, size_t>>> gone;
        {
            std::lock_guard<std::mutex> lk(mu);
            while (cached_bytes > keep && !free_list.empty()) {
                auto it = free_list.begin();
                for (auto j = free_list.begin(); j != free_list.end(); ++j)
                    if (std::get<2>(j->first) > std::get<2>(it->first)) it = j;
                gone.push_back({it->second, it->first});
                cached_bytes -= std::get<2>(it->first);
                free_list.erase(it);
            }
        }
        int cur = 0;
        (void)hipGetDevice(&cur);
        for (auto& g : gone) {
            if (std::get<0>(g.second) != cur) (void)hipSetDevice(std::get<0>(g.second));
            raw_free(g.first, std::get<1>(g.second));
            if (std::get<0>(g.second) != cur) (void)hipSetDevice(cur);
        }
    }
};
MemCache& mem_cache() {
    static MemCache* c = new MemCache;   // (never destroyed: frees may come from static destructors of the host program)
    return *c;
}
}  // namespace

namespace tde {
hipError_t mem_get(void** p, size_t n, unsigned kind) { return mem_cache().get(p, n, kind); }
hipError_t mem_put(void* p) { return mem_cache().put(p); }
void mem_trim() { mem_cache().trim(0); }
size_t mem_cached_bytes() { return mem_cache().cached_bytes; }

int debug_sync() {   // (devmem.h)
    static const int v = getenv("TD_DEBUG_SYNC") ? atoi(getenv("TD_DEBUG_SYNC")) : 0;
    return v;
}
// TD_DEBUG_SYNC & 16: after an upload has been waited for, 512 workgroups read it back and compare (k_debug_verify)
void debug_verify_upload(const char* what, const uint8_t* d, const uint8_t* h, size_t bytes, hipStream_t stream) {
    if (!(debug_sync() & 16) || bytes < 4) return;
    static uint32_t* report = nullptr;    // page-locked, device-visible
    static uint32_t* d_report = nullptr;
    static uint32_t* seg_h = nullptr;
    static uint32_t* seg_d = nullptr;
    static size_t seg_cap = 0;
    if (!report) {
        (void)(hipHostMalloc)((void**)&report, 4096, hipHostMallocMapped | hipHostMallocCoherent);
        (void)hipHostGetDevicePointer((void**)&d_report, report, 0);
    }
    const uint32_t n_words = (uint32_t)(bytes / 4);
    const uint32_t n_seg = (n_words + 63u) / 64u;
    if (n_seg > seg_cap) {
        if (seg_h) (void)(hipHostFree)(seg_h);
        seg_cap = (size_t)n_seg * 2;
        (void)(hipHostMalloc)((void**)&seg_h, seg_cap * 4, hipHostMallocMapped | hipHostMallocCoherent);
        (void)hipHostGetDevicePointer((void**)&seg_d, seg_h, 0);
    }
    const uint32_t* w = (const uint32_t*)h;
    for (uint32_t sg = 0; sg < n_seg; ++sg) {
        uint32_t v = 0;
        for (uint32_t l = 0; l < 64u && sg * 64u + l < n_words; ++l) v += w[sg * 64u + l] * (2u * l + 1u);
        seg_h[sg] = v;
    }
    memset(report, 0, 4096);
    (void)hipStreamSynchronize(stream);
    tdk::launch_debug_verify((const uint32_t*)d, n_words, seg_d, d_report, stream);
    (void)hipStreamSynchronize(stream);
    if (report[0]) {
        fprintf(stderr, "STALE %s bytes %zu: %u of 512 workgroups disagree;", what, bytes, report[0]);
        for (uint32_t k = 0; k < std::min(report[0], 15u); ++k)
            fprintf(stderr, " [wg %u xcc %u seg %u got %08x want %08x]", report[4 + 4 * k], report[5 + 4 * k] & 0xFu, report[6 + 4 * k], report[7 + 4 * k],
                    w[(size_t)report[6 + 4 * k] * 64u]);
        // ... and what a copy engine reads back from the same place (HBM, past the L2s)
        std::vector<uint8_t> back(bytes);
        (void)hipMemcpy(back.data(), d, bytes, hipMemcpyDeviceToHost);
        size_t nd = 0, first = 0;
        for (size_t i = 0; i < bytes; ++i)
            if (back[i] != h[i]) { if (!nd) first = i; ++nd; }
        fprintf(stderr, " | read back by hipMemcpy: %zu bytes differ (first at %zu)\n", nd, first);
    }
}
}  // namespace tde
