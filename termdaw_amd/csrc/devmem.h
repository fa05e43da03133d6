// devmem.h -- where the engine's device and page-locked memory comes from (a per-process cache: devmem.cpp), and the debugging
// aids of tools/cls_run.sh (environment TD_DEBUG_SYNC).  A file that includes this header allocates through the cache: the four
// driver calls are redirected below (same signatures, same meaning -- a free waits for the device first).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

namespace tde {
hipError_t mem_get(void** p, size_t n, unsigned kind);   // kind 0: device memory; 1 + flags: page-locked (hipHostMalloc flags)
hipError_t mem_put(void* p);
void mem_trim();                 // everything on the free lists goes back to the driver
size_t mem_cached_bytes();
template <class T>
inline hipError_t cached_malloc(T** p, size_t n) { return mem_get((void**)p, n, 0u); }
template <class T>
inline hipError_t cached_host_malloc(T** p, size_t n, unsigned flags) { return mem_get((void**)p, n, 1u + flags); }

// TD_DEBUG_SYNC (environment; all off by default): 1 = the stream is drained in front of every upload, 2 = every upload is
// waited for, 4 = every launch is announced on stderr and waited for (the last line of a process that died names the kernel),
// 8 = every block handed out starts as 0xFF bytes with 64 KB of the same behind it, 16 = every upload is read back by 512
// workgroups and by hipMemcpy and compared with the page-locked source (a line starting STALE on stderr when they disagree)
int debug_sync();
void debug_verify_upload(const char* what, const uint8_t* d, const uint8_t* h, size_t bytes, hipStream_t stream);
}  // namespace tde

#define hipMalloc(p, n) tde::cached_malloc((p), (n))
#define hipFree(p) tde::mem_put((void*)(p))
#define hipHostMalloc(p, n, flags) tde::cached_host_malloc((p), (n), (flags))
#define hipHostFree(p) tde::mem_put((void*)(p))
