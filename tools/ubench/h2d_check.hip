// Does an upload arrive?  (DESIGN.md 7 "One process of 40".)  N copies of this program on ONE GPU, each doing what the engine's
// table upload does: wait for the previous copy's event, overwrite ONE page-locked buffer, hipMemcpyAsync it into ONE device
// buffer, launch a kernel on the same stream that checks every word against the pattern of this iteration (a hash of
// iteration and index, computed in the kernel), and every few iterations read the buffer back with hipMemcpy as well.  Nothing of
// termdaw_amd is linked.   hipcc -O2 --offload-arch=gfx950 h2d_check.hip -o h2d_check ; ./h2d_check <seconds> <seed> [coherent|churn|coherent+churn]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

__host__ __device__ static inline uint32_t pat(uint32_t seed, uint32_t it, uint32_t i) {
    uint32_t x = seed * 0x9E3779B9u ^ (it * 0x85EBCA6Bu) ^ (i * 0xC2B2AE35u);
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
__global__ void k_check(const uint32_t* __restrict__ d, uint32_t n, uint32_t seed, uint32_t it, uint32_t* __restrict__ rep) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t v = d[i], w = pat(seed, it, i);
        if (v != w) {
            const uint32_t k = atomicAdd(rep, 1u);
            if (k == 0u) { rep[1] = i; rep[2] = v; rep[3] = w; rep[4] = it; }
            atomicMin(rep + 5, i);
            atomicMax(rep + 6, i);
        }
    }
}
__global__ void k_busy(float* p, int n) {   // a little unrelated work between uploads
    float a = p[threadIdx.x];
    for (int i = 0; i < n; ++i) a = a * 1.0001f + 0.5f;
    p[threadIdx.x] = a;
}
int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 30.0;
    const uint32_t seed = argc > 2 ? (uint32_t)atoi(argv[2]) : 1u;
    const bool coherent = argc > 3 && strstr(argv[3], "coherent");
    const bool churn = argc > 3 && strstr(argv[3], "churn");   // the buffers are freed and allocated again every few uploads, as a new project's are
    const size_t cap = 1u << 20;   // bytes
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    uint32_t *h = nullptr, *d = nullptr, *rep = nullptr, *d_rep = nullptr;
    float* d_busy = nullptr;
    CK(hipHostMalloc((void**)&h, cap, coherent ? hipHostMallocCoherent : hipHostMallocDefault));
    CK(hipMalloc(&d, cap));
    CK(hipMalloc(&d_busy, 4096));
    CK(hipMemset(d_busy, 0, 4096));
    CK(hipHostMalloc((void**)&rep, 4096, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostGetDevicePointer((void**)&d_rep, rep, 0));
    hipEvent_t copied;
    CK(hipEventCreateWithFlags(&copied, hipEventDisableTiming));
    std::vector<uint32_t> back(cap / 4);
    uint32_t rng = seed * 2654435761u + 12345u;
    auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 17; rng ^= rng << 5; return rng; };
    const auto t0 = std::chrono::steady_clock::now();
    uint32_t it = 0;
    unsigned long long uploads = 0, kernel_bad = 0, readback_bad = 0, reallocs = 0;
    bool inflight = false;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        ++it;
        const uint32_t n = 256u + next() % (uint32_t)(cap / 4 - 256u);   // words: 1 KB .. 1 MB
        if (inflight) CK(hipEventSynchronize(copied));
        if (churn && next() % 8u == 0u) {
            CK(hipStreamSynchronize(st));
            CK(hipHostFree(h));
            CK(hipFree(d));
            const size_t c2 = cap + (next() % 64u) * 4096u;
            CK(hipMalloc(&d, c2));
            CK(hipHostMalloc((void**)&h, c2, coherent ? hipHostMallocCoherent : hipHostMallocDefault));
            ++reallocs;
        }
        for (uint32_t i = 0; i < n; ++i) h[i] = pat(seed, it, i);
        memset(rep, 0, 64);
        rep[5] = 0xFFFFFFFFu;
        CK(hipMemcpyAsync(d, h, (size_t)n * 4, hipMemcpyHostToDevice, st));
        CK(hipEventRecord(copied, st));
        inflight = true;
        hipLaunchKernelGGL(k_check, dim3(512), dim3(256), 0, st, d, n, seed, it, d_rep);
        hipLaunchKernelGGL(k_busy, dim3(64), dim3(256), 0, st, d_busy, 2000);
        ++uploads;
        if (it % 4u == 0u) {
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(back.data(), d, (size_t)n * 4, hipMemcpyDeviceToHost));   // (back: always the full 1 MB)
            size_t nb = 0, first = 0;
            for (uint32_t i = 0; i < n; ++i)
                if (back[i] != pat(seed, it, i)) { if (!nb) first = i; ++nb; }
            if (nb) {
                ++readback_bad;
                if (readback_bad < 4) printf("READBACK seed %u it %u words %u: %zu differ, first at word %zu (byte %zu); there: %08x, this iteration: %08x, last iteration: %08x\n",
                                             seed, it, n, nb, first, first * 4, back[first], pat(seed, it, (uint32_t)first), pat(seed, it - 1, (uint32_t)first));
            }
        } else {
            CK(hipStreamSynchronize(st));
        }
        if (rep[0]) {
            ++kernel_bad;
            if (kernel_bad < 4) printf("KERNEL seed %u it %u words %u: %u differ, words %u .. %u; first seen: word %u = %08x, this iteration: %08x, last iteration: %08x\n",
                                       seed, rep[4], n, rep[0], rep[5], rep[6], rep[1], rep[2], rep[3], pat(seed, it - 1, rep[1]));
        }
    }
    printf("seed %u: %llu uploads, %llu seen wrong by the kernel, %llu by the read-back (%llu reallocations)\n", seed, uploads, kernel_bad, readback_bad, reallocs);
    return (kernel_bad || readback_bad) ? 1 : 0;
}
