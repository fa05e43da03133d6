// gather_shape.hip -- which SHAPE of the packed-sample gather the cache hierarchy serves fastest (round 6; stand-alone binary).
// 64 looping int16-stereo sources of BASELINE config 2's lengths (48 000 + 977 k frames, 4 bytes per frame), 2 880 512 frames,
// every lane 16 frames of every source, arithmetic taken out (xor), 8 bytes written per frame:
//   A  the engine's shape: a lane owns 16 CONSECUTIVE frames -- four 16-byte loads 64 bytes apart from the next lane's: every load
//      instruction touches 64 separate 64-byte pieces (a quarter of each), ONE Barrett modulo per lane and source
//   B  wave-coalesced: quad q of lane l = frames 256 q + 4 l of the wave's 1 024 -- every load instruction reads 1 KB in one piece,
//      a modulo per quad (or one modulo and three conditional wraps)
//   C  two runs of 8 consecutive frames per lane, 512 frames apart: a load instruction touches 64 halves of 32-byte pieces
// each with 1 or 2 sources' loads in flight before the first is consumed.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#define UB_GLOBAL __attribute__((address_space(1)))
#define UB_CONST __attribute__((address_space(4)))
typedef unsigned int u4v_u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f4v __attribute__((ext_vector_type(4)));
struct Tab { const uint32_t* p; uint32_t len, magic, t0, pad; };
__device__ __forceinline__ uint32_t barrett_mod(uint32_t x, uint32_t len, uint32_t magic) {
    const uint32_t r = x - __umulhi(x, magic) * len;
    return r >= len ? r - len : r;
}
template <int SHAPE, int B>
__global__ __launch_bounds__(256) void k_gather(const Tab* __restrict__ tabs_generic, int k, uint32_t M, float* __restrict__ out) {
    const Tab UB_CONST* tabs = (const Tab UB_CONST*)(const UB_CONST char*)tabs_generic;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t wbase = blockIdx.x * 4096u + wave * 1024u;
    auto frame_of = [&](int q) -> uint32_t {   // first frame of the lane's quad q
        if (SHAPE == 0) return wbase + 16u * lane + 4u * q;
        if (SHAPE == 1) return wbase + 256u * q + 4u * lane;
        return wbase + 512u * (q >> 1) + 8u * lane + 4u * (q & 1);
    };
    uint32_t acc[16];
#pragma unroll
    for (int f = 0; f < 16; ++f) acc[f] = 0u;
    for (int j = 0; j + B <= k; j += B) {
        u4v_u w[B][4];
#pragma unroll
        for (int u = 0; u < B; ++u) {
            const uint32_t len = tabs[j + u].len, magic = tabs[j + u].magic, t0 = tabs[j + u].t0;
            const uint32_t UB_GLOBAL* g = (const uint32_t UB_GLOBAL*)(const UB_GLOBAL char*)tabs[j + u].p;
            if (SHAPE == 0) {
                const uint32_t idx = barrett_mod(t0 + frame_of(0), len, magic);
#pragma unroll
                for (int q = 0; q < 4; ++q) w[u][q] = *(const u4v_u UB_GLOBAL*)(g + idx + 4u * q);
            } else if (SHAPE == 1) {
                uint32_t idx = barrett_mod(t0 + frame_of(0), len, magic);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    w[u][q] = *(const u4v_u UB_GLOBAL*)(g + idx);
                    idx += 256u;
                    idx = min(idx, idx - len);   // (len >= 1 024: one wrap at most)
                }
            } else {
                uint32_t idx = barrett_mod(t0 + frame_of(0), len, magic);
                w[u][0] = *(const u4v_u UB_GLOBAL*)(g + idx);
                w[u][1] = *(const u4v_u UB_GLOBAL*)(g + idx + 4u);
                idx += 512u;
                idx = min(idx, idx - len);
                w[u][2] = *(const u4v_u UB_GLOBAL*)(g + idx);
                w[u][3] = *(const u4v_u UB_GLOBAL*)(g + idx + 4u);
            }
        }
#pragma unroll
        for (int u = 0; u < B; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) { acc[4 * q] ^= w[u][q].x; acc[4 * q + 1] ^= w[u][q].y; acc[4 * q + 2] ^= w[u][q].z; acc[4 * q + 3] ^= w[u][q].w; }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t mm = frame_of(q);
        if (mm + 3 < M) {
            f4v v0, v1;
            v0.x = __uint_as_float(acc[4 * q] & 0x3FFFFFFFu); v0.y = __uint_as_float((acc[4 * q] >> 2) & 0x3FFFFFFFu);
            v0.z = __uint_as_float(acc[4 * q + 1] & 0x3FFFFFFFu); v0.w = __uint_as_float((acc[4 * q + 1] >> 2) & 0x3FFFFFFFu);
            v1.x = __uint_as_float(acc[4 * q + 2] & 0x3FFFFFFFu); v1.y = __uint_as_float((acc[4 * q + 2] >> 2) & 0x3FFFFFFFu);
            v1.z = __uint_as_float(acc[4 * q + 3] & 0x3FFFFFFFu); v1.w = __uint_as_float((acc[4 * q + 3] >> 2) & 0x3FFFFFFFu);
            *(f4v UB_GLOBAL*)((UB_GLOBAL char*)(out + 2 * (size_t)mm)) = v0;
            *(f4v UB_GLOBAL*)((UB_GLOBAL char*)(out + 2 * (size_t)mm + 4)) = v1;
        }
    }
}
__global__ void k_fill(uint32_t* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i * 2654435761u;
}
template <int SHAPE, int B>
static float run(const Tab* d_tabs, int k, uint32_t frames, float* d_out) {
    const uint32_t gx = (frames + 4095u) / 4096u;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k_gather<SHAPE, B>), dim3(gx), dim3(256), 0, 0, d_tabs, k, frames, d_out);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((k_gather<SHAPE, B>), dim3(gx), dim3(256), 0, 0, d_tabs, k, frames, d_out);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms / 50.f;
}
int main() {
    const int k = 64; const uint32_t frames = 2880512;
    std::vector<Tab> tabs(k);
    size_t words = 0;
    for (int j = 0; j < k; ++j) words += ((size_t)(48000 + 977 * j) + 18) & ~(size_t)3;
    uint32_t* d_all; Tab* d_tabs; float* d_out;
    hipMalloc(&d_all, words * 4); hipMalloc(&d_tabs, sizeof(Tab) * k); hipMalloc(&d_out, ((size_t)frames + 4096) * 8);
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, d_all, words);
    size_t off = 0;
    for (int j = 0; j < k; ++j) { const uint32_t len = 48000 + 977 * j; tabs[j] = {d_all + off, len, (uint32_t)(0x100000000ull / len), 0u, 0u}; off += ((size_t)len + 18) & ~(size_t)3; }
    hipMemcpy(d_tabs, tabs.data(), sizeof(Tab) * k, hipMemcpyHostToDevice);
    const double bytes = (double)frames * k * 4.0;
    for (int rep = 0; rep < 2; ++rep) {
        const float a1 = run<0, 1>(d_tabs, k, frames, d_out), a2 = run<0, 2>(d_tabs, k, frames, d_out);
        const float b1 = run<1, 1>(d_tabs, k, frames, d_out), b2 = run<1, 2>(d_tabs, k, frames, d_out);
        const float c1 = run<2, 1>(d_tabs, k, frames, d_out), c2 = run<2, 2>(d_tabs, k, frames, d_out);
        printf("A (16 consecutive per lane)  B=1 %.4f ms %.0f GB/s   B=2 %.4f ms %.0f GB/s\n", a1, bytes / a1 / 1e6, a2, bytes / a2 / 1e6);
        printf("B (wave-coalesced quads)     B=1 %.4f ms %.0f GB/s   B=2 %.4f ms %.0f GB/s\n", b1, bytes / b1 / 1e6, b2, bytes / b2 / 1e6);
        printf("C (2 x 8 consecutive)        B=1 %.4f ms %.0f GB/s   B=2 %.4f ms %.0f GB/s\n", c1, bytes / c1 / 1e6, c2, bytes / c2 / 1e6);
    }
    return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}
