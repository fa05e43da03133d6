// ceilings.hip -- measured ceilings that bench.py prices the engine's kernels against (libtd_ubench.so, loaded by
// bench.py into the SAME process as the engine, on the same device, right after the timed region).
//
//  td_ubench_gather      the access pattern of the inlined packed-sample sum (tdk::k_sum16w / k_sum<loop16>,
//                        termdaw_amd/csrc/kernels.hip) with the arithmetic taken out: every lane walks the k looping
//                        sources, per source ONE Barrett modulo and NQ dword-aligned 16-byte gathers of 4 packed
//                        frames each, and writes the same 8 bytes per frame the sum kernel writes.  What is left is
//                        what the cache hierarchy (L2 -> Infinity Cache -> HBM) can deliver for this gather: the
//                        ceiling of a kernel whose bytes are served by caches, where "fraction of HBM peak" is
//                        meaningless.  Run on tables of the workload's own sizes ("same tables") and on tables small
//                        enough to stay in one XCD's L2 ("L2-resident").
//  td_ubench_stream      plain float4 copy-like stream (read r bytes, write w bytes per frame): the box's own HBM
//                        rate for the two-pass normalize's second pass and for the edge-buffer sums.
//  td_ubench_valu_chain  ns per dependent VALU instruction of ONE wave (the band-pass warm-up's floor).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

namespace {

struct Tab {
    const uint32_t* p;
    uint32_t len, magic, t0, pad;
};

#define UB_GLOBAL __attribute__((address_space(1)))
#define UB_CONST __attribute__((address_space(4)))
typedef unsigned int u4v_u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t barrett_mod(uint32_t x, uint32_t len, uint32_t magic) {
    const uint32_t r = x - __umulhi(x, magic) * len;
    return r >= len ? r - len : r;
}

// NQ x 16-byte gathers per source and lane, B sources' gathers issued before the first is consumed
template <int NQ, int B>
__global__ __launch_bounds__(256) void k_gather(const Tab* __restrict__ tabs_generic, int k, uint32_t M, float* __restrict__ out) {
    const Tab UB_CONST* tabs = (const Tab UB_CONST*)(const UB_CONST char*)(tabs_generic + (size_t)blockIdx.y * k);
    out += (size_t)blockIdx.y * 2 * ((size_t)M + 64);
    const uint32_t m = blockIdx.x * (1024u * NQ) + 4u * NQ * threadIdx.x;
    const uint32_t mw = blockIdx.x * (1024u * NQ) + (threadIdx.x >> 6) * 1024u + 4u * (threadIdx.x & 63u);   // (NQ == 4: the lane's first quad)
    uint32_t acc[4 * NQ];
#pragma unroll
    for (int f = 0; f < 4 * NQ; ++f) acc[f] = 0u;
    int j = 0;
    for (; j + B <= k; j += B) {
        u4v_u w[B][NQ];
#pragma unroll
        for (int u = 0; u < B; ++u) {
            const uint32_t len = tabs[j + u].len;
            const uint32_t idx = barrett_mod(tabs[j + u].t0 + m, len, tabs[j + u].magic);
            const uint32_t UB_GLOBAL* g = (const uint32_t UB_GLOBAL*)(const UB_GLOBAL char*)tabs[j + u].p;
            if (NQ == 4) {   // the engine's shape since round 6 (kernels.hip quad_frame): quad q = frames 256 q + 4 lane of the wave's 1 024
                uint32_t i = barrett_mod(tabs[j + u].t0 + mw, len, tabs[j + u].magic);
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    w[u][q] = *(const u4v_u UB_GLOBAL*)(g + i);
                    i += 256u;
                    i = len > 256u ? min(i, i - len) : barrett_mod(i, len, tabs[j + u].magic);
                }
                continue;
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) w[u][q] = *(const u4v_u UB_GLOBAL*)(g + idx + 4u * q);
        }
#pragma unroll
        for (int u = 0; u < B; ++u)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                acc[4 * q + 0] ^= w[u][q].x; acc[4 * q + 1] ^= w[u][q].y;
                acc[4 * q + 2] ^= w[u][q].z; acc[4 * q + 3] ^= w[u][q].w;
            }
    }
    for (; j < k; ++j) {
        const uint32_t len = tabs[j].len;
        const uint32_t idx = barrett_mod(tabs[j].t0 + m, len, tabs[j].magic);
        const uint32_t UB_GLOBAL* g = (const uint32_t UB_GLOBAL*)(const UB_GLOBAL char*)tabs[j].p;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const u4v_u w = *(const u4v_u UB_GLOBAL*)(g + (NQ == 4 ? barrett_mod(tabs[j].t0 + mw + 256u * q, len, tabs[j].magic) : idx + 4u * q));
            acc[4 * q + 0] ^= w.x; acc[4 * q + 1] ^= w.y; acc[4 * q + 2] ^= w.z; acc[4 * q + 3] ^= w.w;
        }
    }
    // the sum kernel's write: 8 bytes per frame (2 x 16 B per 4 frames)
#pragma unroll
    for (int q = 0; q < 2 * NQ; ++q) {
        const uint32_t mm = m + 2u * q;
        if (mm + 1 < M) {
            f4v v;
            v.x = __uint_as_float(acc[2 * q] & 0x3FFFFFFFu);
            v.y = __uint_as_float((acc[2 * q] >> 2) & 0x3FFFFFFFu);
            v.z = __uint_as_float(acc[2 * q + 1] & 0x3FFFFFFFu);
            v.w = __uint_as_float((acc[2 * q + 1] >> 2) & 0x3FFFFFFFu);
            *(f4v UB_GLOBAL*)((UB_GLOBAL char*)(out + 2 * (size_t)mm)) = v;
        }
    }
}

// reads rd float4 streams, writes wr float4 streams (16 B per lane per access, 1 KiB per wave instruction)
__global__ __launch_bounds__(256) void k_stream(const float* __restrict__ in, float* __restrict__ out, uint32_t n4, int rd, int wr,
                                                size_t stride4) {
    const uint32_t i = blockIdx.x * 512u + threadIdx.x;
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        const uint32_t e = i + 256u * h;
        if (e >= n4) continue;
        f4v acc = {0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < rd; ++r) acc += *(const f4v UB_GLOBAL*)((const UB_GLOBAL char*)(in + 4 * ((size_t)e + r * stride4)));
        for (int w = 0; w < wr; ++w) *(f4v UB_GLOBAL*)((UB_GLOBAL char*)(out + 4 * ((size_t)e + w * stride4))) = acc;
    }
}

__global__ void k_chain(float* out, float g, float x0, int n8) {
    float y = out[threadIdx.x], x = x0 + threadIdx.x, t;
#pragma unroll 1
    for (int i = 0; i < n8; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            asm volatile("v_sub_f32_dpp %1, %2, %0 quad_perm:[0,1,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mul_f32 %1, %3, %1\n v_add_f32 %0, %0, %1"
                         : "+v"(y), "=&v"(t) : "v"(x), "v"(g));
    }
    out[threadIdx.x] = y;
}

// eight independent v_fma_f32 streams per lane: the issue rate of the fastest VALU class (tools/ubench/issue_rate.hip has the rest)
__global__ __launch_bounds__(256) void k_fma_issue(float* out, float g, int n) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
#pragma unroll 1
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                         " v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(g));
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

__global__ void k_fill(uint32_t* p, size_t n, uint32_t salt) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + salt;
        x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
        p[i] = x;
    }
}

// `per` table sets per launch (0: all n_sets in one grid), the rest in further launches -- the engine's batch slicing
template <int NQ, int B>
float time_gather(const Tab* d_tabs, int k, uint32_t frames, float* d_out, int iters, int n_sets, int per) {
    const uint32_t gx = (frames + 1024u * NQ - 1) / (1024u * NQ);
    if (per <= 0 || per > n_sets) per = n_sets;
    auto pass = [&]() {
        for (int o = 0; o < n_sets; o += per)
            hipLaunchKernelGGL((k_gather<NQ, B>), dim3(gx, std::min(per, n_sets - o)), dim3(256), 0, 0, d_tabs + (size_t)o * k, k, frames,
                               d_out + (size_t)o * 2 * ((size_t)frames + 64));
    };
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) pass();
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) pass();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return ms / (float)iters;
}

}  // namespace

extern "C" {

// lens[k]: loop lengths in frames (= packed 32-bit words) of ONE project; n_sets projects (each with its own tables of
// those lengths, `per` of them per launch like a batch submission; 0 = all in one grid).  frames: timeline length.  nq: 1 | 2 | 4 (4 * nq consecutive
// frames per lane, as the engine's k_sum<loop16> / k_sum16w<2> / k_sum16w<4>).  Returns the best (smallest) average
// ms per launch over the issue variants (1, 2 or 4 sources' gathers in flight before the first use), < 0 on failure.
float td_ubench_gather(const uint32_t* lens, int k, uint32_t frames, int nq, int iters, int n_sets, int per) {
    if (k <= 0 || frames == 0 || iters <= 0 || n_sets <= 0) return -1.f;
    std::vector<Tab> tabs((size_t)k * n_sets);
    size_t words_set = 0;
    for (int j = 0; j < k; ++j) words_set += ((size_t)lens[j] + 18) & ~(size_t)3;
    uint32_t* d_all = nullptr;
    Tab* d_tabs = nullptr;
    float* d_out = nullptr;
    float best = -1.f;
    if (hipMalloc(&d_all, words_set * 4 * (size_t)n_sets) == hipSuccess &&
        hipMalloc(&d_tabs, sizeof(Tab) * tabs.size()) == hipSuccess &&
        hipMalloc(&d_out, ((size_t)frames + 64) * 2 * sizeof(float) * (size_t)n_sets) == hipSuccess) {
        hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, d_all, words_set * (size_t)n_sets, 12345u);
        size_t off = 0;
        for (int s = 0; s < n_sets; ++s)
            for (int j = 0; j < k; ++j) {
                const uint32_t len = lens[j];
                tabs[(size_t)s * k + j] = {d_all + off, len, len >= 2 ? (uint32_t)(0x100000000ull / len) : 0xFFFFFFFFu, 0u, 0u};
                off += ((size_t)len + 18) & ~(size_t)3;
            }
        if (hipMemcpy(d_tabs, tabs.data(), sizeof(Tab) * tabs.size(), hipMemcpyHostToDevice) == hipSuccess) {
            float t[3] = {0, 0, 0};
            if (nq == 4) { t[0] = time_gather<4, 1>(d_tabs, k, frames, d_out, iters, n_sets, per); t[1] = time_gather<4, 2>(d_tabs, k, frames, d_out, iters, n_sets, per); t[2] = time_gather<4, 4>(d_tabs, k, frames, d_out, iters, n_sets, per); }
            else if (nq == 2) { t[0] = time_gather<2, 1>(d_tabs, k, frames, d_out, iters, n_sets, per); t[1] = time_gather<2, 2>(d_tabs, k, frames, d_out, iters, n_sets, per); t[2] = time_gather<2, 4>(d_tabs, k, frames, d_out, iters, n_sets, per); }
            else { t[0] = time_gather<1, 1>(d_tabs, k, frames, d_out, iters, n_sets, per); t[1] = time_gather<1, 2>(d_tabs, k, frames, d_out, iters, n_sets, per); t[2] = time_gather<1, 4>(d_tabs, k, frames, d_out, iters, n_sets, per); }
            if (hipDeviceSynchronize() == hipSuccess && hipGetLastError() == hipSuccess) best = std::min(t[0], std::min(t[1], t[2]));
        }
    }
    if (d_tabs) (void)hipFree(d_tabs);
    if (d_out) (void)hipFree(d_out);
    if (d_all) (void)hipFree(d_all);
    return best;
}

// `frames` stereo frames (8 B each): reads rd and writes wr such streams; average ms per launch
float td_ubench_stream(uint32_t frames, int rd, int wr, int iters) {
    if (!frames || iters <= 0 || rd < 0 || wr < 0 || rd + wr == 0) return -1.f;
    const uint32_t n4 = (frames + 1) / 2;   // float4 = 2 frames
    const size_t stride4 = ((size_t)n4 + 255) & ~(size_t)255;
    float *in = nullptr, *out = nullptr;
    float ms = -1.f;
    if (hipMalloc(&in, stride4 * 16 * (size_t)std::max(rd, 1)) == hipSuccess &&
        hipMalloc(&out, stride4 * 16 * (size_t)std::max(wr, 1)) == hipSuccess) {
        (void)hipMemset(in, 0x3c, stride4 * 16 * (size_t)std::max(rd, 1));
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        const uint32_t grid = (n4 + 511) / 512;
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, 0, in, out, n4, rd, wr, stride4);
        hipEventRecord(e0, 0);
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, 0, in, out, n4, rd, wr, stride4);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess && hipGetLastError() == hipSuccess) ms /= (float)iters;
        else ms = -1.f;
        hipEventDestroy(e0);
        hipEventDestroy(e1);
    }
    if (in) (void)hipFree(in);
    if (out) (void)hipFree(out);
    return ms;
}

// ns per dependent VALU instruction of one wave alone: the band-pass recurrence's three-instruction step (DPP subtract,
// multiply, add) as the speculative warm-up of k_band_spec executes it
float td_ubench_valu_chain_ns(void) {
    float* out = nullptr;
    if (hipMalloc(&out, 256) != hipSuccess) return -1.f;
    (void)hipMemset(out, 0, 256);
    const int n8 = 1 << 16;   // 524 288 steps
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = -1.f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, out, 0.01f, 1.0f, n8);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float t = 0.f;
        if (hipEventElapsedTime(&t, e0, e1) == hipSuccess) ms = ms < 0.f ? t : std::min(ms, t);
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    (void)hipFree(out);
    return ms < 0.f ? -1.f : ms * 1e6f / (float)(n8 * 8 * 3);
}

// ns per wave-level v_fma_f32 per SIMD with 8 waves per SIMD on every CU (n_cu compute units)
float td_ubench_fma_issue_ns(int n_cu) {
    if (n_cu <= 0) return -1.f;
    const int waves = 8, n = 1000, blocks = n_cu * waves;   // one 256-thread block = one wave on each of a CU's four SIMDs
    float* out = nullptr;
    if (hipMalloc(&out, (size_t)blocks * 256 * sizeof(float)) != hipSuccess) return -1.f;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = -1.f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_fma_issue, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, n);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float t = 0.f;
        if (hipEventElapsedTime(&t, e0, e1) == hipSuccess) ms = ms < 0.f ? t : std::min(ms, t);
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    (void)hipFree(out);
    return ms < 0.f ? -1.f : ms * 1e6f / (64.0f * n * waves);
}

}  // extern "C"
