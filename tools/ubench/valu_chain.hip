// Micro-benchmark: cycles per step of the band-pass recurrence y += g * (x - y) in a single wave, for
// several instruction forms (dependent-issue latency of plain / DPP VALU on gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N (1 << 20)
template <int V>
__global__ void k(float* out, uint64_t* cyc, float g, float x0) {
    float y = out[threadIdx.x];
    float x = x0 + threadIdx.x;
    uint64_t t0 = __builtin_readcyclecounter();
    uint64_t c0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N / 8; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (V == 0) {          // plain: sub, mul, add
                asm volatile("v_sub_f32 %0, %1, %0\n v_mul_f32 %0, %2, %0\n v_add_f32 %0, %3, %0" : "+v"(y) : "v"(x), "v"(g), "v"(y));
            }
        }
    }
    uint64_t c1 = clock64();
    uint64_t t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = y;
    if (threadIdx.x == 0) { cyc[0] = c1 - c0; cyc[1] = t1 - t0; }
}
// forms written out with explicit temporaries
__global__ void k_plain(float* out, uint64_t* cyc, float g, float x0) {
    float y = out[threadIdx.x], x = x0 + threadIdx.x, t;
    uint64_t c0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N / 8; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            asm volatile("v_sub_f32 %1, %2, %0\n v_mul_f32 %1, %3, %1\n v_add_f32 %0, %0, %1" : "+v"(y), "=&v"(t) : "v"(x), "v"(g));
    }
    uint64_t c1 = clock64();
    out[threadIdx.x] = y;
    if (threadIdx.x == 0) cyc[0] = c1 - c0;
}
__global__ void k_dpp(float* out, uint64_t* cyc, float g, float x0) {
    float y = out[threadIdx.x], x = x0 + threadIdx.x, t;
    uint64_t c0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N / 8; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            asm volatile("v_sub_f32_dpp %1, %2, %0 quad_perm:[0,1,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mul_f32 %1, %3, %1\n v_add_f32 %0, %0, %1" : "+v"(y), "=&v"(t) : "v"(x), "v"(g));
    }
    uint64_t c1 = clock64();
    out[threadIdx.x] = y;
    if (threadIdx.x == 0) cyc[0] = c1 - c0;
}
__global__ void k_dpp_nop(float* out, uint64_t* cyc, float g, float x0) {
    float y = out[threadIdx.x], x = x0 + threadIdx.x, t;
    uint64_t c0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N / 8; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            asm volatile("s_nop 1\n v_sub_f32_dpp %1, %2, %0 quad_perm:[0,1,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mul_f32 %1, %3, %1\n v_add_f32 %0, %0, %1" : "+v"(y), "=&v"(t) : "v"(x), "v"(g));
    }
    uint64_t c1 = clock64();
    out[threadIdx.x] = y;
    if (threadIdx.x == 0) cyc[0] = c1 - c0;
}
__global__ void k_mov_dpp(float* out, uint64_t* cyc, float g, float x0) {   // broadcast off the chain
    float y = out[threadIdx.x], x = x0 + threadIdx.x, t, e;
    uint64_t c0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N / 8; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            asm volatile("v_mov_b32_dpp %2, %3 quad_perm:[0,1,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n v_sub_f32 %1, %2, %0\n v_mul_f32 %1, %4, %1\n v_add_f32 %0, %0, %1" : "+v"(y), "=&v"(t), "=&v"(e) : "v"(x), "v"(g));
    }
    uint64_t c1 = clock64();
    out[threadIdx.x] = y;
    if (threadIdx.x == 0) cyc[0] = c1 - c0;
}
__global__ void k_indep(float* out, uint64_t* cyc, float g, float x0) {   // 3 independent ops per step (issue rate)
    float y = out[threadIdx.x], x = x0 + threadIdx.x, a = x, b = x, c = x;
    uint64_t c0 = clock64();
#pragma unroll 1
    for (int i = 0; i < N / 8; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            asm volatile("v_sub_f32 %0, %3, %0\n v_mul_f32 %1, %4, %1\n v_add_f32 %2, %2, %3" : "+v"(a), "+v"(b), "+v"(c) : "v"(x), "v"(g));
    }
    uint64_t c1 = clock64();
    out[threadIdx.x] = y + a + b + c;
    if (threadIdx.x == 0) cyc[0] = c1 - c0;
}
int main() {
    float* out; uint64_t* cyc;
    hipMalloc(&out, 64 * 4); hipMalloc(&cyc, 16); hipMemset(out, 0, 256);
    uint64_t h[2];
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    int wclk = 0; hipDeviceGetAttribute(&wclk, hipDeviceAttributeWallClockRate, 0);
    printf("clock rate %d kHz, wall clock rate %d kHz\n", clk, wclk);
#define RUN(K) for (int r = 0; r < 3; ++r) { hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0); \
        hipLaunchKernelGGL(K, 1, 64, 0, 0, out, cyc, 0.01f, 1.0f); hipEventRecord(e1); hipDeviceSynchronize(); float ms; hipEventElapsedTime(&ms, e0, e1); \
        hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost); if (r == 2) printf("%-12s clock64 ticks/step %.2f   (kernel %.1f us -> %.1f ns/step)\n", #K, (double)h[0] / N, ms * 1e3, ms * 1e6 / N); }
    RUN(k_plain) RUN(k_dpp) RUN(k_dpp_nop) RUN(k_mov_dpp) RUN(k_indep)
    return 0;
}
