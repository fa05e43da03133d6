// Hand-off latency between two workgroups (gfx950): a ping-pong over two 8-byte words, the participants chosen by
// workgroup index (a grid of 64 one-wave workgroups: index i runs on XCD i % 8 -- the kernel reports XCC_ID), with the
// cache-control bits of the store and of the polling load spelled out:
//   sc1        agent scope (what the engine's granules use)
//   sc0        workgroup scope: the load misses the CU's L1 and is served by the XCD's L2
//   sc0 sc1    system scope
// Prints ns per ONE-WAY hop (half a round trip).  A wait is bounded: a combination that never sees the other side's word
// (a stale line in the XCD's L2) is reported as "no hand-off", not hung.
//   make -C tools/ubench handoff.bin && tools/ubench/handoff.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define LOADER(name, bits)                                                                                       \
    __device__ inline unsigned long long name(const unsigned long long* p) {                                    \
        unsigned long long v;                                                                                    \
        asm volatile("global_load_dwordx2 %0, %1, off " bits "\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); \
        return v;                                                                                                \
    }
#define STORER(name, bits)                                                                                       \
    __device__ inline void name(unsigned long long* p, unsigned long long v) {                                  \
        asm volatile("global_store_dwordx2 %0, %1, off " bits : : "v"(p), "v"(v) : "memory");                    \
    }
LOADER(ld_sc1, "sc1")
LOADER(ld_sc0, "sc0")
LOADER(ld_sys, "sc0 sc1")
STORER(st_sc1, "sc1")
STORER(st_sc0, "sc0")
STORER(st_sys, "sc0 sc1")
STORER(st_plain, "")

template <int LD, int ST>
__global__ __launch_bounds__(64) void k_pingpong(unsigned long long* words, uint32_t a, uint32_t b, uint32_t rounds, unsigned long long* out) {
    const uint32_t me = blockIdx.x;
    if (threadIdx.x != 0) return;
    if (me != a && me != b) return;
    auto ld = [](const unsigned long long* p) { return LD == 0 ? ld_sc1(p) : LD == 1 ? ld_sc0(p) : ld_sys(p); };
    auto st = [](unsigned long long* p, unsigned long long v) { if (ST == 0) st_sc1(p, v); else if (ST == 1) st_sc0(p, v); else if (ST == 2) st_sys(p, v); else st_plain(p, v); };
    unsigned long long* mine = words + (me == a ? 0 : 16), *theirs = words + (me == a ? 16 : 0);   // (different 128-byte lines)
    const unsigned long long t0 = wall_clock64();
    uint32_t done = 0;
    for (uint32_t i = 1; i <= rounds; ++i) {
        if (me == a) st(mine, i);
        uint32_t polls = 0;
        while (ld(theirs) < i) if (++polls > 2000000u) goto out;
        if (me == b) st(mine, i);
        done = i;
    }
out:
    const unsigned long long t1 = wall_clock64();
    const uint32_t xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) & 0xF;
    unsigned long long* o = out + (me == a ? 0 : 4);
    o[0] = t1 - t0;
    o[1] = done;
    o[2] = xcc;
    if (done != rounds) st_sys(mine, 0xFFFFFFFFull);   // (let the other side out)
}

// The same ping-pong (agent-scope store and loads) with the poll loop PIPELINED: DEPTH loads of the word in flight, the oldest
// looked at while the next is issued -- the wait for a hand-off then ends within a DEPTH-th of a load round trip of the
// word's arrival instead of, on average, half a round trip after it.
template <int DEPTH>
__device__ inline bool poll_pipelined(const uint32_t* p, uint32_t want) {
    uint32_t r0, r1, r2, r3, budget = 4000000u;
    if (DEPTH == 2) {
        asm volatile(
            "global_load_dword %0, %4, off sc1\n"
            "L1_%=:\n"
            "global_load_dword %1, %4, off sc1\n"
            "s_waitcnt vmcnt(1)\n"
            "v_cmp_ge_u32_e32 vcc, %0, %5\n"
            "s_cbranch_vccnz L2_%=\n"
            "global_load_dword %0, %4, off sc1\n"
            "s_waitcnt vmcnt(1)\n"
            "v_cmp_ge_u32_e32 vcc, %1, %5\n"
            "s_cbranch_vccnz L2_%=\n"
            "s_sub_u32 %6, %6, 1\n"
            "s_cmp_eq_u32 %6, 0\n"
            "s_cbranch_scc0 L1_%=\n"
            "L2_%=:\n"
            "s_waitcnt vmcnt(0)\n"
            : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "+v"(p), "+v"(want), "+s"(budget) : : "vcc", "scc", "memory");
    } else {
        asm volatile(
            "global_load_dword %0, %4, off sc1\n"
            "global_load_dword %1, %4, off sc1\n"
            "global_load_dword %2, %4, off sc1\n"
            "L1_%=:\n"
            "global_load_dword %3, %4, off sc1\n"
            "s_waitcnt vmcnt(3)\n"
            "v_cmp_ge_u32_e32 vcc, %0, %5\n"
            "s_cbranch_vccnz L2_%=\n"
            "global_load_dword %0, %4, off sc1\n"
            "s_waitcnt vmcnt(3)\n"
            "v_cmp_ge_u32_e32 vcc, %1, %5\n"
            "s_cbranch_vccnz L2_%=\n"
            "global_load_dword %1, %4, off sc1\n"
            "s_waitcnt vmcnt(3)\n"
            "v_cmp_ge_u32_e32 vcc, %2, %5\n"
            "s_cbranch_vccnz L2_%=\n"
            "global_load_dword %2, %4, off sc1\n"
            "s_waitcnt vmcnt(3)\n"
            "v_cmp_ge_u32_e32 vcc, %3, %5\n"
            "s_cbranch_vccnz L2_%=\n"
            "s_sub_u32 %6, %6, 1\n"
            "s_cmp_eq_u32 %6, 0\n"
            "s_cbranch_scc0 L1_%=\n"
            "L2_%=:\n"
            "s_waitcnt vmcnt(0)\n"
            : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "+v"(p), "+v"(want), "+s"(budget) : : "vcc", "scc", "memory");
    }
    return budget != 0u;
}
__device__ inline void st32_sc1(uint32_t* p, uint32_t v) { asm volatile("global_store_dword %0, %1, off sc1" : : "v"(p), "v"(v) : "memory"); }
template <int DEPTH>
__global__ __launch_bounds__(64) void k_pingpong_pipe(uint32_t* words, uint32_t a, uint32_t b, uint32_t rounds, unsigned long long* out) {
    const uint32_t me = blockIdx.x;
    if (threadIdx.x != 0) return;
    if (me != a && me != b) return;
    uint32_t* mine = words + (me == a ? 0 : 32), *theirs = words + (me == a ? 32 : 0);
    const unsigned long long t0 = wall_clock64();
    uint32_t done = 0;
    for (uint32_t i = 1; i <= rounds; ++i) {
        if (me == a) st32_sc1(mine, i);
        if (!poll_pipelined<DEPTH>(theirs, i)) break;
        if (me == b) st32_sc1(mine, i);
        done = i;
    }
    const unsigned long long t1 = wall_clock64();
    const uint32_t xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) & 0xF;
    unsigned long long* o = out + (me == a ? 0 : 4);
    o[0] = t1 - t0; o[1] = done; o[2] = xcc;
    if (done != rounds) st32_sc1(mine, 0xFFFFFFFFu);
}

int main() {
    unsigned long long *words, *out;
    hipMalloc(&words, 4096);
    hipMalloc(&out, 64);
    const uint32_t rounds = 20000;
    const char* ldn[3] = {"sc1", "sc0", "sc0 sc1"};
    const char* stn[4] = {"sc1", "sc0", "sc0 sc1", "(none)"};
    const uint32_t pairs[4][2] = {{0, 8}, {0, 16}, {0, 1}, {0, 4}};
    typedef void (*K)(unsigned long long*, uint32_t, uint32_t, uint32_t, unsigned long long*);
    K ks[3][4] = {{k_pingpong<0, 0>, k_pingpong<0, 1>, k_pingpong<0, 2>, k_pingpong<0, 3>},
                  {k_pingpong<1, 0>, k_pingpong<1, 1>, k_pingpong<1, 2>, k_pingpong<1, 3>},
                  {k_pingpong<2, 0>, k_pingpong<2, 1>, k_pingpong<2, 2>, k_pingpong<2, 3>}};
    for (int p = 0; p < 4; ++p)
        for (int l = 0; l < 3; ++l)
            for (int s = 0; s < 4; ++s) {
                hipMemset(words, 0, 4096);
                hipMemset(out, 0, 64);
                hipLaunchKernelGGL(ks[l][s], dim3(64), dim3(64), 0, 0, words, pairs[p][0], pairs[p][1], rounds, out);
                if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
                unsigned long long h[8];
                hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
                // wall_clock64: 100 MHz
                if (h[1] == rounds) printf("wg %2u (xcc %llu) <-> wg %2u (xcc %llu)  load %-8s store %-8s  %7.1f ns per hop\n", pairs[p][0], h[2], pairs[p][1], h[6], ldn[l], stn[s], (double)h[0] * 10.0 / (2.0 * rounds));
                else printf("wg %2u (xcc %llu) <-> wg %2u (xcc %llu)  load %-8s store %-8s  no hand-off (stopped at round %llu)\n", pairs[p][0], h[2], pairs[p][1], h[6], ldn[l], stn[s], h[1]);
            }
    for (int p = 0; p < 4; ++p)
        for (int depth = 2; depth <= 4; depth += 2) {
            hipMemset(words, 0, 4096);
            hipMemset(out, 0, 64);
            if (depth == 2) hipLaunchKernelGGL(k_pingpong_pipe<2>, dim3(64), dim3(64), 0, 0, (uint32_t*)words, pairs[p][0], pairs[p][1], rounds, out);
            else hipLaunchKernelGGL(k_pingpong_pipe<4>, dim3(64), dim3(64), 0, 0, (uint32_t*)words, pairs[p][0], pairs[p][1], rounds, out);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
            unsigned long long h[8];
            hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
            if (h[1] == rounds) printf("wg %2u (xcc %llu) <-> wg %2u (xcc %llu)  sc1 / sc1, %d polls in flight  %7.1f ns per hop\n", pairs[p][0], h[2], pairs[p][1], h[6], depth, (double)h[0] * 10.0 / (2.0 * rounds));
            else printf("wg %2u (xcc %llu) <-> wg %2u (xcc %llu)  sc1 / sc1, %d polls in flight  no hand-off (round %llu)\n", pairs[p][0], h[2], pairs[p][1], h[6], depth, h[1]);
        }
    return 0;
}
