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
    return 0;
}
