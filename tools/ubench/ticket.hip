// What drawing tickets costs (gfx950): G workgroups of 256 threads, thread 0 of each draws a ticket from ONE counter
// (atomicAdd, agent scope, the value needed -- k_band_scan / k_band_chain number their tiles this way), hands it to the
// workgroup through LDS, every thread writes one word; against the same kernel numbering its tiles by blockIdx, and
// against drawing from one of 8 counters picked by XCC_ID.
//   make -C tools/ubench ticket.bin && tools/ubench/ticket.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ __launch_bounds__(256) void k_ticket(uint32_t* ctr, uint32_t* out) {
    __shared__ uint32_t t;
    if (threadIdx.x == 0) {
        if (MODE == 0) t = blockIdx.x;
        else if (MODE == 1) t = atomicAdd(ctr, 1u);
        else {
            const uint32_t xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) & 7u;
            t = atomicAdd(ctr + 32u * xcc, 1u) * 8u + xcc;
        }
    }
    __syncthreads();
    out[(t % gridDim.x) * 256u + threadIdx.x] = t;
}
int main() {
    uint32_t *ctr, *out;
    hipMalloc(&ctr, 4096);
    hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int grids[3] = {256, 704, 2816};
    for (int g = 0; g < 3; ++g)
        for (int mode = 0; mode < 3; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 20; ++rep) {
                hipMemsetAsync(ctr, 0, 4096, 0);
                hipEventRecord(a, 0);
                if (mode == 0) hipLaunchKernelGGL(k_ticket<0>, dim3(grids[g]), dim3(256), 0, 0, ctr, out);
                else if (mode == 1) hipLaunchKernelGGL(k_ticket<1>, dim3(grids[g]), dim3(256), 0, 0, ctr, out);
                else hipLaunchKernelGGL(k_ticket<2>, dim3(grids[g]), dim3(256), 0, 0, ctr, out);
                hipEventRecord(b, 0);
                hipEventSynchronize(b);
                float ms;
                hipEventElapsedTime(&ms, a, b);
                if (rep > 2 && ms < best) best = ms;
            }
            printf("%5d workgroups  %-28s %7.2f us\n", grids[g], mode == 0 ? "tiles by blockIdx" : mode == 1 ? "tickets from ONE counter" : "tickets from 8 (per XCD)", best * 1e3f);
        }
    return 0;
}
