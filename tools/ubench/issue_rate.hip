// VALU issue-rate probe (gfx950): eight independent streams of ONE instruction kind per lane, W = 1, 2, 4, 8 waves per
// SIMD on every CU.  Prints ns per wave-instruction per SIMD (at 2.4 GHz: 1.67 ns = 4 cycles, 0.83 ns = 2 cycles).
// Measured on MI355X (profiles/r02_issue_rate.txt): f32 fma / mul / add / sub, mov, and / xor / ashr / add_u32 issue at
// ~1.0-1.15 ns with >= 2 waves per SIMD; min / max / cmp / cvt / rndne / floor / lshl / bfe / perm / SDWA forms, all f64
// and all packed-f32 (v_pk_*) forms at ~1.75 ns -- so a packed op carries two values for the price of ~1.7 plain ones,
// not one; v_rcp_f32 at ~3.4 ns; ONE wave alone never issues faster than ~2.1 ns whatever the kind.
//   make -C tools/ubench issue_rate.bin && tools/ubench/issue_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define OP1(ins) "\n " ins " %0, %0\n " ins " %1, %1\n " ins " %2, %2\n " ins " %3, %3\n " ins " %4, %4\n " ins " %5, %5\n " ins " %6, %6\n " ins " %7, %7"
#define OP2(ins) "\n " ins " %0, %0, %8\n " ins " %1, %1, %8\n " ins " %2, %2, %8\n " ins " %3, %3, %8\n " ins " %4, %4, %8\n " ins " %5, %5, %8\n " ins " %6, %6, %8\n " ins " %7, %7, %8"
#define OP2R(ins) "\n " ins " %0, %8, %0\n " ins " %1, %8, %1\n " ins " %2, %8, %2\n " ins " %3, %8, %3\n " ins " %4, %8, %4\n " ins " %5, %8, %5\n " ins " %6, %8, %6\n " ins " %7, %8, %7"
#define OP3(ins) "\n " ins " %0, %0, %8, %0\n " ins " %1, %1, %8, %1\n " ins " %2, %2, %8, %2\n " ins " %3, %3, %8, %3\n " ins " %4, %4, %8, %4\n " ins " %5, %5, %8, %5\n " ins " %6, %6, %8, %6\n " ins " %7, %7, %8, %7"
#define CMP(ins) "\n " ins " vcc, %0, %8\n " ins " vcc, %1, %8\n " ins " vcc, %2, %8\n " ins " vcc, %3, %8\n " ins " vcc, %4, %8\n " ins " vcc, %5, %8\n " ins " vcc, %6, %8\n " ins " vcc, %7, %8"
#define CND "\n v_cndmask_b32_e64 %0, %0, %8, %9\n v_cndmask_b32_e64 %1, %1, %8, %9\n v_cndmask_b32_e64 %2, %2, %8, %9\n v_cndmask_b32_e64 %3, %3, %8, %9\n v_cndmask_b32_e64 %4, %4, %8, %9\n v_cndmask_b32_e64 %5, %5, %8, %9\n v_cndmask_b32_e64 %6, %6, %8, %9\n v_cndmask_b32_e64 %7, %7, %8, %9"
#define KERNEL32(name, body)                                                                                                      \
    __global__ __launch_bounds__(256) void name(float* out, float g, int n) {                                                     \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;        \
        const unsigned long long mask = 0x5555555555555555ull * (unsigned)n;                                                      \
        _Pragma("unroll 1") for (int i = 0; i < n; ++i) {                                                                         \
            REP8(asm volatile(body : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(g), "s"(mask) : "vcc");) \
        }                                                                                                                         \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                              \
    }
#define KERNEL64(name, body)                                                                                                      \
    __global__ __launch_bounds__(256) void name(float* out, float g, int n) {                                                     \
        double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, gg = g; \
        const unsigned long long mask = 0x5555555555555555ull * (unsigned)n;                                                      \
        _Pragma("unroll 1") for (int i = 0; i < n; ++i) {                                                                         \
            REP8(asm volatile(body : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(gg), "s"(mask) : "vcc");) \
        }                                                                                                                         \
        out[blockIdx.x * 256 + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);                                     \
    }
KERNEL32(k_fma, OP3("v_fma_f32"))
KERNEL32(k_mul, OP2("v_mul_f32"))
KERNEL32(k_add, OP2("v_add_f32"))
KERNEL32(k_sub, OP2("v_sub_f32"))
KERNEL32(k_min, OP2("v_min_f32"))
KERNEL32(k_max, OP2("v_max_f32"))
KERNEL32(k_mov, OP1("v_mov_b32"))
KERNEL32(k_cmp, CMP("v_cmp_lt_f32"))
KERNEL32(k_cnd, CND)
KERNEL32(k_rndne, OP1("v_rndne_f32"))
KERNEL32(k_floor, OP1("v_floor_f32"))
KERNEL32(k_cvt_i32_f32, OP1("v_cvt_i32_f32"))
KERNEL32(k_cvt_f32_i32, OP1("v_cvt_f32_i32"))
KERNEL32(k_rcp, OP1("v_rcp_f32"))
KERNEL32(k_xor, OP2("v_xor_b32"))
KERNEL32(k_and, OP2("v_and_b32"))
KERNEL32(k_lshl, OP2R("v_lshlrev_b32"))
KERNEL32(k_ashr, OP2R("v_ashrrev_i32"))
KERNEL32(k_bfe, OP3("v_bfe_i32"))
KERNEL32(k_addu, OP2("v_add_u32"))
KERNEL32(k_mullo, OP2("v_mul_lo_u32"))
KERNEL32(k_mulhi, OP2("v_mul_hi_u32"))
KERNEL32(k_perm, OP3("v_perm_b32"))
KERNEL32(k_cvt_f32_f16, OP1("v_cvt_f32_f16"))
KERNEL32(k_sdwa, "\n v_cvt_f32_i32_sdwa %0, sext(%0) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_i32_sdwa %1, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_i32_sdwa %2, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_i32_sdwa %3, sext(%3) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_i32_sdwa %4, sext(%4) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_i32_sdwa %5, sext(%5) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_i32_sdwa %6, sext(%6) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_i32_sdwa %7, sext(%7) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0")
KERNEL64(k_fma64, OP3("v_fma_f64"))
KERNEL64(k_mul64, OP2("v_mul_f64"))
KERNEL64(k_add64, OP2("v_add_f64"))
KERNEL64(k_rndne64, OP1("v_rndne_f64"))
KERNEL64(k_pkfma, OP3("v_pk_fma_f32"))
KERNEL64(k_pkmul, OP2("v_pk_mul_f32"))
KERNEL64(k_pkadd, OP2("v_pk_add_f32"))
KERNEL64(k_pkmov, "\n v_pk_mov_b32 %0, %0, %8\n v_pk_mov_b32 %1, %1, %8\n v_pk_mov_b32 %2, %2, %8\n v_pk_mov_b32 %3, %3, %8\n v_pk_mov_b32 %4, %4, %8\n v_pk_mov_b32 %5, %5, %8\n v_pk_mov_b32 %6, %6, %8\n v_pk_mov_b32 %7, %7, %8")

// Round 3: int16 -> f32 of a packed (l | r << 16) word, the two forms side by side (ns per SEQUENCE, not per instruction):
//  k_cvt16_sdwa   what sum_terms16w does: v_cvt_f32_i32_sdwa sext(WORD_0) + sext(WORD_1)                        (2 instructions)
//  k_cvt16_magic  the exact magic-number form: w ^ 0x80008000; (t & 0xffff) | 0x4B400000; 0x4B400000 | t.WORD_1 (an SDWA or);
//                 then - 12615680.0f on each half                                                              (5 instructions)
#define SEQ8(a, b) a("%0") b("%0") a("%1") b("%1") a("%2") b("%2") a("%3") b("%3") a("%4") b("%4") a("%5") b("%5") a("%6") b("%6") a("%7") b("%7")
#define CVT_LO(r) "\n v_cvt_f32_i32_sdwa %10, sext(" r ") dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0"
#define CVT_HI(r) "\n v_cvt_f32_i32_sdwa " r ", sext(" r ") dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1"
#define MAG_A(r) "\n v_xor_b32 " r ", 0x80008000, " r "\n v_and_or_b32 %10, " r ", %11, %12\n v_or_b32_sdwa " r ", %12, " r " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
#define MAG_B(r) "\n v_sub_f32 %10, %10, %13\n v_sub_f32 " r ", " r ", %13"
#define KERNEL_CVT(name, A, B)                                                                                                    \
    __global__ __launch_bounds__(256) void name(float* out, float g, int n) {                                                     \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, t = 0;  \
        const unsigned long long mask = 0x5555555555555555ull * (unsigned)n;                                                      \
        const unsigned lo16 = 0xffffu, magic = 0x4B400000u;                                                                       \
        const float c = 12615680.0f;                                                                                              \
        _Pragma("unroll 1") for (int i = 0; i < n; ++i) {                                                                         \
            REP8(asm volatile(SEQ8(A, B) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)        \
                              : "v"(g), "s"(mask), "v"(t), "v"(lo16), "v"(magic), "v"(c) : "vcc");)                               \
        }                                                                                                                         \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + t;                                          \
    }
KERNEL_CVT(k_cvt16_sdwa, CVT_LO, CVT_HI)
KERNEL_CVT(k_cvt16_magic, MAG_A, MAG_B)

typedef void (*kern_t)(float*, float, int);
static void run(const char* name, kern_t k, float* d) {
    printf("%-16s", name);
    for (int w : {1, 2, 4, 8}) {
        const int n = 1000, blocks = 256 * w;   // 256 CUs x (one 4-wave block = 1 wave per SIMD) x w
        hipEvent_t a, b;
        (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, n);
        (void)hipEventRecord(a, 0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, n);
        (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
        float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
        printf("  %dw %.3f ns", w, ms * 1e6 / (64.0 * n * w));
    }
    printf("\n");
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
#define R(k) run(#k, k, d)
    R(k_fma); R(k_mul); R(k_add); R(k_sub); R(k_min); R(k_max); R(k_mov); R(k_cmp); R(k_cnd); R(k_rndne); R(k_floor);
    R(k_cvt_i32_f32); R(k_cvt_f32_i32); R(k_rcp); R(k_xor); R(k_and); R(k_lshl); R(k_ashr); R(k_bfe); R(k_addu); R(k_mullo); R(k_mulhi);
    R(k_perm); R(k_cvt_f32_f16); R(k_sdwa); R(k_fma64); R(k_mul64); R(k_add64); R(k_rndne64); R(k_pkfma); R(k_pkmul); R(k_pkadd); R(k_pkmov);
    printf("(the next two: ns per int16-pair conversion SEQUENCE -- 2 and 5 instructions)\n");
    R(k_cvt16_sdwa); R(k_cvt16_magic);
    return 0;
}
