/* The double-precision sequence of kernels.hip sin_glibc, on the host, against the host's own sinf over EVERY finite float
 * (test infrastructure: tests/test_sinf_restate.py builds and runs it; nothing of the product links it).
 *
 * glibc's sinf (2.28 and later; sysdeps/ieee754/flt-32/s_sinf.c + sincosf.h, after ARM's optimized-routines) is restated from its
 * published algorithm; the constants are the ones its table holds (libm.so.6 __sincosf_table / __inv_pio4).  The x86-64 build of
 * glibc picks its FMA variant on any CPU of the last decade: every `a + b * c` of the source is ONE fused operation there, and
 * with fma() in exactly those places this file agrees with sinf on all 4 278 190 080 finite inputs (without them: 12 differ).
 *   gcc -O2 -mfma -ffp-contract=off tools/sinf_restate.c -o sinf_restate -lm -lpthread && ./sinf_restate      (~8 s on 8 cores) */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <pthread.h>
typedef struct { double sign[4]; double hpi_inv, hpi, c0, c1, c2, c3, c4, s1, s2, s3; } sc_t;
static const sc_t T[2] = {
 {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, 0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5, -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13},
 {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.55553e1068f19p-5, 0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13}};
static const uint32_t inv_pio4[24] = {0xa2, 0xa2f9, 0xa2f983, 0xa2f9836e, 0xf9836e4e, 0x836e4e44, 0x6e4e4415, 0x4e441529, 0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1, 0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0, 0x34ddc0db, 0xddc0db62, 0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041};
static inline uint32_t asuint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline uint32_t abstop12(float x) { return (asuint(x) >> 20) & 0x7ff; }
static inline float poly(double x, double x2, const sc_t* p, int n) {
    if ((n & 1) == 0) { double x3 = x * x2; double s1 = fma(x2, p->s3, p->s2); double x7 = x3 * x2; double s = fma(x3, p->s1, x); return (float)fma(x7, s1, s); }
    double x4 = x2 * x2; double c2 = fma(x2, p->c4, p->c3); double c1 = fma(x2, p->c1, p->c0); double x6 = x4 * x2; double c = fma(x4, p->c2, c1); return (float)fma(x6, c2, c);
}
float my_sinf(float y) {
    double x = y; double s; int n; const sc_t* p = &T[0];
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        s = x * x;
        if (abstop12(y) < abstop12(0x1p-12f)) return y;
        return poly(x, s, p, 0);
    } else if (abstop12(y) < abstop12(120.0f)) {
        double r = x * p->hpi_inv;
        n = ((int32_t)r + 0x800000) >> 24;
        x = fma(-(double)n, p->hpi, x);
        s = p->sign[n & 3];
        if (n & 2) p = &T[1];
        return poly(x * s, x * x, p, n);
    } else if (abstop12(y) < abstop12(INFINITY)) {
        uint32_t xi = asuint(y); int sign = xi >> 31;
        const uint32_t* arr = &inv_pio4[(xi >> 26) & 15];
        int shift = (xi >> 23) & 7;
        uint64_t nn, res0, res1, res2;
        xi = (xi & 0xffffff) | 0x800000;
        xi <<= shift;
        res0 = xi * arr[0];
        res1 = (uint64_t)xi * arr[4];
        res2 = (uint64_t)xi * arr[8];
        res0 = (res2 >> 32) | (res0 << 32);
        res0 += res1;
        nn = (res0 + (1ULL << 61)) >> 62;
        res0 -= nn << 62;
        x = (double)(int64_t)res0 * 0x1.921FB54442D18p-62;
        n = (int)nn;
        s = p->sign[(n + sign) & 3];
        if ((n + sign) & 2) p = &T[1];
        return poly(x * s, x * x, p, n);
    }
    return y - y;
}
static unsigned long long bad[8]; static uint32_t firstbad[8];
static void* work(void* a) {
    int t = (int)(intptr_t)a; unsigned long long b = 0;
    for (uint64_t u = (uint64_t)t << 29; u < ((uint64_t)(t + 1) << 29); ++u) {
        uint32_t ui = (uint32_t)u; float f; memcpy(&f, &ui, 4);
        if (!isfinite(f)) continue;
        float r = sinf(f), m = my_sinf(f);
        if (asuint(r) != asuint(m)) { if (!b) firstbad[t] = ui; ++b; }
    }
    bad[t] = b; return 0;
}
int main() {
    pthread_t th[8];
    for (int t = 0; t < 8; ++t) pthread_create(&th[t], 0, work, (void*)(intptr_t)t);
    unsigned long long tot = 0;
    for (int t = 0; t < 8; ++t) { pthread_join(th[t], 0); tot += bad[t]; if (bad[t]) printf("thread %d: %llu differ, first %08x\n", t, bad[t], firstbad[t]); }
    printf("all finite floats: %llu differ from glibc sinf\n", tot);
    return 0;
}
