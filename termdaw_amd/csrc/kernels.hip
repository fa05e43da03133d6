// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for termdaw's vertex render path.
//
// All kernels are whole-timeline ("chunk") kernels: instead of the reference's per-block recursion
// (graph.rs:98-121, one 8 KB buffer per vertex per block) a launch produces a vertex' edge buffer for
// every frame of the chunk, batched over same-kind vertices through blockIdx.y.  Arithmetic is f32,
// IEEE, in the reference's operation order; the build passes -ffp-contract=off so no FMA contraction
// changes a rounding.  No MFMA anywhere: these are streaming / gather / short-recurrence loops whose
// roofline is HBM bandwidth (DESIGN.md "Kernels").
//
// Access pattern: a 256-thread workgroup owns a 1024-frame tile; thread t touches frame pairs
// tile+2t and tile+512+2t, i.e. one 16-byte float4 per lane per access and 1 KiB contiguous per
// wave-instruction.
#include "kernels.h"

#include <algorithm>

namespace tdk {

#define TD_DEV __device__ __forceinline__

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
TD_DEV float2 epilogue(float2 v, const PanGain& pg) {
    if (pg.flags & 1u) { v.x *= pg.l_amp; v.y *= pg.r_amp; }   // Sample::apply_angle sample.rs:102-105
    if (pg.flags & 2u) { v.x *= pg.gain;  v.y *= pg.gain;  }   // Sample::apply_gain  sample.rs:110-113
    return v;
}
TD_DEV float4 epilogue4(float4 v, const PanGain& pg) {
    float2 a = epilogue(make_float2(v.x, v.y), pg), b = epilogue(make_float2(v.z, v.w), pg);
    return make_float4(a.x, a.y, b.x, b.y);
}

// Every buffer pointer reaches a kernel through a descriptor in memory, so the compiler only knows it as a
// generic ("flat") pointer and would emit flat_load / flat_store -- which count on both vmcnt and lgkmcnt
// and force `s_waitcnt vmcnt(0) lgkmcnt(0)` before ANY use, i.e. no load can stay in flight across a use.
// All of them are device-global: these helpers say so (address space 1 -> global_load / global_store with
// counted vmcnt waits, so prefetched batches really overlap the arithmetic).
#define TD_GLOBAL __attribute__((address_space(1)))
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef unsigned int u2v __attribute__((ext_vector_type(2)));
typedef int i4v __attribute__((ext_vector_type(4)));
typedef int i2v __attribute__((ext_vector_type(2)));
TD_DEV float4 gload4(const void* p) { const f4v v = *reinterpret_cast<const f4v TD_GLOBAL*>((const TD_GLOBAL char*)p); return make_float4(v.x, v.y, v.z, v.w); }
typedef float f4v_u __attribute__((ext_vector_type(4), aligned(8)));
TD_DEV float4 gload4u(const void* p) { const f4v_u v = *reinterpret_cast<const f4v_u TD_GLOBAL*>((const TD_GLOBAL char*)p); return make_float4(v.x, v.y, v.z, v.w); }
TD_DEV float2 gload2(const void* p) { const f2v v = *reinterpret_cast<const f2v TD_GLOBAL*>((const TD_GLOBAL char*)p); return make_float2(v.x, v.y); }
TD_DEV float gload1(const float* p) { return *reinterpret_cast<const float TD_GLOBAL*>((const TD_GLOBAL char*)p); }
TD_DEV void gstore4(void* p, float4 v) { f4v w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w; *reinterpret_cast<f4v TD_GLOBAL*>((TD_GLOBAL char*)p) = w; }
TD_DEV void gstore2(void* p, float2 v) { f2v w; w.x = v.x; w.y = v.y; *reinterpret_cast<f2v TD_GLOBAL*>((TD_GLOBAL char*)p) = w; }

// Two consecutive frames starting at frame m (m even).  Buffers are padded to an even frame count, so a
// pair whose first frame is valid may always be accessed as one 16-byte word.
TD_DEV float4 load_pair(const float2* p, uint32_t m, uint32_t M) {
    if (m + 1 < M) return gload4(p + m);
    if (m < M) { const float2 a = gload2(p + m); return make_float4(a.x, a.y, 0.f, 0.f); }   // odd tail: pad reads as 0
    return make_float4(0.f, 0.f, 0.f, 0.f);
}
TD_DEV void store_pair(float2* p, uint32_t m, uint32_t M, float4 v) {
    if (m < M) gstore4(p + m, v);
}

// The planar-in-4 copy of a stereo stream (a band-pass vertex' materialised input sum): every aligned four frames as
// {l0 l1 l2 l3}{r0 r1 r2 r3}, same word addresses as the interleaved four.  Frame m: two dwords, 16 bytes apart.
TD_DEV float2 q4_frame(const float* q, uint32_t m) {
    const float* p = q + (size_t)(m >> 2) * 8u + (m & 3u);
    return make_float2(gload1(p), gload1(p + 4));
}
TD_DEV float q4_chan(const float* q, uint32_t m, uint32_t ch) { return gload1(q + (size_t)(m >> 2) * 8u + (m & 3u) + 4u * ch); }

typedef float f32x2 __attribute__((ext_vector_type(2)));
// A double moved across lanes by DPP (two 32-bit moves on the VALU; no trip through the LDS crossbar as __shfl takes).
// Lanes whose source lies outside the row / wave, or that the row mask leaves out, receive 0.
template <int CTRL, int ROW_MASK>
TD_DEV double dpp_f64(double v) {
    const long long u = __double_as_longlong(v);
    constexpr bool kAll = ROW_MASK == 0xF;   // (every row written: "no source -> 0" is the instruction's own bound_ctrl, no 0 to pre-load)
    const int lo = __builtin_amdgcn_update_dpp(0, (int)u, CTRL, ROW_MASK, 0xF, kAll);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(u >> 32), CTRL, ROW_MASK, 0xF, kAll);
    return __longlong_as_double((long long)(((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo));
}
constexpr int kDppRowShr = 0x110, kDppWaveShr1 = 0x138, kDppRowBcast15 = 0x142, kDppRowBcast31 = 0x143;
// the wave's sum of v, left in lane 63 (running sums inside the rows of 16, then row and half-wave totals passed on: a fixed order)
TD_DEV double wave_sum_to_lane63(double v) {
    v += dpp_f64<kDppRowShr + 1, 0xF>(v);
    v += dpp_f64<kDppRowShr + 2, 0xF>(v);
    v += dpp_f64<kDppRowShr + 4, 0xF>(v);
    v += dpp_f64<kDppRowShr + 8, 0xF>(v);
    v += dpp_f64<kDppRowBcast15, 0xA>(v);
    v += dpp_f64<kDppRowBcast31, 0xC>(v);
    return v;
}
TD_DEV float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// Rust `(x * amplitude) as i16` / `as i32` (state.rs:521-522, 529-530): truncate toward zero, saturate,
// NaN -> 0.
TD_DEV int32_t quant16(float x, float amp) {
    x = x * amp;
    if (!(x == x)) return 0;
    x = fminf(fmaxf(x, -32768.0f), 32767.0f);
    return (int32_t)x;
}
TD_DEV int32_t quant32(float x, float amp) {
    x = x * amp;
    if (!(x == x)) return 0;
    if (x <= -2147483648.0f) return INT32_MIN;
    if (x >= 2147483648.0f) return INT32_MAX;
    return (int32_t)x;
}
TD_DEV void store_quant_pair(void* pcm, uint32_t qmode, uint32_t m, uint32_t M, float4 v, float amp) {
    if (m >= M) return;
    if (qmode == 1) {
        uint32_t w0 = ((uint32_t)quant16(v.x, amp) & 0xFFFFu) | ((uint32_t)quant16(v.y, amp) << 16);
        uint32_t w1 = ((uint32_t)quant16(v.z, amp) & 0xFFFFu) | ((uint32_t)quant16(v.w, amp) << 16);
        uint32_t* o = reinterpret_cast<uint32_t*>(pcm) + m;  // one 32-bit word per frame
        if (m + 1 < M) { u2v w; w.x = w0; w.y = w1; *reinterpret_cast<u2v TD_GLOBAL*>((TD_GLOBAL char*)o) = w; }
        else *reinterpret_cast<uint32_t TD_GLOBAL*>((TD_GLOBAL char*)o) = w0;
    } else {
        int32_t* o = reinterpret_cast<int32_t*>(pcm) + 2 * (size_t)m;
        if (m + 1 < M) {
            i4v w; w.x = quant32(v.x, amp); w.y = quant32(v.y, amp); w.z = quant32(v.z, amp); w.w = quant32(v.w, amp);
            *reinterpret_cast<i4v TD_GLOBAL*>((TD_GLOBAL char*)o) = w;
        } else {
            i2v w; w.x = quant32(v.x, amp); w.y = quant32(v.y, amp);
            *reinterpret_cast<i2v TD_GLOBAL*>((TD_GLOBAL char*)o) = w;
        }
    }
}

TD_DEV float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// Inter-workgroup hand-off inside a launch: 8-byte {tag = 1, value} granules, one agent-scope atomic store each, read back
// with agent-scope atomic loads (global_store / global_load ... sc1: L1 bypassed, the data word carries its own validity,
// so no fence on either side; cdna_hip_programming.md Guideline 16, form R2); the granule words are zeroed by the engine
// before every launch (ONE memset per submission) -- except the single-pass Normalize's tile words, whose tag is the
// submission's epoch (granule_store's `tag`).
typedef unsigned long long TD_GLOBAL* gu64;
typedef uint32_t TD_GLOBAL* gu32;
// (tag: 1 for words the engine zeroes before the launch; the single-pass Normalize's tile words carry the submission's EPOCH
// instead -- a kernel argument, larger than every earlier one -- and are not zeroed between launches: engine.cpp, submit_chunk)
TD_DEV void granule_store(unsigned long long* p, uint32_t value, uint32_t tag = 1u) {
    __hip_atomic_store((gu64)(TD_GLOBAL char*)p, ((unsigned long long)tag << 32) | (unsigned long long)value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
TD_DEV unsigned long long granule_load(const unsigned long long* p) {
    return __hip_atomic_load((gu64)(TD_GLOBAL char*)const_cast<unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
constexpr uint32_t kScanSpinLimitSum = 20000;   // polls (~1 us each) of one earlier tile's granule before a single-pass Normalize tile gives up
// A single-pass Normalize tile (SumDesc modes 4 / 5) whose bounded wait gave up: the device-side flag k_norm_fix looks at,
// and the word in page-locked host memory the engine looks at once the stream has drained (system scope: it must be
// visible to the host when the launch has completed).
TD_DEV void raise_violated(NormState* st, uint32_t* host_flag) {
    st->violated = 1u;
    if (host_flag) __hip_atomic_store((gu32)(TD_GLOBAL char*)host_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// The granules of ALL tiles below `n`, a workgroup's threads striding over them: f(index, value) for each; false when one
// did not show within `limit` polls.  Four loads per thread are in flight before the first is looked at -- a thread's
// granules, one dependent round trip after the other, were 2 - 3 us of every launch that ends in such a gather.
template <typename F>
TD_DEV bool for_lower_granules(const unsigned long long* sync, uint32_t n, uint32_t limit, F f, uint32_t tag = 1u) {
    bool ok = true;
    for (uint32_t base = threadIdx.x; base < n; base += 4u * (uint32_t)kThreads) {
        unsigned long long g[4];
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
            const uint32_t idx = base + u * (uint32_t)kThreads;
            g[u] = idx < n ? granule_load(sync + idx) : 0ull;
        }
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
            const uint32_t idx = base + u * (uint32_t)kThreads;
            if (idx >= n) continue;
            for (uint32_t spin = 0; (uint32_t)(g[u] >> 32) != tag && spin < limit; ++spin) {
                __builtin_amdgcn_s_sleep(2);
                g[u] = granule_load(sync + idx);
            }
            if ((uint32_t)(g[u] >> 32) == tag) f(idx, (uint32_t)g[u]);
            else ok = false;
        }
    }
    return ok;
}

// ... the same over two granule arrays at once: f(index, value_a, value_b)
template <typename F>
TD_DEV bool for_lower_granules2(const unsigned long long* sa, const unsigned long long* sb, uint32_t n, uint32_t limit, F f) {
    bool ok = true;
    for (uint32_t base = threadIdx.x; base < n; base += 2u * (uint32_t)kThreads) {
        unsigned long long ga[2], gb[2];
#pragma unroll
        for (uint32_t u = 0; u < 2u; ++u) {
            const uint32_t idx = base + u * (uint32_t)kThreads;
            ga[u] = idx < n ? granule_load(sa + idx) : 0ull;
            gb[u] = idx < n ? granule_load(sb + idx) : 0ull;
        }
#pragma unroll
        for (uint32_t u = 0; u < 2u; ++u) {
            const uint32_t idx = base + u * (uint32_t)kThreads;
            if (idx >= n) continue;
            for (uint32_t spin = 0; ((uint32_t)(ga[u] >> 32) != 1u || (uint32_t)(gb[u] >> 32) != 1u) && spin < limit; ++spin) {
                __builtin_amdgcn_s_sleep(2);
                ga[u] = granule_load(sa + idx);
                gb[u] = granule_load(sb + idx);
            }
            if ((uint32_t)(ga[u] >> 32) == 1u && (uint32_t)(gb[u] >> 32) == 1u) f(idx, (uint32_t)ga[u], (uint32_t)gb[u]);
            else ok = false;
        }
    }
    return ok;
}

template <typename IDX>
TD_DEV float4 loop_pair(const float2* s, IDX len, IDX idx) {
    float2 a = gload2(s + idx);
    IDX i1 = idx + 1;
    if (i1 == len) i1 = 0;
    float2 b = gload2(s + i1);
    return make_float4(a.x, a.y, b.x, b.y);
}
// x mod len for x < 2^32 with magic = floor(2^32 / len): q underestimates x / len by at most 1
TD_DEV uint32_t barrett_mod(uint32_t x, uint32_t len, uint32_t magic) {
    const uint32_t r = x - __umulhi(x, magic) * len;
    return r >= len ? r - len : r;
}
// sample_loop_gen for one frame pair (extensions.rs:337-338), 32-bit cursor: one 16-byte load at any frame
// (global_load_dwordx4 needs dword alignment only).
TD_DEV float4 loop_pair32(const float2* s, uint32_t len, uint32_t magic, uint32_t x) {
    const uint32_t idx = barrett_mod(x, len, magic);
    return gload4u(s + idx);   // bank entries end with a copy of their first frame: the pair never needs a wrap
}
// stand-alone sample_loop vertex (k_sample_loop): wave-uniform choice of the 32-bit form
TD_DEV float4 gather_loop_pair(const float2* s, uint64_t len64, uint64_t t0, uint32_t magic, uint32_t m) {
    if (magic) return loop_pair32(s, (uint32_t)len64, magic, (uint32_t)t0 + m);
    return loop_pair<uint64_t>(s, len64, (t0 + m) % len64);
}

TD_DEV float4 zero_tail(float4 v, uint32_t m, uint32_t M) {   // frames at or beyond M contribute nothing
    if (m + 1 >= M) { v.z = 0.f; v.w = 0.f; }
    if (m >= M) { v.x = 0.f; v.y = 0.f; }
    return v;
}

// The term table of a vertex is read-only for the whole launch and indexed uniformly: reading it through
// the constant address space turns every field access into a scalar load (s_load) instead of a per-lane
// flat load.
#define TD_CONST __attribute__((address_space(4)))
typedef const InTerm TD_CONST* TermTab;
TD_DEV TermTab term_tab(const InTerm* p) { return (TermTab)(const TD_CONST char*)p; }
TD_DEV PanGain term_pg(TermTab t, uint32_t j) {
    PanGain pg;
    pg.l_amp = t[j].pg.l_amp; pg.r_amp = t[j].pg.r_amp; pg.gain = t[j].pg.gain; pg.flags = t[j].pg.flags;
    return pg;
}
TD_DEV float2 unpack16(uint32_t w, float sl, float sr);
// value of one input term for the frame pair starting at m (generic form)
TD_DEV float4 term_pair(TermTab t, uint32_t j, uint32_t m, uint32_t M) {
    const float2* p = t[j].p;
    const uint32_t kind = t[j].kind;
    if (kind == 0) return load_pair(p, m, M);
    if (kind == 4) {   // a single-input Sum vertex read through: zero, += input (-0 becomes +0), pan, gain
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        return epilogue4(add4(z, load_pair(p, m, M)), term_pg(t, j));
    }
    const uint64_t len = t[j].len, t0 = t[j].t0;
    float4 v;
    if (kind == 3) {   // packed 16-bit form, two frames
        const uint32_t TD_GLOBAL* g = reinterpret_cast<const uint32_t TD_GLOBAL*>((const TD_GLOBAL char*)p);
        const uint32_t l32 = (uint32_t)len, i0 = barrett_mod((uint32_t)t0 + m, l32, t[j].magic);
        typedef unsigned int u2v_u __attribute__((ext_vector_type(2), aligned(4)));
        const u2v_u w = *reinterpret_cast<const u2v_u TD_GLOBAL*>(g + i0);   // frame i0 and its successor (wrap frames follow the loop)
        const float2 a = unpack16(w.x, t[j].scale_l, t[j].scale_r), b = unpack16(w.y, t[j].scale_l, t[j].scale_r);
        v = make_float4(a.x, a.y, b.x, b.y);
    } else {
        v = kind == 1 ? loop_pair32(p, (uint32_t)len, t[j].magic, (uint32_t)t0 + m) : loop_pair<uint64_t>(p, len, (t0 + m) % len);
    }
    return zero_tail(epilogue4(v, term_pg(t, j)), m, M);
}
TD_DEV float4 loop_term_pair(TermTab t, uint32_t j, uint32_t m, uint32_t M) {
    return zero_tail(epilogue4(loop_pair32(t[j].p, (uint32_t)t[j].len, t[j].magic, (uint32_t)t[j].t0 + m), term_pg(t, j)), m, M);
}

// InTerm kind 5: an input read THROUGH an Adsr vertex that has this one input (and, magic != 0, a single-input Sum
// stage behind it): the consumer does the vertex' own work per frame pair -- `0.0 + x` (its sum_inputs over one input,
// itself a term of kind 0 .. 4: AdsrVDesc::ins), the envelope (adsr_frame), pan / gain; then the stage's `0.0 + x`, pan,
// gain -- the same f32 operations in the same order as the materialised vertices, without their launches and buffers.
// `len` holds the vertex' AdsrVDesc.
TD_DEV float4 adsr_term_pair(TermTab t, uint32_t j, uint32_t m, uint32_t M) {
    const AdsrVDesc TD_CONST* d = (const AdsrVDesc TD_CONST*)(const TD_CONST char*)(uintptr_t)t[j].len;   // (uniform: scalar loads)
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 x = add4(z, term_pair(term_tab(d->ins), 0u, m, M));   // the vertex' own input: an edge buffer, a stage, a loop source
    float4 v = z;
    if (m < M) {
        // `buf[i] *= vel` (extensions.rs:646-647) with the vertex' gain of this frame from its envelope buffer (k_adsr_env:
        // adsr_vel(), the same f32 operations adsr_frame does; 1.0 where the reference leaves the frame untouched)
        const float2 e = gload2(d->env + m);   // (m is even; the buffer holds frames + 1 gains)
        PanGain pg;
        pg.l_amp = d->pg.l_amp; pg.r_amp = d->pg.r_amp; pg.gain = d->pg.gain; pg.flags = d->pg.flags;
        v = epilogue4(make_float4(x.x * e.x, x.y * e.x, x.z * e.y, x.w * e.y), pg);
        if (t[j].magic) v = epilogue4(add4(z, v), term_pg(t, j));
        v = zero_tail(v, m, M);
    }
    return v;
}

// sum_inputs (extensions.rs:310-319): zero, then += each input in edge order.  Terms are fetched four
// (edge buffers: eight) at a time so that 8-16 x 16 B loads are in flight per lane before the first add;
// the adds themselves stay strictly sequential per element.
template <int MODE>
TD_DEV void sum_terms(TermTab ins, uint32_t k, uint32_t m0, uint32_t m1, uint32_t M, float4& acc0, float4& acc1) {
    uint32_t j = 0;
    if (MODE == TERMS_ADSR1) {   // exactly one term, kind 5
        acc0 = add4(acc0, adsr_term_pair(ins, 0u, m0, M));
        acc1 = add4(acc1, adsr_term_pair(ins, 0u, m1, M));
        return;
    }
    if (MODE == TERMS_WITH_ADSR) {   // one term at a time (the envelope code once in the loop, not once per unrolled slot)
        for (; j < k; ++j) {
            if (ins[j].kind == 5u) {
                acc0 = add4(acc0, adsr_term_pair(ins, j, m0, M));
                acc1 = add4(acc1, adsr_term_pair(ins, j, m1, M));
            } else {
                acc0 = add4(acc0, term_pair(ins, j, m0, M));
                acc1 = add4(acc1, term_pair(ins, j, m1, M));
            }
        }
        return;
    }
    if (MODE == TERMS_ALL_EDGE && k >= 8) {
        // software pipeline over groups of four edge buffers, two register sets (no copies): the eight loads
        // of the next group are in flight while this group's adds retire (counted vmcnt waits)
        float4 a0[4], a1[4], b0[4], b1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const float2* p = ins[u].p; a0[u] = load_pair(p, m0, M); a1[u] = load_pair(p, m1, M); }
        for (; j + 8 <= k; j += 8) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { const float2* p = ins[j + 4 + u].p; b0[u] = load_pair(p, m0, M); b1[u] = load_pair(p, m1, M); }
#pragma unroll
            for (int u = 0; u < 4; ++u) { acc0 = add4(acc0, a0[u]); acc1 = add4(acc1, a1[u]); }
            if (j + 12 <= k) {
#pragma unroll
                for (int u = 0; u < 4; ++u) { const float2* p = ins[j + 8 + u].p; a0[u] = load_pair(p, m0, M); a1[u] = load_pair(p, m1, M); }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { acc0 = add4(acc0, b0[u]); acc1 = add4(acc1, b1[u]); }
        }
        if (j + 4 <= k) {   // one more full group is already loaded in a*
#pragma unroll
            for (int u = 0; u < 4; ++u) { acc0 = add4(acc0, a0[u]); acc1 = add4(acc1, a1[u]); }
            j += 4;
        }
    }
    for (; j + 4 <= k; j += 4) {
        float4 x0[4], x1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == TERMS_ALL_EDGE || MODE == TERMS_EDGE_FEW) { const float2* p = ins[j + u].p; x0[u] = load_pair(p, m0, M); x1[u] = load_pair(p, m1, M); }
            else if (MODE == TERMS_ALL_LOOP32) { x0[u] = loop_term_pair(ins, j + u, m0, M); x1[u] = loop_term_pair(ins, j + u, m1, M); }
            else { x0[u] = term_pair(ins, j + u, m0, M); x1[u] = term_pair(ins, j + u, m1, M); }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { acc0 = add4(acc0, x0[u]); acc1 = add4(acc1, x1[u]); }
    }
    for (; j < k; ++j) {
        acc0 = add4(acc0, term_pair(ins, j, m0, M));
        acc1 = add4(acc1, term_pair(ins, j, m1, M));
    }
}
// ---- packed 16-bit sample form (InTerm kind 3): four consecutive frames per lane ----
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
TD_DEV float2 unpack16(uint32_t w, float sl, float sr) {
    return make_float2((float)(int16_t)(w & 0xFFFFu) * sl, (float)(int16_t)(w >> 16) * sr);
}
// Frames idx .. idx+3 of a looping sample in ONE 16-byte load: the packed form is the loop followed by its own
// first 15 frames (the wide kernels read up to 16 consecutive frames behind one modulo), and global_load_dwordx4
// only needs dword alignment.  (A first version kept four
// phase-shifted copies to make the load 16-byte aligned: four times the footprint in L2 / Infinity Cache for
// nothing -- 0.101 ms against 0.086 ms for the 64-source sum.)
typedef unsigned int u4v_u __attribute__((ext_vector_type(4), aligned(4)));
TD_DEV void loop16_quad(const uint32_t* s, uint32_t len, uint32_t idx, uint32_t out[4]) {
    (void)len;
    const uint32_t TD_GLOBAL* g = reinterpret_cast<const uint32_t TD_GLOBAL*>((const TD_GLOBAL char*)s);
    const u4v_u q = *reinterpret_cast<const u4v_u TD_GLOBAL*>(g + idx);   // dword-aligned global_load_dwordx4
    out[0] = q.x; out[1] = q.y; out[2] = q.z; out[3] = q.w;
}
// all terms kind 3: lane t owns frames m, m+1 (acc0) and m+2, m+3 (acc1) with m = tile + 4t.
// Per value: int16 -> f32, x scale, x pan amplitude, x gain, += -- the reference's roundings in the
// reference's order.  A term whose pan or gain step is skipped (flags) multiplies by 1.0f instead (make_pg
// leaves the unused amplitudes at 1.0f): the values are finite (they come from int16), so x * 1.0f == x
// bit for bit and no select is needed.  L/R pairs are kept as 2-vectors so the multiplies and adds can issue packed.
// Frames at or beyond M pick up garbage-free but meaningless sums; they are zeroed once, after the loop.
TD_DEV f2v cvt16(uint32_t w) { f2v v; v.x = (float)(int16_t)(w & 0xFFFFu); v.y = (float)(int16_t)(w >> 16); return v; }
TD_DEV void sum_terms16(TermTab ins, uint32_t k, uint32_t m, uint32_t M, float4& acc0, float4& acc1) {
    f2v c[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) { c[f].x = 0.f; c[f].y = 0.f; }
    auto add_term = [&](uint32_t j, const uint32_t w[4]) {
        f2v sc, am, gn;   // (the host leaves l_amp / r_amp / gain at 1.0f when their flag is clear)
        sc.x = ins[j].scale_l; sc.y = ins[j].scale_r;
        am.x = ins[j].pg.l_amp; am.y = ins[j].pg.r_amp;
        gn.x = gn.y = ins[j].pg.gain;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            f2v v = cvt16(w[f]) * sc;
            v = v * am;
            v = v * gn;
            c[f] = c[f] + v;
        }
    };
    auto gather = [&](uint32_t j, uint32_t w[4]) {
        const uint32_t len = (uint32_t)ins[j].len;
        loop16_quad(reinterpret_cast<const uint32_t*>(ins[j].p), len, barrett_mod((uint32_t)ins[j].t0 + m, len, ins[j].magic), w);
    };
    uint32_t j = 0;
    for (; j + 4 <= k; j += 4) {
        uint32_t w[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) gather(j + u, w[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) add_term(j + u, w[u]);
    }
    for (; j < k; ++j) {
        uint32_t w[4];
        gather(j, w);
        add_term(j, w);
    }
    acc0 = zero_tail(make_float4(c[0].x, c[0].y, c[1].x, c[1].y), m, M);
    acc1 = zero_tail(make_float4(c[2].x, c[2].y, c[3].x, c[3].y), m + 2u, M);
}

// all terms kind 3, 4 * NQ consecutive frames per lane (NQ 16-byte gathers per source).  More frames per lane =
// more of the timeline resident per XCD at any moment = more of the concurrently running tiles touch the same
// lines of a looping source while they are still in L2.  The packed form carries 15 wrap frames, so one modulo
// per source and lane is enough.
// Which frames a lane owns (round 6).  NQ < 4: 4 NQ consecutive ones from m.  NQ == 4 (BASELINE config 2's fused launch): quad q =
// frames m + 256 q of the wave's 1 024, m = the wave's base + 4 lane -- every load instruction of the wave reads ONE KILOBYTE IN ONE
// PIECE.  With sixteen consecutive frames per lane a lane's four 16-byte loads lie 64 bytes from the next lane's: every load
// instruction touches 64 separate 64-byte pieces and takes a quarter of each, and the lines have to stay in the CU's L1 until the
// fourth instruction has had its quarter (tools/ubench/gather_shape.hip, config 2's 737 MB with the arithmetic taken out: 52 us in
// that shape, 40 us in this one with two sources' loads in flight -- in the old shape a second source in flight LOSES).
// 256 mod len on the scalar unit (len, magic are a source's: uniform): a quad's index = the one before + this, wrapped ONCE, whatever
// the loop's length -- no branch on `len > 256` in the gather (behind a branch the compiler's waits for the loads in flight fall
// back to "all of them")
TD_DEV uint32_t step256(uint32_t len, uint32_t magic) {
    const uint32_t r = 256u - __umulhi(256u, magic) * len;   // (magic = floor(2^32 / len): the quotient is short by 1 at most)
    return r >= len ? r - len : r;
}
template <int NQ>
TD_DEV uint32_t quad_frame(uint32_t m, int q) { return NQ == 4 ? m + 256u * (uint32_t)q : m + 4u * (uint32_t)q; }
template <int NQ>
TD_DEV uint32_t pair_frame(uint32_t m, int p) { return quad_frame<NQ>(m, p >> 1) + 2u * (uint32_t)(p & 1); }
template <int NQ>
TD_DEV void sum_terms16w(TermTab ins, uint32_t k, uint32_t m, uint32_t M, float4 acc[2 * NQ]) {
    f2v c[4 * NQ];
#pragma unroll
    for (int f = 0; f < 4 * NQ; ++f) { c[f].x = 0.f; c[f].y = 0.f; }
    auto add_term = [&](uint32_t j, const uint32_t w[4 * NQ]) {
        f2v sc, am, gn;   // (the host leaves l_amp / r_amp / gain at 1.0f when their flag is clear)
        sc.x = ins[j].scale_l; sc.y = ins[j].scale_r;
        am.x = ins[j].pg.l_amp; am.y = ins[j].pg.r_amp;
        gn.x = gn.y = ins[j].pg.gain;
        // (the three multipliers stay scalar operands of the packed multiplies: copied to vector registers first -- three vector
        // register pairs read per v_pk_mul_f32 -- the launch measures 58.7 -> 59.6 (gain only) -> 63 us (all three), round 6)
#pragma unroll
        for (int f = 0; f < 4 * NQ; ++f) {
            f2v v = cvt16(w[f]) * sc;
            v = v * am;
            v = v * gn;
            c[f] = c[f] + v;
        }
    };
    auto gather = [&](uint32_t j, uint32_t w[4 * NQ]) {
        const uint32_t len = (uint32_t)ins[j].len;
        const uint32_t idx = barrett_mod((uint32_t)ins[j].t0 + m, len, ins[j].magic);
        if (NQ == 4) {
            const uint32_t* p = reinterpret_cast<const uint32_t*>(ins[j].p);
            const uint32_t step = step256(len, ins[j].magic);
            uint32_t i = idx;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                loop16_quad(p, len, i, w + 4 * q);
                i += step;
                i = min(i, i - len);
            }
            return;
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) loop16_quad(reinterpret_cast<const uint32_t*>(ins[j].p), len, idx + 4u * q, w + 4 * q);
    };
    uint32_t j = 0;
    // sources per batch: 16 x 16-byte loads in flight per lane (NQ == 4, round 6: four sources since every load instruction reads whole
    // lines -- in the old shape a second source in flight lost: 1 / 2 / 3 / 4 sources 63.1 / 64.5 / 62.9 / 61.9 us on config 2;
    // with the wrap made branch-free, step256 above, 58.9 us).  Measured and dropped: the same loads as a ROLLING pipeline -- the
    // next sources' loads issued while this one's frames are added, two or four register sets in rotation, records a source ahead --
    // 68 us: the memory side serves a wave's burst of sixteen loads followed by silence faster than a steady trickle.
    // (Also measured and dropped, same instruction mix to the last VALU operation: the four records of a batch as explicit,
    // non-overlapping scalar loads behind ONE wait -- the compiler's own loads pack `t0` into the unused half of `len`'s register
    // pair and wait five times in a row at the head of every batch -- 58.2 -> 69.4 us on the same box, with 64-byte records as
    // well; and the NEXT batch's records asked for behind this batch's vector loads: 112 scalar registers, spills, 68.6 us.  A
    // start-up stagger of the workgroups that share a CU (0 / 0.8 / 1.6 us) moves nothing.)
    constexpr int B = 4;
    for (; j + B <= k; j += B) {
        uint32_t w[B][4 * NQ];
#pragma unroll
        for (int u = 0; u < B; ++u) gather(j + u, w[u]);
#pragma unroll
        for (int u = 0; u < B; ++u) add_term(j + u, w[u]);
    }
    for (; j < k; ++j) { uint32_t w[4 * NQ]; gather(j, w); add_term(j, w); }
#pragma unroll
    for (int q = 0; q < 2 * NQ; ++q) acc[q] = zero_tail(make_float4(c[2 * q].x, c[2 * q].y, c[2 * q + 1].x, c[2 * q + 1].y), pair_frame<NQ>(m, q), M);
}

// the same for all-f32 looping sources (kind 1): 4 * NQ consecutive frames per lane = 2 * NQ 16-byte loads per source
// behind one modulo (bank entries end with 15 wrap frames).  Pan / gain as unconditional multiplies: make_pg leaves the
// unused amplitudes at 1.0f and x * 1.0f == x for every non-NaN x.
template <int NQ>
TD_DEV void sum_terms32w(TermTab ins, uint32_t k, uint32_t m, uint32_t M, float4 acc[2 * NQ]) {
    f4v c[2 * NQ];
#pragma unroll
    for (int q = 0; q < 2 * NQ; ++q) { c[q].x = 0.f; c[q].y = 0.f; c[q].z = 0.f; c[q].w = 0.f; }
    constexpr int B = NQ >= 4 ? 1 : 2;   // sources per batch
    auto gather = [&](uint32_t j, f4v x[2 * NQ]) {
        const uint32_t len = (uint32_t)ins[j].len;
        const uint32_t idx = barrett_mod((uint32_t)ins[j].t0 + m, len, ins[j].magic);
        const float2* p = ins[j].p + idx;
#pragma unroll
        for (int q = 0; q < 2 * NQ; ++q) {
            const float4 v = gload4u(p + 2 * q);
            x[q].x = v.x; x[q].y = v.y; x[q].z = v.z; x[q].w = v.w;
        }
    };
    auto add_term = [&](uint32_t j, const f4v x[2 * NQ]) {
        f4v am, gn;
        am.x = am.z = ins[j].pg.l_amp; am.y = am.w = ins[j].pg.r_amp;
        gn.x = gn.y = gn.z = gn.w = ins[j].pg.gain;
#pragma unroll
        for (int q = 0; q < 2 * NQ; ++q) {
            f4v v = x[q] * am;
            v = v * gn;
            c[q] = c[q] + v;
        }
    };
    uint32_t j = 0;
    for (; j + B <= k; j += B) {
        f4v x[B][2 * NQ];
#pragma unroll
        for (int u = 0; u < B; ++u) gather(j + u, x[u]);
#pragma unroll
        for (int u = 0; u < B; ++u) add_term(j + u, x[u]);
    }
    for (; j < k; ++j) { f4v x[2 * NQ]; gather(j, x); add_term(j, x); }
#pragma unroll
    for (int q = 0; q < 2 * NQ; ++q) acc[q] = zero_tail(make_float4(c[q].x, c[q].y, c[q].z, c[q].w), m + 2u * q, M);
}

TD_DEV void sum_inputs_pairs(const InTerm* ins_generic, uint32_t k, uint32_t term_mode, uint32_t m0, uint32_t m1, uint32_t M,
                             float4& a0, float4& a1) {
    const TermTab ins = term_tab(ins_generic);
    a0 = make_float4(0.f, 0.f, 0.f, 0.f);
    a1 = a0;
    if (term_mode == TERMS_ALL_EDGE) sum_terms<TERMS_ALL_EDGE>(ins, k, m0, m1, M, a0, a1);
    else if (term_mode == TERMS_ALL_LOOP32) sum_terms<TERMS_ALL_LOOP32>(ins, k, m0, m1, M, a0, a1);
    else if (term_mode == TERMS_ADSR1) sum_terms<TERMS_ADSR1>(ins, k, m0, m1, M, a0, a1);
    else if (term_mode == TERMS_WITH_ADSR) sum_terms<TERMS_WITH_ADSR>(ins, k, m0, m1, M, a0, a1);
    else sum_terms<TERMS_MIXED>(ins, k, m0, m1, M, a0, a1);
}

TD_DEV float absmax4(float m, float4 v) {
    // absmaxlen's fold `if a > max {a} else {max}` (sample.rs:12-14): NaNs never win; fmaxf agrees.
    m = fmaxf(m, fabsf(v.x)); m = fmaxf(m, fabsf(v.y));
    m = fmaxf(m, fabsf(v.z)); m = fmaxf(m, fabsf(v.w));
    return m;
}

// ------------------------------------------------------------------------------------------------
// k_sum: Sum vertex / Normalize pass A (k-input sum + per-reference-block peak)
// ------------------------------------------------------------------------------------------------
// tiles_per_block = bl / 1024 when bl is a multiple of the tile, else 0 (generic per-frame peak path).
// One instantiation per term mode (every descriptor of a launch shares it): each keeps only its own summing
// loop, so the register budget -- and with it the waves per SIMD -- is set by that loop alone.
template <int TMODE>
__global__ __launch_bounds__(kThreads) void k_sum(const SumDesc* __restrict__ descs, uint32_t M, uint32_t bl,
                                                  uint32_t tiles_per_block) {
    const SumDesc& d = descs[blockIdx.y];
    // frame mapping: lane t owns pairs tile+2t and tile+512+2t -- or, when every term is a packed 16-bit
    // source, the four consecutive frames tile+4t.. (one 16-byte word of packed frames per source)
    constexpr bool quad_map = TMODE == TERMS_ALL_LOOP16;
    const uint32_t m0 = blockIdx.x * kTileFrames + (quad_map ? 4 : 2) * threadIdx.x;
    const uint32_t m1 = m0 + (quad_map ? 2 : kTileFrames / 2);
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    if (quad_map) sum_terms16(term_tab(d.ins), d.k, m0, M, a0, a1);
    else sum_terms<TMODE>(term_tab(d.ins), d.k, m0, m1, M, a0, a1);
    if (d.mode == 0) {
        store_pair(d.out, m0, M, epilogue4(a0, d.pg));
        store_pair(d.out, m1, M, epilogue4(a1, d.pg));
        return;
    }
    float spec_init = 0.0f;
    if (d.mode == 3) {
        // speculative single pass: every block is assumed to keep the carried max, so the scale is 1 / max for all
        // of them (extensions.rs:323-328 with max_b == max_{b-1}); raw sums never reach memory
        spec_init = d.use_init ? d.init_max : d.state->max;
        const float r = 1.0f / spec_init;
        const float4 s0 = epilogue4(make_float4(a0.x * r, a0.y * r, a0.z * r, a0.w * r), d.pg);
        const float4 s1 = epilogue4(make_float4(a1.x * r, a1.y * r, a1.z * r, a1.w * r), d.pg);
        if (d.out) {
            store_pair(d.out, m0, M, s0);
            store_pair(d.out, m1, M, s1);
        }
        if (d.qmode) {
            store_quant_pair(d.pcm, d.qmode, m0, M, s0, d.amplitude);
            store_quant_pair(d.pcm, d.qmode, m1, M, s1, d.amplitude);
        }
    } else if (d.out) {   // (mode 2 -- a band-pass vertex' input sum -- keeps only the planar copy since round 6)
        store_pair(d.out, m0, M, a0);
        store_pair(d.out, m1, M, a1);
    }
    if (d.mode == 2) {
        if (!quad_map) {
            // planar-in-4 copy: lanes 2u / 2u+1 hold frames 4u..4u+3; the even lane assembles the L word, the
            // odd lane the R word, each with its neighbour's two values (DPP quad_perm [1,0,3,2]) and stores
            // it at its own word address.  The chunk's last, partial four frames too (the buffer is padded to whole fours;
            // frames at or beyond M are written as 0): the planar copy is the ONLY copy the band-pass kernels read.
            auto nb = [](float v) {
                return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
            };
            const bool odd = (threadIdx.x & 1u) != 0u;
            if (m0 >= M) a0 = make_float4(0.f, 0.f, 0.f, 0.f); else if (m0 + 1u >= M) { a0.z = 0.f; a0.w = 0.f; }
            if (m1 >= M) a1 = make_float4(0.f, 0.f, 0.f, 0.f); else if (m1 + 1u >= M) { a1.z = 0.f; a1.w = 0.f; }
            const float n0x = nb(a0.x), n0y = nb(a0.y), n0z = nb(a0.z), n0w = nb(a0.w);
            const float n1x = nb(a1.x), n1y = nb(a1.y), n1z = nb(a1.z), n1w = nb(a1.w);
            const float4 q0 = odd ? make_float4(n0y, n0w, a0.y, a0.w) : make_float4(a0.x, a0.z, n0x, n0z);
            const float4 q1 = odd ? make_float4(n1y, n1w, a1.y, a1.w) : make_float4(a1.x, a1.z, n1x, n1z);
            if ((m0 & ~3u) < M) gstore4(d.out_q4 + m0, q0);
            if ((m1 & ~3u) < M) gstore4(d.out_q4 + m1, q1);
        }
        if (d.rp && !quad_map) {
            // block responses for k_band_spec's warm-up guess (BandRespParam): weights gamma (1 - gamma)^(255 - i) in double
            // by binary powers, the lane's four frames, then a fixed-order sum over the block's 128 lanes
            const BandRespParam& rp = *d.rp;
            const uint32_t i0 = (2u * threadIdx.x) & 255u;      // in-block index of the lane's first frame (even)
            const uint32_t k1 = 254u - i0;                      // exponent of the second frame; the first has k1 + 1
            const bool want_l = rp.gl != 0.0, want_h = rp.gh != 0.0;   // (uniform: a smoother without responses costs nothing here)
            double pl = 1.0, ph = 1.0;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if ((k1 >> j) & 1u) { if (want_l) pl *= rp.ql[j]; if (want_h) ph *= rp.qh[j]; }
            const double wl1 = rp.gl * pl, wl0 = wl1 * rp.ql[0], wh1 = rp.gh * ph, wh0 = wh1 * rp.qh[0];
            // the lane's two frames of each of its two blocks, four chains; then a fixed-order sum over each block's 128 lanes:
            // every wave adds up its 64 lanes on the VALU (a serial walk of 32 threads over LDS partials used to hold the
            // whole workgroup for 64 dependent LDS round trips), 16 threads add the two halves of a block
            double part[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            if (want_l) {
                part[0] = wl0 * (double)a0.x + wl1 * (double)a0.z;
                part[1] = wl0 * (double)a0.y + wl1 * (double)a0.w;
                part[4] = wl0 * (double)a1.x + wl1 * (double)a1.z;
                part[5] = wl0 * (double)a1.y + wl1 * (double)a1.w;
            }
            if (want_h) {
                part[2] = wh0 * (double)a0.x + wh1 * (double)a0.z;
                part[3] = wh0 * (double)a0.y + wh1 * (double)a0.w;
                part[6] = wh0 * (double)a1.x + wh1 * (double)a1.z;
                part[7] = wh0 * (double)a1.y + wh1 * (double)a1.w;
            }
            __shared__ double rh[8][4];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (!((e & 2) ? want_h : want_l)) continue;   // (uniform)
                const double t = wave_sum_to_lane63(part[e]);
                if ((threadIdx.x & 63u) == 63u) rh[e][threadIdx.x >> 6] = t;
            }
            if ((threadIdx.x & 63u) == 63u) {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (!((e & 2) ? want_h : want_l)) rh[e][threadIdx.x >> 6] = 0.0;
            }
            __syncthreads();
            if (threadIdx.x < 16u) {
                const uint32_t blk_in_tile = threadIdx.x >> 2, chain = threadIdx.x & 3u;   // blocks 0, 1: first pair; 2, 3: second pair
                const uint32_t q0 = (blk_in_tile & 1u) * 2u, e = (blk_in_tile >> 1) * 4u + chain;
                const uint32_t blk = blockIdx.x * 4u + blk_in_tile;
                if (blk * 256u < M) rp.resp[4u * blk + chain] = rh[e][q0] + rh[e][q0 + 1u];
            }
        }
        // Liveness of the tile's four 256-frame blocks (threads 0..127 / 128..255 x first / second frame pair):
        // the block's absolute peak, or -1 when every frame of the block is bit-identical (a held constant
        // parks the filter state just like silence does).
        __shared__ float bp[kThreads / 64][2];
        __shared__ uint32_t bc[kThreads / 64][2];
        __shared__ float2 first[4];
        if ((threadIdx.x & 127) == 0) {
            first[(threadIdx.x >> 7)] = make_float2(a0.x, a0.y);        // blocks 0, 1
            first[2 + (threadIdx.x >> 7)] = make_float2(a1.x, a1.y);    // blocks 2, 3
        }
        __syncthreads();
        const float2 f0 = first[threadIdx.x >> 7], f1 = first[2 + (threadIdx.x >> 7)];
        auto same2 = [](float4 a, float2 f) {
            return __float_as_uint(a.x) == __float_as_uint(f.x) && __float_as_uint(a.y) == __float_as_uint(f.y) &&
                   __float_as_uint(a.z) == __float_as_uint(f.x) && __float_as_uint(a.w) == __float_as_uint(f.y);
        };
        float p0 = m0 < M ? absmax4(0.0f, a0) : 0.0f, p1 = m1 < M ? absmax4(0.0f, a1) : 0.0f;
        const bool c0 = __all((m0 + 1 >= M || same2(a0, f0)) ? 1 : 0) != 0, c1 = __all((m1 + 1 >= M || same2(a1, f1)) ? 1 : 0) != 0;
        p0 = wave_max(p0);
        p1 = wave_max(p1);
        if ((threadIdx.x & 63) == 0) {
            bp[threadIdx.x >> 6][0] = p0; bp[threadIdx.x >> 6][1] = p1;
            bc[threadIdx.x >> 6][0] = c0; bc[threadIdx.x >> 6][1] = c1;
        }
        __syncthreads();
        if (threadIdx.x < 4) {
            const uint32_t half = threadIdx.x & 1u, pair = threadIdx.x >> 1;   // block = pair * 2 + half
            const uint32_t blk = blockIdx.x * 4u + pair * 2u + half;
            const bool cst = bc[2 * half][pair] && bc[2 * half + 1][pair];
            if (blk * 256u < M) d.peaks[blk] = cst ? -1.0f : fmaxf(bp[2 * half][pair], bp[2 * half + 1][pair]);
        }
        return;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        d.init_copy[0] = d.use_init ? d.init_max : d.state->max;
        d.init_copy[1] = d.state->scan_max;
    }
    const bool spec = d.mode == 3;
    if (tiles_per_block) {
        float pk = 0.0f;
        if (m0 < M) pk = absmax4(pk, a0);
        if (m1 < M) pk = absmax4(pk, a1);
        pk = wave_max(pk);
        __shared__ float wmax[kThreads / 64];
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = pk;
        __syncthreads();
        if (threadIdx.x == 0) {
            pk = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
            uint32_t b = blockIdx.x / tiles_per_block;
            if (tiles_per_block == 1) d.peaks[b] = pk;
            else atomicMax(reinterpret_cast<unsigned int*>(d.peaks + b), __float_as_uint(pk));  // pk >= 0
            if (spec && pk > spec_init) const_cast<NormState*>(d.state)->violated = 1u;   // the speculation failed
        }
    } else {
        if (spec) {
            const float pk = fmaxf(m0 < M ? absmax4(0.0f, a0) : 0.0f, m1 < M ? absmax4(0.0f, a1) : 0.0f);
            if (pk > spec_init) const_cast<NormState*>(d.state)->violated = 1u;
        }
        // generic block length: per-frame block id, peaks pre-zeroed by the host
        const float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            uint32_t m = (f < 2 ? m0 : m1) + (f & 1);
            if (m < M) {
                float pk = fmaxf(fmaxf(0.0f, fabsf(v[2 * f])), fabsf(v[2 * f + 1]));
                atomicMax(reinterpret_cast<unsigned int*>(d.peaks + m / bl), __float_as_uint(pk));
            }
        }
    }
}

// k_sum for all-packed-loop (or, PACKED = false, all-f32-loop) terms with 4 * NQ frames per lane: Sum vertices, and Normalize pass A when the
// reference block is the 1024-frame tile (a workgroup then covers NQ whole blocks, a block 4 / NQ waves).
template <int NQ, bool PACKED = true>
__global__ __launch_bounds__(kThreads) void k_sum16w(const SumDesc* __restrict__ descs, uint32_t M, uint32_t tag) {
    const SumDesc& d = descs[blockIdx.y];
    // (the lane's frames: quad_frame / pair_frame above -- NQ == 4 with packed sources: four quads 256 frames apart inside the wave's block)
    constexpr int SH = (NQ == 4 && PACKED) ? 4 : 1;   // (the frame shape's tag: 1 = consecutive)
    const uint32_t m = SH == 4 ? blockIdx.x * (kTileFrames * NQ) + (threadIdx.x >> 6) * kTileFrames + 4u * (threadIdx.x & 63u)
                               : blockIdx.x * (kTileFrames * NQ) + 4u * NQ * threadIdx.x;
    // (mode 4: the carried max, read before anything else -- the last tile replaces it once every tile has published)
    const float spec_init_early = d.mode >= 4 ? (d.use_init ? d.init_max : gload1(&d.state->max)) : 0.0f;
    float4 a[2 * NQ];
    if (PACKED) sum_terms16w<NQ>(term_tab(d.ins), d.k, m, M, a);
    else sum_terms32w<NQ>(term_tab(d.ins), d.k, m, M, a);
    if (d.mode == 0) {
#pragma unroll
        for (int q = 0; q < 2 * NQ; ++q) store_pair(d.out, pair_frame<SH>(m, q), M, epilogue4(a[q], d.pg));
        return;
    }
    float pk = 0.0f;   // block peaks are those of the RAW sum
#pragma unroll
    for (int q = 0; q < 2 * NQ; ++q) if (pair_frame<SH>(m, q) < M) pk = absmax4(pk, a[q]);
    float spec_init = 0.0f;
    if (d.mode >= 4) {
        // Single-pass RUNNING-PEAK normalize (fresh renders: `*max = buf_max.max(*max)` block by block, extensions.rs:321-329)
        // for a grid that is resident at once: the tile's block peaks first, the tile's maximum published as one granule,
        // every earlier tile's granule read back (704 tiles: three 8-byte loads per lane, spinning until tagged) -- their
        // maximum with the carried max is the running peak entering this tile.  Then the frames are scaled, panned, gained
        // and quantised straight out of the registers: the raw sums never reach memory and pass B (k_scale) is not launched.
        // A workgroup only waits for LOWER tiles, which the dispatcher starts first in practice but need not: the wait is
        // BOUNDED (modes 4 and 5 alike), a tile that gives up raises `violated` -- and the host-visible word `host_flag` -- and
        // k_norm_fix redoes the vertex the two-pass way from the block peaks stored here.  The engine enqueues that launch
        // right behind this one where anything in the submission reads the vertex' output, and otherwise only when the
        // host-visible word says so (settle(), engine.cpp): nothing in here can trap or hang.
        const float init = spec_init_early;
        __shared__ float wm4[kThreads / 64], pm4[kThreads / 64];
        __shared__ uint32_t bad4;
        const uint32_t wave = threadIdx.x >> 6;
        const float pw = wave_max(pk);
        if ((threadIdx.x & 63) == 0) wm4[wave] = pw;
        if (threadIdx.x == 0) bad4 = 0u;
        __syncthreads();
        constexpr uint32_t wpb = 4 / NQ;   // waves per reference block
        float pb[NQ], T = 0.0f;
#pragma unroll
        for (int b = 0; b < NQ; ++b) {
            float p = wm4[b * wpb];
            for (uint32_t u = 1; u < wpb; ++u) p = fmaxf(p, wm4[b * wpb + u]);
            pb[b] = p;
            T = fmaxf(T, p);
        }
        if (threadIdx.x < (uint32_t)NQ) {
            const uint32_t b = blockIdx.x * NQ + threadIdx.x;
            if (b * kTileFrames < M) d.peaks[b] = threadIdx.x == 0 ? pb[0] : threadIdx.x == 1 ? pb[NQ > 1 ? 1 : 0] : threadIdx.x == 2 ? pb[NQ > 2 ? 2 : 0] : pb[NQ > 3 ? 3 : 0];
        }
        unsigned long long* const sync = d.sync;
        if (threadIdx.x == 0) {
            asm volatile("" ::"v"(init));   // (the carried max has been READ before this tile counts as published: the last tile replaces it)
            granule_store(sync + blockIdx.x, __float_as_uint(T), tag);
        }
        float pm = 0.0f;
        const bool forced = (d.debug & 1u) != 0u && blockIdx.x != 0u;   // (tests: every wait gives up at once)
        const bool ok = (d.debug & 2u) ? true : !forced && for_lower_granules(sync, blockIdx.x, kScanSpinLimitSum,
                                                      [&pm](uint32_t, uint32_t v) { pm = fmaxf(pm, __uint_as_float(v)); }, tag);
        pm = wave_max(pm);
        if ((threadIdx.x & 63) == 0) pm4[wave] = pm;
        if (!ok) bad4 = 1u;
        __syncthreads();
        float run = fmaxf(fmaxf(fmaxf(pm4[0], pm4[1]), fmaxf(pm4[2], pm4[3])), init);   // max_{b-1} entering the tile
        float r_mine = 0.0f;
        const uint32_t my_block = wave / wpb;
#pragma unroll
        for (int b = 0; b < NQ; ++b) {
            run = fmaxf(pb[b], run);   // *max = buf_max.max(*max)
            if ((uint32_t)b == my_block) r_mine = 1.0f / run;
        }
        if (threadIdx.x == 0) {
            NormState* st = const_cast<NormState*>(d.state);
            if (bad4) raise_violated(st, d.host_flag);
            else if (blockIdx.x == gridDim.x - 1u) st->max = run;   // (every earlier tile has read the old value: see above)
        }
        spec_init = -1.0f;   // (no speculation to check below)
#pragma unroll
        for (int q = 0; q < 2 * NQ; ++q) a[q] = epilogue4(make_float4(a[q].x * r_mine, a[q].y * r_mine, a[q].z * r_mine, a[q].w * r_mine), d.pg);
    }
    if (d.mode == 3) {   // speculative single-pass normalize (see SumDesc): scaled, finished frames out of registers
        spec_init = d.use_init ? d.init_max : d.state->max;
        const float r = 1.0f / spec_init;
#pragma unroll
        for (int q = 0; q < 2 * NQ; ++q) a[q] = epilogue4(make_float4(a[q].x * r, a[q].y * r, a[q].z * r, a[q].w * r), d.pg);
    }
    if (d.mode >= 3) {
        if (d.out) {   // (nullptr: nobody reads the f32 form of this output vertex -- engine option "output_f32" 0)
#pragma unroll
            for (int q = 0; q < 2 * NQ; ++q) store_pair(d.out, pair_frame<SH>(m, q), M, a[q]);
        }
        if (d.qmode == 1u) {
            // int16 PCM: the lane's 4 * NQ frames are 16 * NQ contiguous bytes -> one 16-byte store per 4 frames
            uint32_t* o = reinterpret_cast<uint32_t*>(d.pcm);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const uint32_t mm = quad_frame<SH>(m, q);
                if (mm + 3u < M) {
                    const float4 v0 = a[2 * q], v1 = a[2 * q + 1];
                    u4v w;
                    w.x = ((uint32_t)quant16(v0.x, d.amplitude) & 0xFFFFu) | ((uint32_t)quant16(v0.y, d.amplitude) << 16);
                    w.y = ((uint32_t)quant16(v0.z, d.amplitude) & 0xFFFFu) | ((uint32_t)quant16(v0.w, d.amplitude) << 16);
                    w.z = ((uint32_t)quant16(v1.x, d.amplitude) & 0xFFFFu) | ((uint32_t)quant16(v1.y, d.amplitude) << 16);
                    w.w = ((uint32_t)quant16(v1.z, d.amplitude) & 0xFFFFu) | ((uint32_t)quant16(v1.w, d.amplitude) << 16);
                    *reinterpret_cast<u4v TD_GLOBAL*>((TD_GLOBAL char*)(o + mm)) = w;
                } else {
                    store_quant_pair(d.pcm, 1u, mm, M, a[2 * q], d.amplitude);
                    store_quant_pair(d.pcm, 1u, mm + 2u, M, a[2 * q + 1], d.amplitude);
                }
            }
        } else if (d.qmode) {
#pragma unroll
            for (int q = 0; q < 2 * NQ; ++q) store_quant_pair(d.pcm, d.qmode, pair_frame<SH>(m, q), M, a[q], d.amplitude);
        }
    } else {
#pragma unroll
        for (int q = 0; q < 2 * NQ; ++q) store_pair(d.out, pair_frame<SH>(m, q), M, a[q]);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        d.init_copy[0] = d.mode >= 4 ? spec_init_early : (d.use_init ? d.init_max : d.state->max);
        d.init_copy[1] = d.state->scan_max;
    }
    if (d.mode >= 4) return;   // (block peaks stored above)
    pk = wave_max(pk);
    __shared__ float wm[kThreads / 64];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = pk;
    __syncthreads();
    // wave w covers frames [256 NQ w, 256 NQ (w + 1)) of the tile: block (of 1024) = w * NQ / 4
    if (threadIdx.x < (uint32_t)NQ) {
        constexpr uint32_t wpb = 4 / NQ;   // waves per block (NQ = 1, 2, 4)
        float p = wm[threadIdx.x * wpb];
        for (uint32_t u = 1; u < wpb; ++u) p = fmaxf(p, wm[threadIdx.x * wpb + u]);
        const uint32_t b = blockIdx.x * NQ + threadIdx.x;
        if (b * kTileFrames < M) {
            d.peaks[b] = p;
            if (d.mode == 3 && p > spec_init) const_cast<NormState*>(d.state)->violated = 1u;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_norm1: Normalize in ONE launch for any input terms (SumDesc mode 5), reference block = the 1024-frame tile
// ------------------------------------------------------------------------------------------------
// The same scheme as k_sum16w's mode 5 for the narrow forms: a workgroup sums TPW consecutive tiles (1, 2 or 4: chosen by the
// host so that the whole grid is resident at once -- the waits below then need no way out), publishes the largest of their
// block peaks as one granule, reads every earlier workgroup's, and scales / pans / gains / quantises its frames out of the
// registers with the running peak  max_b = peak_b.max(max_{b-1})  (extensions.rs:321-329): same f32 operations as k_sum
// mode 1 + k_scale, one launch instead of two and no raw-sum round trip.
template <int TMODE, int TPW>
__global__ __launch_bounds__(kThreads) void k_norm1(const SumDesc* __restrict__ descs, uint32_t M, uint32_t n_tiles, uint32_t tag) {
    const SumDesc& d = descs[blockIdx.y];
    // (the carried max, read before anything else: the last workgroup replaces it once every workgroup has published)
    const float init = d.use_init ? d.init_max : gload1(&d.state->max);
    constexpr bool quad_map = TMODE == TERMS_ALL_LOOP16;
    const uint32_t wave = threadIdx.x >> 6;
    float4 a[2 * TPW];
    uint32_t mm[2 * TPW];
    __shared__ float wm[TPW][kThreads / 64], pm4[kThreads / 64];
    __shared__ uint32_t bad1;
    if (threadIdx.x == 0) bad1 = 0u;   // (read after two barriers)
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        const uint32_t tile = blockIdx.x * (uint32_t)TPW + (uint32_t)u;
        const uint32_t m0 = tile * kTileFrames + (quad_map ? 4 : 2) * threadIdx.x;
        const uint32_t m1 = m0 + (quad_map ? 2 : kTileFrames / 2);
        mm[2 * u] = m0;
        mm[2 * u + 1] = m1;
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
        if (tile < n_tiles) {   // (uniform)
            if (quad_map) sum_terms16(term_tab(d.ins), d.k, m0, M, a0, a1);
            else sum_terms<TMODE>(term_tab(d.ins), d.k, m0, m1, M, a0, a1);
        }
        a[2 * u] = a0;
        a[2 * u + 1] = a1;
        float pk = 0.0f;   // block peaks are those of the RAW sum
        if (m0 < M) pk = absmax4(pk, a0);
        if (m1 < M) pk = absmax4(pk, a1);
        pk = wave_max(pk);
        if ((threadIdx.x & 63) == 0) wm[u][wave] = pk;
    }
    __syncthreads();
    float pb[TPW], T = 0.0f;
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        pb[u] = fmaxf(fmaxf(wm[u][0], wm[u][1]), fmaxf(wm[u][2], wm[u][3]));
        T = fmaxf(T, pb[u]);
    }
    if (threadIdx.x < (uint32_t)TPW) {
        const uint32_t tile = blockIdx.x * (uint32_t)TPW + threadIdx.x;
        float p = pb[0];
#pragma unroll
        for (int u = 1; u < TPW; ++u) if (threadIdx.x == (uint32_t)u) p = pb[u];
        if (tile < n_tiles) d.peaks[tile] = p;
    }
    if (threadIdx.x == 0) {
        asm volatile("" ::"v"(init));   // (the carried max has been READ before this workgroup counts as published)
        granule_store(d.sync + blockIdx.x, __float_as_uint(T), tag);
        if (blockIdx.x == 0) {
            d.init_copy[0] = init;
            d.init_copy[1] = d.state->scan_max;
        }
    }
    float pm = 0.0f;
    // (bounded, as in k_sum16w: a workgroup that gives up raises `violated` and k_norm_fix redoes the vertex)
    const bool forced = (d.debug & 1u) != 0u && blockIdx.x != 0u;   // (tests: every wait gives up at once)
    const bool ok = !forced && for_lower_granules(d.sync, blockIdx.x, kScanSpinLimitSum, [&pm](uint32_t, uint32_t v) { pm = fmaxf(pm, __uint_as_float(v)); }, tag);
    pm = wave_max(pm);
    if ((threadIdx.x & 63) == 0) pm4[wave] = pm;
    if (!ok) bad1 = 1u;
    __syncthreads();
    float run = fmaxf(fmaxf(fmaxf(pm4[0], pm4[1]), fmaxf(pm4[2], pm4[3])), init);   // max_{b-1} entering the first tile
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
        run = fmaxf(pb[u], run);   // *max = buf_max.max(*max)
        const float r = 1.0f / run;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t m = mm[2 * u + h];
            if (m < M) {
                float4 v = a[2 * u + h];
                v = epilogue4(make_float4(v.x * r, v.y * r, v.z * r, v.w * r), d.pg);
                if (d.out) store_pair(d.out, m, M, v);
                if (d.qmode) store_quant_pair(d.pcm, d.qmode, m, M, v, d.amplitude);
            }
        }
    }
    if (threadIdx.x == 0) {
        NormState* st = const_cast<NormState*>(d.state);
        if (bad1) raise_violated(st, d.host_flag);
        else if (blockIdx.x == gridDim.x - 1u) st->max = run;   // (every earlier workgroup has read the old value)
    }
}

// ------------------------------------------------------------------------------------------------
// k_scale: Normalize pass B (running peak -> scale by 1/max, epilogue, optional fused quantise)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_scale(const ScaleDesc* __restrict__ descs, uint32_t M, uint32_t bl,
                                                    uint32_t nb, int is_scan) {
    const ScaleDesc& d = descs[blockIdx.y];
    const uint32_t tile0 = blockIdx.x * kTileFrames;
    const uint32_t m0 = tile0 + 2 * threadIdx.x;
    const uint32_t m1 = m0 + kTileFrames / 2;
    const uint32_t b_lo = tile0 / bl;
    const float init = d.init_copy[0];
    // the tile's own frames first: their loads fly while the peak table is reduced
    const float4 in0 = load_pair(d.buf, m0, M), in1 = load_pair(d.buf, m1, M);
    // max of the peaks of all blocks before this tile's first block (identity 0: peaks are >= 0, never NaN)
    __shared__ float wmax[kThreads / 64];
    float p = 0.0f;
    for (uint32_t b = threadIdx.x; b < b_lo; b += 4u * kThreads) {   // four independent loads in flight per lane
        const float p0 = d.peaks[b];
        const float p1 = b + kThreads < b_lo ? d.peaks[b + kThreads] : 0.0f;
        const float p2 = b + 2u * kThreads < b_lo ? d.peaks[b + 2u * kThreads] : 0.0f;
        const float p3 = b + 3u * kThreads < b_lo ? d.peaks[b + 3u * kThreads] : 0.0f;
        p = fmaxf(fmaxf(fmaxf(p0, p1), fmaxf(p2, p3)), p);
    }
    p = wave_max(p);
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = p;
    __syncthreads();
    const float before = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    const float r_stale = 1.0f / init;
    // the common shape -- the tile IS the reference block -- has one scale for the whole workgroup
    const bool one_block = bl == (uint32_t)kTileFrames;
    const float r_tile = one_block ? (is_scan ? r_stale : 1.0f / fmaxf(d.peaks[b_lo], b_lo ? fmaxf(before, init) : init)) : 0.0f;
    // 1.0 / max for the block holding frame m:  max_b = peak_b.max(max_{b-1}),  max_{-1} = init
    auto rscale_of = [&](uint32_t m) -> float {
        if (is_scan) return r_stale;
        float run = b_lo ? fmaxf(before, init) : init;
        const uint32_t b = m / bl;
        for (uint32_t bb = b_lo; bb <= b; ++bb) run = fmaxf(d.peaks[bb], run);
        return 1.0f / run;
    };
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint32_t m = h ? m1 : m0;
        if (m < M) {
            float4 v = h ? in1 : in0;
            const float r0 = one_block ? r_tile : rscale_of(m);
            const float r1 = one_block ? r_tile : ((m + 1 < M) ? rscale_of(m + 1) : r0);
            v = epilogue4(make_float4(v.x * r0, v.y * r0, v.z * r1, v.w * r1), d.pg);
            if (!d.pcm_only) store_pair(d.buf, m, M, v);
            if (d.qmode) store_quant_pair(d.pcm, d.qmode, m, M, v, d.amplitude);
        }
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        float all = before;   // max over every block peak of the chunk
        for (uint32_t bb = b_lo; bb < nb; ++bb) all = fmaxf(d.peaks[bb], all);
        if (is_scan) {
            d.state->max = init;
            d.state->scan_max = fmaxf(all, d.init_copy[1]);   // *scan_max = buf_max.max(*scan_max)
        } else {
            d.state->max = fmaxf(all, init);
            d.state->scan_max = d.init_copy[1];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_norm_fix: second half of the speculative single-pass normalize (SumDesc mode 3)
// ------------------------------------------------------------------------------------------------
// Nothing to do -- one scalar load per workgroup -- unless some block peak exceeded the carried max.  Then every
// tile whose running max differs from the carried one is redone the two-pass way (sum_inputs again, the exact
// running max of its blocks from the peak table, scale, epilogue, quantise); the workgroup that finishes last
// stores the new carried max and re-arms the flag.
__global__ __launch_bounds__(kThreads) void k_norm_fix(const SumDesc* __restrict__ descs, uint32_t M, uint32_t bl, uint32_t nb) {
    const SumDesc& d = descs[blockIdx.y];
    NormState* st = const_cast<NormState*>(d.state);
    if ((d.mode != 3u && d.mode != 4u && d.mode != 5u) || st->violated == 0u) return;   // the normal case: the whole launch is a few hundred one-load workgroups
    const float init = d.init_copy[0];
    __shared__ float wmax[kThreads / 64];
    const uint32_t n_tiles = (M + kTileFrames - 1) / kTileFrames;
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint32_t tile0 = tile * kTileFrames;
        const uint32_t m0 = tile0 + 2 * threadIdx.x, m1 = m0 + kTileFrames / 2;
        const uint32_t b_lo = tile0 / bl;
        const uint32_t b_hi = min((min(tile0 + (uint32_t)kTileFrames, M) - 1u) / bl, nb - 1u);
        float p = 0.0f;
        for (uint32_t b = threadIdx.x; b < b_lo; b += kThreads) p = fmaxf(d.peaks[b], p);
        p = wave_max(p);
        __syncthreads();   // (wmax of the previous tile has been read by everyone)
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = p;
        __syncthreads();
        const float before = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        const float upto = b_lo ? fmaxf(before, init) : init;      // running max entering the tile's first block
        float last = upto;
        for (uint32_t bb = b_lo; bb <= b_hi; ++bb) last = fmaxf(d.peaks[bb], last);
        if (!(last > init)) continue;   // (uniform) every block of this tile was scaled by 1 / init: already right
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
        if (d.term_mode == TERMS_ADSR1) sum_terms<TERMS_ADSR1>(term_tab(d.ins), d.k, m0, m1, M, a0, a1);
        else if (d.term_mode == TERMS_WITH_ADSR) sum_terms<TERMS_WITH_ADSR>(term_tab(d.ins), d.k, m0, m1, M, a0, a1);
        else sum_terms<TERMS_MIXED>(term_tab(d.ins), d.k, m0, m1, M, a0, a1);
        auto rscale_of = [&](uint32_t m) -> float {
            float run = upto;
            const uint32_t b = m / bl;
            for (uint32_t bb = b_lo; bb <= b; ++bb) run = fmaxf(d.peaks[bb], run);
            return 1.0f / run;
        };
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t m = h ? m1 : m0;
            if (m < M) {
                float4 v = h ? a1 : a0;
                const float r0 = rscale_of(m), r1 = (m + 1 < M) ? rscale_of(m + 1) : r0;
                v = epilogue4(make_float4(v.x * r0, v.y * r0, v.z * r1, v.w * r1), d.pg);
                if (d.out) store_pair(d.out, m, M, v);
                if (d.qmode) store_quant_pair(d.pcm, d.qmode, m, M, v, d.amplitude);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // (the flag was read above, before this increment: the workgroup that draws the last ticket is the last reader)
        const uint32_t t = __hip_atomic_fetch_add(&st->ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (t == gridDim.x - 1u) {
            float all = init;
            for (uint32_t bb = 0; bb < nb; ++bb) all = fmaxf(d.peaks[bb], all);
            st->max = all;                       // *max = buf_max.max(*max) over every block of the chunk
            st->ticket = 0u;
            st->violated = 0u;
        }
    }
}

__global__ __launch_bounds__(kThreads) void k_quantise(const QuantDesc* __restrict__ descs, uint32_t M) {
    const QuantDesc& d = descs[blockIdx.y];
    const uint32_t m0 = blockIdx.x * kTileFrames + 2 * threadIdx.x;
    const uint32_t m1 = m0 + kTileFrames / 2;
    if (m0 < M) store_quant_pair(d.pcm, d.qmode, m0, M, load_pair(d.in, m0, M), d.amplitude);
    if (m1 < M) store_quant_pair(d.pcm, d.qmode, m1, M, load_pair(d.in, m1, M), d.amplitude);
}

// ------------------------------------------------------------------------------------------------
// k_sample_loop: out[m] = sample[(t0 + m) % len]   (extensions.rs:331-341)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_sample_loop(const LoopDesc* __restrict__ descs, uint32_t M) {
    const LoopDesc& d = descs[blockIdx.y];
    const uint32_t m0 = blockIdx.x * kTileFrames + 2 * threadIdx.x;
    const uint32_t m1 = m0 + kTileFrames / 2;
    const float4 v0 = gather_loop_pair(d.sample, d.len, d.t0, d.magic, m0);
    const float4 v1 = gather_loop_pair(d.sample, d.len, d.t0, d.magic, m1);
    store_pair(d.out, m0, M, epilogue4(v0, d.pg));
    store_pair(d.out, m1, M, epilogue4(v1, d.pg));
}

// ------------------------------------------------------------------------------------------------
// k_sample_multi (extensions.rs:344-381)
// ------------------------------------------------------------------------------------------------
TD_DEV float2 multi_frame(const MultiDesc& d, int64_t m) {
    // live voices: origin in (m - len, m]; hits are sorted by origin (onset order = deque order).  The per-tile
    // table gives the first candidate for the tile's first frame; from there it is a short walk, no search.
    const int64_t lo_key = m - (int64_t)d.len;
    uint32_t first = d.tile_first[(uint32_t)m / kTileFrames];
    while (first < d.n_hits && d.hits[first].origin <= lo_key) ++first;
    uint32_t last = first;
    while (last < d.n_hits && d.hits[last].origin <= m) ++last;
    float2 acc = make_float2(0.0f, 0.0f);
    for (uint32_t j = first; j < last; ++j) {
        const MultiHit h = d.hits[j];
        int64_t pos = m - h.origin;
        if (pos < 0) pos = 0;
        const float2 s = d.sample[pos];
        acc.x += s.x * h.vel;
        acc.y += s.y * h.vel;
    }
    return acc;
}
__global__ __launch_bounds__(kThreads) void k_sample_multi(const MultiDesc* __restrict__ descs, uint32_t M) {
    const MultiDesc& d = descs[blockIdx.y];
    const uint32_t m0 = blockIdx.x * kTileFrames + 2 * threadIdx.x;
    const uint32_t m1 = m0 + kTileFrames / 2;
    if (m0 < M) {
        float2 a = multi_frame(d, m0), b = multi_frame(d, (int64_t)m0 + 1);
        store_pair(d.out, m0, M, epilogue4(make_float4(a.x, a.y, b.x, b.y), d.pg));
    }
    if (m1 < M) {
        float2 a = multi_frame(d, m1), b = multi_frame(d, (int64_t)m1 + 1);
        store_pair(d.out, m1, M, epilogue4(make_float4(a.x, a.y, b.x, b.y), d.pg));
    }
}

// ------------------------------------------------------------------------------------------------
// k_sample_lerp (extensions.rs:384-421)
// ------------------------------------------------------------------------------------------------
TD_DEV float2 lerp_frame(const LerpDesc& d, int64_t m) {
    uint32_t lo = d.tile_first[(uint32_t)m / kTileFrames];   // number of entries with key <= m (>= 2: the carried pair)
    while (lo < d.n_hits && d.hits[lo].key <= m) ++lo;
    const LerpHit p = d.hits[lo - 1];
    const int64_t last = (int64_t)d.len - 1;
    int64_t ppos = m - p.origin;
    ppos = ppos < 0 ? 0 : ppos;
    ppos = ppos > last ? last : ppos;
    const float2 ps = d.sample[ppos];
    float l = ps.x * p.vel, r = ps.y * p.vel;
    const int64_t since = m - p.fade;            // frames since countdown := lerp_len
    if (since >= 0 && since < (int64_t)d.lerp_len) {
        const uint32_t countdown = d.lerp_len - 1u - (uint32_t)since;   // after the decrement
        const float t = (float)countdown / (float)d.lerp_len;
        const LerpHit g = d.hits[lo - 2];
        int64_t gpos = m - g.origin;
        gpos = gpos < 0 ? 0 : gpos;
        gpos = gpos > last ? last : gpos;
        const float2 gs = d.sample[gpos];
        const float gl = gs.x * g.vel, gr = gs.y * g.vel;
        l = gl * t + l * (1.0f - t);
        r = gr * t + r * (1.0f - t);
    }
    return make_float2(l, r);
}
TD_DEV void sample_lerp_block(const LerpDesc& d, uint32_t bx, uint32_t M) {
    const uint32_t m0 = bx * kTileFrames + 2 * threadIdx.x;
    const uint32_t m1 = m0 + kTileFrames / 2;
    if (m0 < M) {
        float2 a = lerp_frame(d, m0), b = lerp_frame(d, (int64_t)m0 + 1);
        store_pair(d.out, m0, M, epilogue4(make_float4(a.x, a.y, b.x, b.y), d.pg));
    }
    if (m1 < M) {
        float2 a = lerp_frame(d, m1), b = lerp_frame(d, (int64_t)m1 + 1);
        store_pair(d.out, m1, M, epilogue4(make_float4(a.x, a.y, b.x, b.y), d.pg));
    }
}
__global__ __launch_bounds__(kThreads) void k_sample_lerp(const LerpDesc* __restrict__ descs, uint32_t M) {
    sample_lerp_block(descs[blockIdx.y], blockIdx.x, M);
}

// envelope math: adsr_math.h (shared with the host event compiler)

// Interval of frame m: start from the tile's first interval (host-made table, one scalar load) and walk
// forward -- a tile holds its block-start interval plus a handful of event intervals, so the walk is short
// and replaces a 13-step dependent binary search per frame.
TD_DEV uint32_t find_interval(const IntervalTab& tab, uint32_t m) {
    const uint32_t TD_GLOBAL* s = reinterpret_cast<const uint32_t TD_GLOBAL*>((const TD_GLOBAL char*)tab.istart);
    uint32_t it = *reinterpret_cast<const uint32_t TD_GLOBAL*>((const TD_GLOBAL char*)(tab.tile_first + m / kTileFrames));
    while (it + 1u < tab.n_int && s[it + 1u] <= m) ++it;
    return it;
}

constexpr float kPi = 3.14159274101257324f;   // core::f32::consts::PI

// sin of an f32 argument of any size, for the tolerance-class kernels (debug_sine, synth: <= 1e-6 RMS against the
// oracle's glibc sinf).  The argument itself is rounded exactly like the reference rounds it (`time * hz * 2.0 * PI`,
// extensions.rs:450,501 -- it reaches ~1e6 rad, where an f32 ulp is 0.06 rad: that rounding IS the signal).  Range
// reduction by half turns in f32 with two fused steps: n = rint(arg / pi); arg - n * fl(pi) is exact in one FMA (the
// product has 43 bits, the difference is a multiple of ulp(fl(pi)) below 2), the second FMA adds n * (fl(pi) - pi);
// sin(arg) = (-1)^n sin(r), |r| <= pi/2 (+ 0.2 where rint() saw a misrounded quotient, arguments beyond 1e6), and the
// quarter is a degree-11 odd polynomial (truncation 6e-8 at pi/2).  Measured against sin() in double over [0, 2e6]:
// max 3.3e-7, RMS 2.8e-8.  14 instructions, all of them f32 -- and in the two-frame form below all but the rint /
// convert / sign steps are packed (v_pk_mul_f32, v_pk_fma_f32: both frames of a lane in one issue slot).
typedef float f2 __attribute__((ext_vector_type(2)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
TD_DEV f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
TD_DEV f2 sin_any2(f2 arg) {
    const f2 t = arg * 0.318309886f;   // half turns
    f2 n;
    n.x = __builtin_rintf(t.x);
    n.y = __builtin_rintf(t.y);
    f2 r = fma2(-n, (f2)(3.14159274f), arg);
    r = fma2(-n, (f2)(-8.74227766e-8f), r);
    // the sign (-1)^n as a factor 1 - 4 fract(n / 2) -- exact, like flipping the sign bit, but two packed operations and two
    // v_fract instead of two conversions, two shifts and two xors, and the product is a canonical float: the min / med3 that
    // follow it in the oscillators need no canonicalising v_max in front (round 6: 95 -> 88 VALU per voice and wave)
    const f2 h = n * 0.5f;
    f2 fr;
    fr.x = __builtin_amdgcn_fractf(h.x);
    fr.y = __builtin_amdgcn_fractf(h.y);
    const f2 sgn = fma2(fr, (f2)(-4.0f), (f2)(1.0f));
    const f2 r2 = r * r;
    f2 p = (f2)(-2.5052108385441718775e-8f);                 // -1/11!
    p = fma2(p, r2, (f2)(2.7557319223985890653e-6f));        //  1/9!
    p = fma2(p, r2, (f2)(-1.9841269841269841270e-4f));       // -1/7!
    p = fma2(p, r2, (f2)(8.3333333333333333333e-3f));        //  1/5!
    p = fma2(p, r2, (f2)(-1.6666666666666666667e-1f));       // -1/3!
    const f2 q = fma2(r, r2 * p, r);
    return q * sgn;
}
TD_DEV float sin_any(float arg) { return sin_any2((f2)(arg)).x; }
// The same for arguments below 2e6 rad (SynthDesc::small_args: the host has looked at the chunk's last frame and the tables'
// largest hz; the rounding trick itself holds to 2^22 half turns, 1.3e7 rad: the bound is the polynomial's, below).  n = rint(arg / pi) comes out of ONE fused step: arg / pi + 1.5 * 2^23 is rounded to an integer by
// the addition itself, and the sum's last mantissa bit is n's parity -- shifted to the top and ADDED to r's pattern it flips r's
// sign (v_lshl_add_u32), in front of the odd polynomial, whose result is canonical again.  13 instructions per frame pair
// instead of 18 (round 6; same reduction, a polynomial one degree-step shorter: below).
TD_DEV f2 sin_small2(f2 arg) {
    const f2 big = (f2)(12582912.0f);
    const f2 t = fma2(arg, (f2)(0.318309886f), big);
    const f2 n = t - big;
    f2 r = fma2(-n, (f2)(3.14159274f), arg);
    r = fma2(-n, (f2)(-8.74227766e-8f), r);
    const f2 r2 = r * r;
    f2 rs;
    rs.x = __uint_as_float((__float_as_uint(t.x) << 31) + __float_as_uint(r.x));
    rs.y = __uint_as_float((__float_as_uint(t.y) << 31) + __float_as_uint(r.y));
    // (degree 9, the closest to sin in the maximum norm on |r| <= pi/2 + 0.1: tools/sin_minimax.py -- 9e-9; the whole function
    // against sin() in double over [0, 2e6]: max 1.2e-7, RMS 2.2e-8.  fl(1 / pi) is short by 4e-8 of itself: at 2e6 rad the
    // quotient is off by 0.026 and |r| reaches pi/2 + 0.08 -- the host's bound.  sin_any2's product-then-rint quotient is off by
    // up to half a turn at 1e7 rad and keeps the Taylor polynomial, which degrades more gently outside its range.)
    f2 p = (f2)(2.580226009740727e-06f);
    p = fma2(p, r2, (f2)(-0.00019797123968601227f));
    p = fma2(p, r2, (f2)(0.008332878351211548f));
    p = fma2(p, r2, (f2)(-0.16666650772094727f));
    return fma2(rs, r2 * p, rs);
}

// sin of an f32 argument EXACTLY as glibc's sinf returns it (engine option "sine_mode" 1).  The reference's `f32::sin`
// (extensions.rs:450,501) is libm's sinf; glibc's (2.28 and later: sysdeps/ieee754/flt-32/s_sinf.c + sincosf.h, the ARM
// optimized-routines algorithm) is a short computation in IEEE double -- |y| < pi/4: an odd polynomial; |y| < 120: n =
// round(y 2/pi) by a 2^24-scaled multiply, x = y - n pi/2; beyond: 2/pi from a 24-word table times the mantissa in 64-bit
// integers, the top two bits n, the rest x pi / 2^63 -- then sin or cos polynomial by n & 1, sign by n & 2.  Restated here
// operation for operation, with a fused multiply-add exactly where the x86-64 build of glibc has one (its FMA variant, what
// an EPYC runs; every `a + b * c` of the source): tools/sinf_restate.c compares the same sequence with the host's sinf over ALL
// 4 278 190 080 finite floats -- 0 differ (without the fusing: 12).  Double arithmetic on the device is IEEE, the conversions
// round to nearest even: the kernels' oscillators then carry the oracle's bits.  ~45 instructions, half of them f64 or 64-bit
// integer -- six times sin_any2's cost per frame: the mode is for byte parity, not for the bench.
__device__ const uint32_t kInvPio4[24] = {0xa2u, 0xa2f9u, 0xa2f983u, 0xa2f9836eu, 0xf9836e4eu, 0x836e4e44u, 0x6e4e4415u, 0x4e441529u,
                                          0x441529fcu, 0x1529fc27u, 0x29fc2757u, 0xfc2757d1u, 0x2757d1f5u, 0x57d1f534u, 0xd1f534ddu, 0xf534ddc0u,
                                          0x34ddc0dbu, 0xddc0db62u, 0xc0db6295u, 0xdb629599u, 0x6295993cu, 0x95993c43u, 0x993c4390u, 0x3c439041u};
TD_DEV float sin_glibc_poly(double x, double x2, double csign, int n) {
    if ((n & 1) == 0) {
        const double x3 = x * x2;
        const double s1 = __builtin_fma(x2, -0x1.994eb3774cf24p-13, 0x1.1107605230bc4p-7);
        const double x7 = x3 * x2;
        const double s = __builtin_fma(x3, -0x1.555545995a603p-3, x);
        return (float)__builtin_fma(x7, s1, s);
    }
    // (the second row of glibc's table is the first with the cosine coefficients negated: csign = -1)
    const double x4 = x2 * x2;
    const double c2 = __builtin_fma(x2, csign * 0x1.99343027bf8c3p-16, csign * -0x1.6c087e89a359dp-10);
    const double c1 = __builtin_fma(x2, csign * -0x1.ffffffd0c621cp-2, csign * 0x1p0);
    const double x6 = x4 * x2;
    const double c = __builtin_fma(x4, csign * 0x1.55553e1068f19p-5, c1);
    return (float)__builtin_fma(x6, c2, c);
}
TD_DEV float sin_glibc(float y) {
    const uint32_t yi = __float_as_uint(y);
    const uint32_t top = (yi >> 20) & 0x7ffu;   // abstop12
    double x = (double)y;
    if (top < 0x3f4u) {                          // |y| < pi/4  (abstop12(0x1.921FB6p-1f))
        if (top < 0x398u) return y;              // |y| < 2^-12
        return sin_glibc_poly(x, x * x, 1.0, 0);
    }
    if (top < 0x42fu) {                          // |y| < 120
        const double r = x * 0x1.45F306DC9C883p+23;
        const int n = ((int)r + 0x800000) >> 24;
        x = __builtin_fma(-(double)n, 0x1.921FB54442D18p0, x);
        const double s = ((n + 1) & 2) ? -1.0 : 1.0;   // sign[n & 3] = {1, -1, -1, 1}
        return sin_glibc_poly(x * s, x * x, (n & 2) ? -1.0 : 1.0, n);
    }
    if (top < 0x7f8u) {                          // finite
        const uint32_t sign = yi >> 31;
        const uint32_t* arr = kInvPio4 + ((yi >> 26) & 15u);
        const uint32_t shift = (yi >> 23) & 7u;
        uint32_t xi = ((yi & 0xffffffu) | 0x800000u) << shift;
        uint64_t res0 = (uint64_t)(uint32_t)(xi * arr[0]);
        const uint64_t res1 = (uint64_t)xi * arr[4], res2 = (uint64_t)xi * arr[8];
        res0 = (res2 >> 32) | (res0 << 32);
        res0 += res1;
        const uint64_t nn = (res0 + (1ull << 61)) >> 62;
        res0 -= nn << 62;
        x = (double)(int64_t)res0 * 0x1.921FB54442D18p-62;
        const int n = (int)nn, ns = n + (int)sign;
        const double s = ((ns + 1) & 2) ? -1.0 : 1.0;
        return sin_glibc_poly(x * s, x * x, (ns & 2) ? -1.0 : 1.0, n);
    }
    return y - y;                                // inf, NaN -> NaN
}
// x / y for the same kernels: v_rcp_f32 + multiply (1 ulp) instead of the 10-instruction IEEE division.  0 / 0 is still
// NaN and t / 0 still +-inf (quirk Q6's cases).
TD_DEV float fdiv_fast(float x, float y) { return x * __builtin_amdgcn_rcpf(y); }
TD_DEV float ads_internal_fast(const AdsrConfD& c, float t) {   // adsr.rs:46-60
    if (t <= c.attack_sec) return lerpf(c.std_vel, c.attack_vel, fdiv_fast(t, c.attack_sec));
    if (t <= c.attack_sec + c.decay_sec) return lerpf(c.attack_vel, c.decay_vel, fdiv_fast(t - c.attack_sec, c.decay_sec));
    if (t <= c.attack_sec + c.decay_sec + c.sustain_sec)
        return lerpf(c.decay_vel, c.sustain_vel, fdiv_fast(t - c.attack_sec - c.decay_sec, c.sustain_sec));
    return -1000.0f;
}
TD_DEV float apply_ads_fast(const AdsrConfD& c, float t) {
    const float res = ads_internal_fast(c, t);
    return res <= -1.0f ? c.sustain_vel : res;
}
TD_DEV float apply_r_rt_fast(const AdsrConfD& c, float t, float rt) {   // adsr.rs:71-73, 89-92
    return lerpf(apply_ads_fast(c, rt), c.release_vel, fminf(fdiv_fast(t, c.release_sec), 1.0f));
}

// ------------------------------------------------------------------------------------------------
// k_debug_sine (extensions.rs:423-457)
// ------------------------------------------------------------------------------------------------
TD_DEV float sine_frame(const SineDesc& d, uint32_t m) {
    const uint32_t it = find_interval(d.tab, m);
    const uint32_t v0 = d.tab.ivoff[it], v1 = d.tab.ivoff[it + 1];
    const float time = (float)(d.t0 + m) / (float)d.sr;
    float acc = 0.0f;
    for (uint32_t v = v0; v < v1; ++v) {
        const float4 nv = d.tab.voices[v];   // (hz, vel)
        const float arg = time * nv.x * 2.0f * kPi;
        acc += (d.exact_sin ? sin_glibc(arg) : sin_any(arg)) * nv.y;
    }
    return acc;
}
__global__ __launch_bounds__(kThreads) void k_debug_sine(const SineDesc* __restrict__ descs, uint32_t M) {
    const SineDesc& d = descs[blockIdx.y];
    const uint32_t m0 = blockIdx.x * kTileFrames + 2 * threadIdx.x;
    const uint32_t m1 = m0 + kTileFrames / 2;
    if (m0 < M) {
        float a = sine_frame(d, m0), b = (m0 + 1 < M) ? sine_frame(d, m0 + 1) : 0.0f;
        store_pair(d.out, m0, M, epilogue4(make_float4(a, a, b, b), d.pg));
    }
    if (m1 < M) {
        float a = sine_frame(d, m1), b = (m1 + 1 < M) ? sine_frame(d, m1 + 1) : 0.0f;
        store_pair(d.out, m1, M, epilogue4(make_float4(a, a, b, b), d.pg));
    }
}

// ------------------------------------------------------------------------------------------------
// k_synth (extensions.rs:460-529, synth.rs:21-34)
// ------------------------------------------------------------------------------------------------
// a voice's envelope at one frame (extensions.rs:505-507): the reciprocal form of the tolerance class, or -- sine_mode 1, where
// the vertex is to carry the reference's bits -- adsr.rs's own divisions (apply_ads / apply_r_rt, as the Adsr vertex has them)
TD_DEV float synth_env1(const SynthDesc& d, const AdsrConfD& c, float env_time, float rel_t) {
    if (d.exact_sin) return rel_t == 0.0f ? apply_ads(c, env_time) : apply_r_rt(c, env_time, rel_t);   // (uniform)
    return rel_t == 0.0f ? apply_ads_fast(c, env_time) : apply_r_rt_fast(c, env_time, rel_t);
}
// one voice, one frame: oscillators x velocity x envelope x volume (extensions.rs:499-524), in the reference's order
TD_DEV float synth_voice(const SynthDesc& d, const float4 n, float time, float off) {   // n = (hz, vel, env_t, rel_t)
    const float hz = n.x, vel = n.y, rel_t = n.w;
    const float env_time = n.z + off;
    float s = 0.0f;
    float sn = 0.0f;
    float env_sq = 0.0f, env_tf = 0.0f;
    if (d.square.volume > 0.0f || d.topflat.volume > 0.0f) sn = d.exact_sin ? sin_glibc(time * hz * 2.0f * kPi) : sin_any(time * hz * 2.0f * kPi);
    if (d.square.volume > 0.0f) {
        const float z = d.square.param;
        const float osc = fminf(fmaxf(sn, -z), z) * (1.0f / z);
        env_sq = synth_env1(d, d.square.adsr, env_time, rel_t);
        s += osc * vel * env_sq * d.square.volume;
    }
    if (d.topflat.volume > 0.0f) {
        const float z = d.topflat.param;
        const float osc = (fminf(sn, z) + ((1.0f - z) / 2.0f)) * (2.0f / (1.0f + z));
        env_tf = d.tf_env_src == 1u ? env_sq
               : synth_env1(d, d.topflat.adsr, env_time, rel_t);
        s += osc * vel * env_tf * d.topflat.volume;
    }
    if (d.triangle.volume > 0.0f) {
        const float th = time * hz;
        const float osc = 4.0f * fabsf(th - floorf(th + 0.5f)) - 1.0f;
        const float env = d.tr_env_src == 1u ? env_sq : d.tr_env_src == 2u ? env_tf
                        : synth_env1(d, d.triangle.adsr, env_time, rel_t);
        s += osc * vel * env * d.triangle.volume;
    }
    return s * d.osc_amp_multiplier;
}
TD_DEV float synth_frame(const SynthDesc& d, uint32_t m) {
    const uint32_t it = find_interval(d.tab, m);
    const uint32_t v0 = d.tab.ivoff[it], v1 = d.tab.ivoff[it + 1];
    const float time = (float)(d.t0 + m) / (float)d.sr;
    const float off = (float)(m % d.bl) / (float)d.sr;
    float acc = 0.0f;
    for (uint32_t v = v0; v < v1; ++v) acc += synth_voice(d, d.tab.voices[v], time, off);
    return acc;
}
// The same for the lane's four frames (two pairs) at once, on 2-vectors -- same operations in the same order as
// synth_voice, organised around what k_synth is bound by, VALU issue (tools/ubench/issue_rate.hip: fma / mul / add issue in
// ~1.05 ns per wave and SIMD, compares / selects / min / max / conversions and every packed op in ~1.75 ns):
//  * `held`, the level a released voice was released at (apply_ads(conf, rel_t), adsr.rs:89-92), depends on the voice only:
//    once per voice, not once per frame;
//  * the envelope of a held voice: a wave's frames nearly always lie on ONE piece of the curve (attack / decay / sustain
//    ramp / hold last for thousands of frames, a wave spans 640) -- the piece is found once, from the wave's first and last
//    envelope time, and all four frames take its lerp as straight packed code; a wave that straddles a breakpoint, or a
//    conf whose levels could reach the `res <= -1` escape of adsr.rs:62-69, takes the per-frame form;
//  * products and sums of the two frames of a pair issue packed.
struct SynthHeld { float sq, tf, tr; };
TD_DEV SynthHeld synth_held(const SynthDesc& d, float rel_t) {
    SynthHeld h{0.0f, 0.0f, 0.0f};
    if (rel_t != 0.0f) {
        if (d.square.volume > 0.0f) h.sq = apply_ads_fast(d.square.adsr, rel_t);
        if (d.topflat.volume > 0.0f) h.tf = d.tf_env_src == 1u ? h.sq : apply_ads_fast(d.topflat.adsr, rel_t);
        if (d.triangle.volume > 0.0f)
            h.tr = d.tr_env_src == 1u ? h.sq : d.tr_env_src == 2u ? h.tf : apply_ads_fast(d.triangle.adsr, rel_t);
    }
    return h;
}
TD_DEV int ads_piece(const AdsrConfD& c, float t) {   // 0 attack, 1 decay, 2 sustain ramp, 3 beyond (NaN: 3, like adsr.rs:46-60)
    return t <= c.attack_sec ? 0 : t <= c.attack_sec + c.decay_sec ? 1 : t <= c.attack_sec + c.decay_sec + c.sustain_sec ? 2 : 3;
}
TD_DEV void synth_env4(const AdsrConfD& c, f2 ta, f2 tb, float rel_t, float held, f2& ea, f2& eb) {   // ta, tb: envelope times
    if (rel_t != 0.0f) {   // apply_r (adsr.rs:71-73): lerp(held, release_vel, min(t / release_sec, 1))
        const float rs = __builtin_amdgcn_rcpf(c.release_sec), dv = c.release_vel - held;
        f2 ua = ta * rs, ub = tb * rs;
        ua.x = fminf(ua.x, 1.0f); ua.y = fminf(ua.y, 1.0f);
        ub.x = fminf(ub.x, 1.0f); ub.y = fminf(ub.y, 1.0f);
        ea = held + ua * dv;
        eb = held + ub * dv;
        return;
    }
    // (times grow with the frame: the wave's first and last; the builtins carry ints)
    const float lo = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(ta.x)));
    const float hi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tb.y), 63));
    const int piece = ads_piece(c, lo);
    const bool tame = fminf(fminf(c.std_vel, c.attack_vel), fminf(c.decay_vel, c.sustain_vel)) > -0.999f;
    if (tame && lo >= 0.0f && piece == ads_piece(c, hi)) {
        if (piece == 0) {
            const float r = __builtin_amdgcn_rcpf(c.attack_sec), dv = c.attack_vel - c.std_vel;
            ea = c.std_vel + (ta * r) * dv;
            eb = c.std_vel + (tb * r) * dv;
        } else if (piece == 1) {
            const float r = __builtin_amdgcn_rcpf(c.decay_sec), dv = c.decay_vel - c.attack_vel;
            ea = c.attack_vel + ((ta - c.attack_sec) * r) * dv;
            eb = c.attack_vel + ((tb - c.attack_sec) * r) * dv;
        } else if (piece == 2) {
            const float r = __builtin_amdgcn_rcpf(c.sustain_sec), dv = c.sustain_vel - c.decay_vel;
            ea = c.decay_vel + (((ta - c.attack_sec) - c.decay_sec) * r) * dv;
            eb = c.decay_vel + (((tb - c.attack_sec) - c.decay_sec) * r) * dv;
        } else {
            ea = eb = (f2)(c.sustain_vel);
        }
        return;
    }
    ea.x = apply_ads_fast(c, ta.x); ea.y = apply_ads_fast(c, ta.y);
    eb.x = apply_ads_fast(c, tb.x); eb.y = apply_ads_fast(c, tb.y);
}
struct SynthEnv2 { f2 sq, tf, tr; };
// oscillators x velocity x envelope x volume for the two frames of a pair (extensions.rs:499-524), added to `acc`.
// k = (vel * volume * osc_amp_multiplier) per oscillator, folded once per voice: per oscillator and pair the product is one
// multiply and one FMA instead of three multiplies, an add and the final `* osc_amp_multiplier` (tolerance class); the sine's
// argument `time * hz * 2.0 * PI` as (time * hz) * (2 PI): the doubling is exact, the roundings are the reference's, and
// time * hz is the triangle's argument anyway.  SPEC 1: all three oscillators on, the triangle sharing the top-flat's
// envelope (BASELINE config 3's shape) -- the uniform flag tests of the generic form cost a branch each per voice.
struct SynthAmp { float sq, tf, tr; };
template <int SPEC>
TD_DEV f2 synth_osc2(const SynthDesc& d, const float4 n, const SynthEnv2& e, const SynthAmp& k, f2 time, f2 acc) {   // n = (hz, vel, env_t, rel_t)
    const float hz = n.x;
    const bool sq_on = SPEC ? true : d.square.volume > 0.0f, tf_on = SPEC ? true : d.topflat.volume > 0.0f;
    const bool tr_on = SPEC ? true : d.triangle.volume > 0.0f;
    const f2 th = time * hz;
    f2 sn = (f2)(0.0f);
    if (sq_on || tf_on) {
        const f2 arg = th * (2.0f * kPi);
        if (d.exact_sin) { sn.x = sin_glibc(arg.x); sn.y = sin_glibc(arg.y); }   // (uniform; engine option sine_mode 1)
        else sn = sin_any2(arg);
    }
    if (sq_on) {
        const float z = d.square.param;
        f2 osc;
        osc.x = fminf(fmaxf(sn.x, -z), z);
        osc.y = fminf(fmaxf(sn.y, -z), z);
        acc = fma2(osc * (1.0f / z), e.sq * k.sq, acc);
    }
    if (tf_on) {
        const float z = d.topflat.param;
        f2 m;
        m.x = fminf(sn.x, z);
        m.y = fminf(sn.y, z);
        acc = fma2((m + ((1.0f - z) / 2.0f)) * (2.0f / (1.0f + z)), e.tf * k.tf, acc);
    }
    if (tr_on) {
        f2 fl = th + 0.5f;
        fl.x = floorf(fl.x);
        fl.y = floorf(fl.y);
        f2 dd = th - fl;
        dd.x = fabsf(dd.x);
        dd.y = fabsf(dd.y);
        acc = fma2(fma2((f2)(4.0f), dd, (f2)(-1.0f)), e.tr * k.tr, acc);
    }
    return acc;
}
// one voice, the lane's four frames: a += voice(pair a), b += voice(pair b)
template <int SPEC>
TD_DEV void synth_voice4(const SynthDesc& d, const float4 n, f2 ta, f2 tb, f2 oa, f2 ob, f2& a, f2& b) {
    const float rel_t = n.w;
    const SynthHeld h = synth_held(d, rel_t);
    const f2 eta = n.z + oa, etb = n.z + ob;
    SynthEnv2 ea{(f2)(0.0f), (f2)(0.0f), (f2)(0.0f)}, eb = ea;
    const bool sq_on = SPEC ? true : d.square.volume > 0.0f, tf_on = SPEC ? true : d.topflat.volume > 0.0f;
    const bool tr_on = SPEC ? true : d.triangle.volume > 0.0f;
    const uint32_t tf_src = SPEC ? 0u : d.tf_env_src, tr_src = SPEC ? 2u : d.tr_env_src;
    if (sq_on) synth_env4(d.square.adsr, eta, etb, rel_t, h.sq, ea.sq, eb.sq);
    if (tf_on) {
        if (tf_src == 1u) { ea.tf = ea.sq; eb.tf = eb.sq; }
        else synth_env4(d.topflat.adsr, eta, etb, rel_t, h.tf, ea.tf, eb.tf);
    }
    if (tr_on) {
        if (tr_src == 1u) { ea.tr = ea.sq; eb.tr = eb.sq; }
        else if (tr_src == 2u) { ea.tr = ea.tf; eb.tr = eb.tf; }
        else synth_env4(d.triangle.adsr, eta, etb, rel_t, h.tr, ea.tr, eb.tr);
    }
    const float kv = n.y * d.osc_amp_multiplier;
    const SynthAmp k{kv * d.square.volume, kv * d.topflat.volume, kv * d.triangle.volume};
    a = synth_osc2<SPEC>(d, n, ea, k, ta, a);
    b = synth_osc2<SPEC>(d, n, eb, k, tb, b);
}
// The lane's frame pairs ma, ma + 1 and mb, mb + 1.  A wave's frames nearly always lie in ONE interval (intervals start
// at block starts and event frames): the voice list is then the same for every lane, the voice records come in through
// scalar loads (constant address space) and all four frames share one pass over the voices.  A wave that straddles an
// interval start takes the per-lane form.
TD_DEV f2 synth_time2(const SynthDesc& d, uint32_t m) {
    return f2{(float)(d.t0 + m) / (float)d.sr, (float)(d.t0 + m + 1u) / (float)d.sr};
}
TD_DEV f2 synth_off2(const SynthDesc& d, uint32_t m) {
    return f2{(float)(m % d.bl) / (float)d.sr, (float)((m + 1u) % d.bl) / (float)d.sr};
}
TD_DEV void synth_quad(const SynthDesc& d, uint32_t ma, uint32_t mb, uint32_t M, float2& pa, float2& pb) {
    const bool two_a = ma + 1u < M, two_b = mb + 1u < M;
    const uint32_t i0 = find_interval(d.tab, ma), i1 = two_a ? find_interval(d.tab, ma + 1u) : i0;
    const uint32_t i2 = find_interval(d.tab, mb), i3 = two_b ? find_interval(d.tab, mb + 1u) : i2;
    const uint32_t it0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)i0);
    const bool one_interval = __all((i0 == it0 && i1 == it0 && i2 == it0 && i3 == it0) ? 1 : 0) != 0;
    if (d.exact_sin && one_interval) {
        // sine_mode 1: every frame in the reference's own order of operations (synth_voice: exact divisions, sin_glibc) -- what the
        // wave shares is the walk: one interval look-up, the voice records through scalar loads, once for its four frames each
        const uint32_t TD_CONST* off_c = (const uint32_t TD_CONST*)(const TD_CONST char*)d.tab.ivoff;
        const uint32_t v0 = off_c[it0], v1 = off_c[it0 + 1u];
        typedef float f4c __attribute__((ext_vector_type(4)));
        const f4c TD_CONST* vc = (const f4c TD_CONST*)(const TD_CONST char*)d.tab.voices;
        const f2 ta = synth_time2(d, ma), tb = synth_time2(d, mb), oa = synth_off2(d, ma), ob = synth_off2(d, mb);
        float a0 = 0.0f, a1 = 0.0f, b0 = 0.0f, b1 = 0.0f;
        for (uint32_t v = v0; v < v1; ++v) {
            const f4c q = vc[v];
            const float4 n = make_float4(q.x, q.y, q.z, q.w);
            a0 += synth_voice(d, n, ta.x, oa.x);
            a1 += synth_voice(d, n, ta.y, oa.y);
            b0 += synth_voice(d, n, tb.x, ob.x);
            b1 += synth_voice(d, n, tb.y, ob.y);
        }
        pa = make_float2(a0, two_a ? a1 : 0.0f);
        pb = make_float2(b0, two_b ? b1 : 0.0f);
        return;
    }
    if (!d.exact_sin && one_interval) {
        const uint32_t TD_CONST* off_c = (const uint32_t TD_CONST*)(const TD_CONST char*)d.tab.ivoff;
        const uint32_t v0 = off_c[it0], v1 = off_c[it0 + 1u];
        typedef float f4c __attribute__((ext_vector_type(4)));
        const f4c TD_CONST* vc = (const f4c TD_CONST*)(const TD_CONST char*)d.tab.voices;
        f2 ta = synth_time2(d, ma), tb = synth_time2(d, mb), oa = synth_off2(d, ma), ob = synth_off2(d, mb);
        // a pair's second frame beyond the chunk is never stored: it shadows the first, so that times keep growing with
        // the lane (its own in-block offset would have wrapped to 0, and synth_env4 reads the wave's last time from lane 63)
        if (!two_a) { ta.y = ta.x; oa.y = oa.x; }
        if (!two_b) { tb.y = tb.x; ob.y = ob.x; }
        f2 a = (f2)(0.0f), b = (f2)(0.0f);
        if (d.square.volume > 0.0f && d.topflat.volume > 0.0f && d.triangle.volume > 0.0f && d.tf_env_src == 0u && d.tr_env_src == 2u) {
            f4c q = vc[v0];   // (v0 == v1: the terminating record of the table is read and not used)
            for (uint32_t v = v0; v < v1; ++v) {   // (one uniform test per wave instead of half a dozen per voice)
                const f4c qn = vc[v + 1u];   // the next record's scalar load flies under this voice's arithmetic
                synth_voice4<1>(d, make_float4(q.x, q.y, q.z, q.w), ta, tb, oa, ob, a, b);
                q = qn;
            }
        } else {
            for (uint32_t v = v0; v < v1; ++v) {
                const f4c q = vc[v];
                synth_voice4<0>(d, make_float4(q.x, q.y, q.z, q.w), ta, tb, oa, ob, a, b);
            }
        }
        pa = make_float2(a.x, two_a ? a.y : 0.0f);
        pb = make_float2(b.x, two_b ? b.y : 0.0f);
        return;
    }
    pa = make_float2(synth_frame(d, ma), two_a ? synth_frame(d, ma + 1u) : 0.0f);
    pb = make_float2(synth_frame(d, mb), two_b ? synth_frame(d, mb + 1u) : 0.0f);
}
// ---- the affine form (SynthDesc::affine): every oscillator's envelope x velocity x volume x amplitude scale is
// A + B ((t - s1) - s2) inside an interval, (s1, s2, A, B) from the voice record (scalar loads) -- no piece selection, no
// held level, no per-voice products on the device.  One voice, one frame pair:
//   th = time hz;  sn = sin(th 2 pi);  t = env_t + off;
//   square    clamp(sn, -z, z)            x  fma(u, B, A)      (1 / z folded into A, B)
//   top-flat  (min(sn, z) + (1 - z) / 2)  x  fma(u, B, A)      (2 / (1 + z) folded)
//   triangle  (4 |th - floor(th + 0.5)| - 1) x fma(u, B, A)
// SHARE_TR: the triangle's conf is the top-flat's (BASELINE config 3's shape): same (s1, s2), one u for both.
typedef float f4c __attribute__((ext_vector_type(4)));
// An oscillator whose record holds A = B = 0 -- its envelope piece is identically 0: a hit-shaped envelope behind its decay, BASELINE
// config 3's square oscillator most of the time -- adds `c * 0` to the sum, i.e. nothing (c is finite); the host says so in the
// voice's first record (bits 0..2 of its third word), the skipped block is a uniform branch (round 6).
// (clang 19 note: __builtin_bit_cast of a vector ELEMENT (`q.z`) reads element 0 -- copy the element to a scalar first.)
// (one voice, the lane's TWO frame pairs: every test on the record -- scalar unit, one per CU -- is made once for both)
template <bool SQ, bool TF, bool TR, bool SHARE_TR, bool SMALL>
TD_DEV void synth_quad_affine_voice(const f4c q0, const f4c q1, const f4c q2, const f4c q3, float zsq, float ztf, float tf_bias,
                                    f2 time_a, f2 off_a, f2 time_b, f2 off_b, f2& acc_a, f2& acc_b) {
    const f2 tha = time_a * q0.x, thb = time_b * q0.x;
    const f2 t_a = q0.y + off_a, t_b = q0.y + off_b;
    const float lz = q0.z;   // (the host's verdict per oscillator, bits 0..2: synth_refine_affine)
    const uint32_t live = __float_as_uint(lz);
    const bool l1 = SQ && (live & 1u), l2 = TF && (live & 2u), l3 = TR && (live & 4u);
    f2 sna = (f2)(0.0f), snb = (f2)(0.0f);
    if (l1 || l2) {
        asm volatile("");
        sna = SMALL ? sin_small2(tha * (2.0f * kPi)) : sin_any2(tha * (2.0f * kPi));
        snb = SMALL ? sin_small2(thb * (2.0f * kPi)) : sin_any2(thb * (2.0f * kPi));
    }
    if (l1) {
        asm volatile("");
        const f2 eka = fma2((t_a - q1.x) - q1.y, (f2)(q1.w), (f2)(q1.z)), ekb = fma2((t_b - q1.x) - q1.y, (f2)(q1.w), (f2)(q1.z));
        f2 ca, cb;   // clamp(sn, -z, z) as ONE v_med3_f32 (sn is finite: a polynomial of a finite argument)
        ca.x = __builtin_amdgcn_fmed3f(sna.x, -zsq, zsq);
        ca.y = __builtin_amdgcn_fmed3f(sna.y, -zsq, zsq);
        cb.x = __builtin_amdgcn_fmed3f(snb.x, -zsq, zsq);
        cb.y = __builtin_amdgcn_fmed3f(snb.y, -zsq, zsq);
        acc_a = fma2(ca, eka, acc_a);
        acc_b = fma2(cb, ekb, acc_b);
    }
    f2 utfa = (f2)(0.0f), utfb = (f2)(0.0f);
    if (l2 || (SHARE_TR && l3)) { utfa = (t_a - q2.x) - q2.y; utfb = (t_b - q2.x) - q2.y; }
    if (l2) {
        asm volatile("");
        const f2 eka = fma2(utfa, (f2)(q2.w), (f2)(q2.z)), ekb = fma2(utfb, (f2)(q2.w), (f2)(q2.z));
        f2 ma, mb;   // min(sn, z) as the median of (sn, z, -inf): no canonicalising v_max in front of it
        ma.x = __builtin_amdgcn_fmed3f(sna.x, ztf, -__builtin_inff());
        ma.y = __builtin_amdgcn_fmed3f(sna.y, ztf, -__builtin_inff());
        mb.x = __builtin_amdgcn_fmed3f(snb.x, ztf, -__builtin_inff());
        mb.y = __builtin_amdgcn_fmed3f(snb.y, ztf, -__builtin_inff());
        acc_a = fma2(ma + tf_bias, eka, acc_a);
        acc_b = fma2(mb + tf_bias, ekb, acc_b);
    }
    if (l3) {
        asm volatile("");
        const f2 ua = SHARE_TR ? utfa : (t_a - q3.x) - q3.y, ub = SHARE_TR ? utfb : (t_b - q3.x) - q3.y;
        const f2 eka = fma2(ua, (f2)(q3.w), (f2)(q3.z)), ekb = fma2(ub, (f2)(q3.w), (f2)(q3.z));
        f2 fla = tha + 0.5f, flb = thb + 0.5f;
        fla.x = floorf(fla.x); fla.y = floorf(fla.y);
        flb.x = floorf(flb.x); flb.y = floorf(flb.y);
        const f2 da = tha - fla, db = thb - flb;
        f2 wa, wb;   // 4 |d| - 1: the absolute value rides as a source modifier of a plain v_fma_f32 (written out: left to itself the
        // compiler clears the sign bits with two v_and and packs the two FMAs -- 5.6 ns of issue instead of 2.1)
        asm("v_fma_f32 %0, |%1|, 4.0, -1.0" : "=v"(wa.x) : "v"(da.x));
        asm("v_fma_f32 %0, |%1|, 4.0, -1.0" : "=v"(wa.y) : "v"(da.y));
        asm("v_fma_f32 %0, |%1|, 4.0, -1.0" : "=v"(wb.x) : "v"(db.x));
        asm("v_fma_f32 %0, |%1|, 4.0, -1.0" : "=v"(wb.y) : "v"(db.y));
        acc_a = fma2(wa, eka, acc_a);
        acc_b = fma2(wb, ekb, acc_b);
    }
}
// all voices of interval `it` for the lane's two frame pairs (uniform: the records come in through scalar loads)
typedef float f16c __attribute__((ext_vector_type(16)));
template <bool SQ, bool TF, bool TR, bool SHARE_TR, bool SMALL>
TD_DEV void synth_interval_affine(const SynthDesc& d, uint32_t it_, f2 ta, f2 tb, f2 oa, f2 ob, f2& a, f2& b) {
    const uint32_t it = (uint32_t)__builtin_amdgcn_readfirstlane((int)it_);   // (wave-uniform by construction: say so)
    const uint32_t TD_CONST* off_c = (const uint32_t TD_CONST*)(const TD_CONST char*)d.tab.ivoff;
    const uint32_t v0 = off_c[it], v1 = off_c[it + 1u];
    const float zsq = d.square.param, ztf = d.topflat.param, tf_bias = (1.0f - ztf) / 2.0f;
    // a voice's four records are 64 consecutive bytes: ONE scalar load (and one address) per voice -- as four 16-byte loads
    // they were four 64-bit address computations on the scalar unit, which the CU's four SIMDs share
    const f16c TD_CONST* vc = (const f16c TD_CONST*)(const TD_CONST char*)d.tab.voices + v0;
    f16c q = *vc;   // (v0 == v1: the table ends with spare records)
    // (the first record waited for HERE: with it still in flight at the loop's head the compiler's wait lands behind the loop's own
    // load -- every trip then waits for the record it has just asked for instead of finding it a trip later)
    asm volatile("" :: "s"(q));
    for (uint32_t v = v0; v < v1; ++v) {
        // (the next voice's load: the compiler sinks it to the top of the next trip; issuing it by hand a trip ahead was measured
        // in round 6 and changes nothing: six waves per SIMD cover the round trip)
        ++vc;
        const f16c n = *vc;
        synth_quad_affine_voice<SQ, TF, TR, SHARE_TR, SMALL>(__builtin_shufflevector(q, q, 0, 1, 2, 3), __builtin_shufflevector(q, q, 4, 5, 6, 7),
                                                             __builtin_shufflevector(q, q, 8, 9, 10, 11), __builtin_shufflevector(q, q, 12, 13, 14, 15),
                                                             zsq, ztf, tf_bias, ta, oa, tb, ob, a, b);
        q = n;
    }
}
template <bool SMALL>
TD_DEV void synth_interval_affine_kinds(const SynthDesc& d, uint32_t it, f2 ta, f2 tb, f2 oa, f2 ob, f2& a, f2& b) {
    const bool sq = d.square.volume > 0.0f, tf = d.topflat.volume > 0.0f, tr = d.triangle.volume > 0.0f;   // (uniform)
    if (sq && tf && tr) {
        if (d.tf_env_src == 0u && d.tr_env_src == 2u) synth_interval_affine<true, true, true, true, SMALL>(d, it, ta, tb, oa, ob, a, b);
        else synth_interval_affine<true, true, true, false, SMALL>(d, it, ta, tb, oa, ob, a, b);
    } else if (sq && tf) synth_interval_affine<true, true, false, false, SMALL>(d, it, ta, tb, oa, ob, a, b);
    else if (sq && tr) synth_interval_affine<true, false, true, false, SMALL>(d, it, ta, tb, oa, ob, a, b);
    else if (tf && tr) synth_interval_affine<false, true, true, false, SMALL>(d, it, ta, tb, oa, ob, a, b);
    else if (sq) synth_interval_affine<true, false, false, false, SMALL>(d, it, ta, tb, oa, ob, a, b);
    else if (tf) synth_interval_affine<false, true, false, false, SMALL>(d, it, ta, tb, oa, ob, a, b);
    else if (tr) synth_interval_affine<false, false, true, false, false>(d, it, ta, tb, oa, ob, a, b);   // (no sine in a triangle)
}
TD_DEV void synth_interval_affine_any(const SynthDesc& d, uint32_t it, f2 ta, f2 tb, f2 oa, f2 ob, f2& a, f2& b) {
    if (d.small_args) synth_interval_affine_kinds<true>(d, it, ta, tb, oa, ob, a, b);   // (uniform)
    else synth_interval_affine_kinds<false>(d, it, ta, tb, oa, ob, a, b);
}
TD_DEV uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, o, 64));
    return v;
}
TD_DEV void synth_quad_affine(const SynthDesc& d, uint32_t ma, uint32_t mb, uint32_t M, float2& pa, float2& pb) {
    const bool two_a = ma + 1u < M, two_b = mb + 1u < M;
    const uint32_t i0 = find_interval(d.tab, ma), i1 = two_a ? find_interval(d.tab, ma + 1u) : i0;
    const uint32_t i2 = find_interval(d.tab, mb), i3 = two_b ? find_interval(d.tab, mb + 1u) : i2;
    f2 ta = synth_time2(d, ma), tb = synth_time2(d, mb), oa = synth_off2(d, ma), ob = synth_off2(d, mb);
    if (!two_a) { ta.y = ta.x; oa.y = oa.x; }
    if (!two_b) { tb.y = tb.x; ob.y = ob.x; }
    const uint32_t it0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)i0);
    f2 a = (f2)(0.0f), b = (f2)(0.0f);
    if (__all((i0 == it0 && i1 == it0 && i2 == it0 && i3 == it0) ? 1 : 0)) {
        synth_interval_affine_any(d, it0, ta, tb, oa, ob, a, b);
    } else {
        // The wave's frames lie in several intervals (an event or an envelope breakpoint inside its two 128-frame runs):
        // one pass of the same voice loop per interval that holds any of them, every frame keeping the pass of its own.
        uint32_t it = wave_min_u32(min(min(i0, i1), min(i2, i3)));
        for (;;) {
            f2 ca = (f2)(0.0f), cb = (f2)(0.0f);
            synth_interval_affine_any(d, it, ta, tb, oa, ob, ca, cb);
            if (i0 == it) a.x = ca.x;
            if (i1 == it) a.y = ca.y;
            if (i2 == it) b.x = cb.x;
            if (i3 == it) b.y = cb.y;
            // the next interval any frame of the wave lies in
            const uint32_t big = 0xFFFFFFFFu;
            const uint32_t nx = wave_min_u32(min(min(i0 > it ? i0 : big, i1 > it ? i1 : big), min(i2 > it ? i2 : big, i3 > it ? i3 : big)));
            if (nx == big) break;
            it = nx;
        }
    }
    pa = make_float2(a.x, two_a ? a.y : 0.0f);
    pb = make_float2(b.x, two_b ? b.y : 0.0f);
}
// (six workgroups per CU: 80 registers and a few spilled words instead of 100 -- 0.134 -> 0.126 ms on config 3; seven and eight
// spill into the voice loop and lose)
__global__ __launch_bounds__(kThreads, 6) void k_synth(const SynthDesc* __restrict__ descs, uint32_t M) {
    const SynthDesc& d = descs[blockIdx.y];
    // (the costliest tiles go first: IntervalTab::tile_order)
    const uint32_t tile = d.tab.tile_order ? ((const uint32_t TD_CONST*)(const TD_CONST char*)d.tab.tile_order)[blockIdx.x] : blockIdx.x;
    const uint32_t m0 = tile * kTileFrames + 2 * threadIdx.x;
    const uint32_t m1 = m0 + kTileFrames / 2;
    // (frames at or beyond M are computed on clamped indices and not stored: the uniform-interval test needs whole waves)
    const uint32_t mc0 = min(m0, M - 1u), mc1 = min(m1, M - 1u);
    float2 p0, p1;
    synth_quad(d, mc0, mc1, M, p0, p1);
    if (m0 < M) store_pair(d.out, m0, M, epilogue4(make_float4(p0.x, p0.x, p0.y, p0.y), d.pg));
    if (m1 < M) store_pair(d.out, m1, M, epilogue4(make_float4(p1.x, p1.x, p1.y, p1.y), d.pg));
}
// the affine form (every descriptor of the launch has SynthDesc::affine set: the engine groups them)
TD_DEV void synth_affine_block(const SynthDesc& d, uint32_t bx, uint32_t M) {
    const uint32_t tile = d.tab.tile_order ? ((const uint32_t TD_CONST*)(const TD_CONST char*)d.tab.tile_order)[bx] : bx;
    const uint32_t m0 = tile * kTileFrames + 2 * threadIdx.x;
    const uint32_t m1 = m0 + kTileFrames / 2;
    const uint32_t mc0 = min(m0, M - 1u), mc1 = min(m1, M - 1u);
    float2 p0, p1;
    synth_quad_affine(d, mc0, mc1, M, p0, p1);
    if (m0 < M) store_pair(d.out, m0, M, epilogue4(make_float4(p0.x, p0.x, p0.y, p0.y), d.pg));
    if (m1 < M) store_pair(d.out, m1, M, epilogue4(make_float4(p1.x, p1.x, p1.y, p1.y), d.pg));
}
__global__ __launch_bounds__(kThreads, 6) void k_synth_affine(const SynthDesc* __restrict__ descs, uint32_t M) {
    synth_affine_block(descs[blockIdx.y], blockIdx.x, M);
}

// ------------------------------------------------------------------------------------------------
// k_sine_probe: what the fast sine kinds' output differs by from the reference's own arithmetic, MEASURED on a sample of the
// chunk's frames (kernels.h ProbeDesc; engine option "sine_mode" 2)
// ------------------------------------------------------------------------------------------------
// A workgroup owns 16 sample frames, a row of 16 lanes each: lane `sub` evaluates voices sub, sub + 16, ... of the frame's
// interval the reference's way (synth_voice with exact_sin / sin_glibc) and the row adds them up in the voice order of the
// reference's loop (`acc += voice`: the additions are what is serial, the voices are not) -- lane 0 takes lane k's value as the
// DPP operand of a move (row_shl:k), no trip through LDS.  The kernel is latency: ~700 workgroups for 60 s, one pass.
TD_DEV float probe_dev2(float got, float want) {
    if (got == want || (got != got && want != want)) return 0.0f;   // (NaN where the reference is NaN: nothing to answer for)
    const float d = got - want, e = d * d;
    return e == e ? e : __builtin_inff();                          // a NaN / infinity on one side only: over any bound
}
template <int K>
TD_DEV float row_shl(float v) {   // lane i of a row of 16 receives lane i + K's value (lanes past the row's end: 0)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x100 + K, 0xF, 0xF, true));
}
// sample `idx` of the chunk, lane `sub` of its row of sixteen: the sample's energy x the frames it stands for, in the row's lane 0
// (0 for a sample whose frame lies beyond the chunk)
TD_DEV float probe_energy(const ProbeDesc& p, uint32_t idx, uint32_t sub, uint32_t M) {
    const uint32_t lg = p.stride_log2, stride = 1u << lg;
    const uint32_t m = probe_frame(idx, lg);
    const bool valid = m < M;                                    // (the last sample's frame may lie beyond the chunk: nothing measured)
    const uint32_t mc = valid ? m : 0u;
    const bool synth = p.kind == 1u;
    const IntervalTab& tab = synth ? p.syn.tab : p.sine.tab;
    const uint64_t t0 = synth ? p.syn.t0 : p.sine.t0;
    const uint32_t sr = synth ? p.syn.sr : p.sine.sr;
    // (the frame's interval is the host's look-up: one load instead of find_interval's walk; what the fast launch left, early)
    u2v vr; vr.x = 0u; vr.y = 0u;
    if (valid) vr = *reinterpret_cast<const u2v TD_GLOBAL*>((const TD_GLOBAL char*)(p.ranges + 2u * idx));
    const uint32_t v0 = vr.x, v1 = vr.y;
    const float2 got = (valid && sub == 0u) ? gload2((synth ? p.syn.out : p.sine.out) + m) : make_float2(0.0f, 0.0f);
    const float time = (float)(t0 + mc) / (float)sr;
    const float off = synth ? (float)(mc % p.syn.bl) / (float)sr : 0.0f;
    float acc = 0.0f;
    for (uint32_t r = v0;; r += 16u) {
        if (!__any((r < v1) ? 1 : 0)) break;
        const uint32_t v = r + sub;
        float val = 0.0f;
        if (v < v1) {
            const float4 nv = tab.voices[v];
            if (synth) val = synth_voice(p.syn, nv, time, off);
            else val = sin_glibc(time * nv.x * 2.0f * kPi) * nv.y;   // sine_frame's term, exact_sin
        }
        const uint32_t left = r < v1 ? v1 - r : 0u;               // voices of this round (lane 0 of the row adds them in order)
#define TD_PROBE_ADD(K) { const float vk = (K) ? row_shl<(K) ? (K) : 1>(val) : val; if ((uint32_t)(K) < left) acc += vk; }
        TD_PROBE_ADD(0) TD_PROBE_ADD(1) TD_PROBE_ADD(2) TD_PROBE_ADD(3) TD_PROBE_ADD(4) TD_PROBE_ADD(5) TD_PROBE_ADD(6) TD_PROBE_ADD(7)
        TD_PROBE_ADD(8) TD_PROBE_ADD(9) TD_PROBE_ADD(10) TD_PROBE_ADD(11) TD_PROBE_ADD(12) TD_PROBE_ADD(13) TD_PROBE_ADD(14) TD_PROBE_ADD(15)
#undef TD_PROBE_ADD
    }
    float e = 0.0f;
    if (sub == 0u && valid) {
        const PanGain& pg = synth ? p.syn.pg : p.sine.pg;
        const float2 want = epilogue(make_float2(acc, acc), pg);
        e = fmaxf(probe_dev2(got.x, want.x), probe_dev2(got.y, want.y)) * (float)min(stride, M - (idx << lg));
    }
    return e;
}
__global__ __launch_bounds__(kThreads) void k_sine_probe(const ProbeDesc* __restrict__ descs, uint32_t M) {
    const ProbeDesc& p = descs[blockIdx.y];
    const uint32_t idx = blockIdx.x * 16u + (threadIdx.x >> 4), sub = threadIdx.x & 15u;   // the sample stands for frames [idx << lg, (idx + 1) << lg)
    const float e = probe_energy(p, idx, sub, M);
    if (sub == 0u && (idx << p.stride_log2) < M) p.noise[idx] = e;
}
void launch_sine_probe(const ProbeDesc* d, int n_desc, uint32_t frames, uint32_t n_groups, hipStream_t s) {   // n_groups: workgroups, 16 samples each
    if (n_desc > 0 && n_groups > 0) hipLaunchKernelGGL(k_sine_probe, dim3(n_groups, (uint32_t)n_desc), dim3(kThreads), 0, s, d, frames);
}

// ------------------------------------------------------------------------------------------------
// k_sampsyn (extensions.rs:532-578; oscillator and table format are this engine's own, see kernels.h)
// ------------------------------------------------------------------------------------------------
TD_DEV float wavetable_act(const WaveTableD& w, float hz, float t) {
    float ph = t * hz;
    ph = ph - floorf(ph);
    const float pos = ph * (float)w.frame_len;
    uint32_t i0 = pos >= 0.0f ? (uint32_t)pos : 0u;   // NaN / negative -> 0, like Rust's `as usize`
    if (i0 >= w.frame_len) i0 = w.frame_len - 1u;
    const float a = pos - (float)i0;
    const uint32_t i1 = i0 + 1u == w.frame_len ? 0u : i0 + 1u;
    float fp = fminf(t / w.table_seconds, 1.0f) * (float)(w.n_frames - 1u);
    if (!(fp >= 0.0f)) fp = 0.0f;
    uint32_t f0 = (uint32_t)fp;
    if (f0 >= w.n_frames) f0 = w.n_frames - 1u;
    const float b = fp - (float)f0;
    const uint32_t f1 = f0 + 1u < w.n_frames ? f0 + 1u : w.n_frames - 1u;
    // ONE 16-byte gather per voice-frame: the table is held as quads {w[f][i], w[f][i+1 wrapped], w[f+1 clamped][i],
    // w[f+1 clamped][i+1 wrapped]} (four scattered dwords cost four L1 line look-ups per lane, which bounded this kernel)
    (void)i1; (void)f1;
    const float4 q = w.quads[(size_t)f0 * w.frame_len + i0];
    const float s0 = lerpf(q.x, q.y, a);
    const float s1 = lerpf(q.z, q.w, a);
    return lerpf(s0, s1, b);
}
TD_DEV float sampsyn_frame(const SampsynDesc& d, uint32_t m) {
    const uint32_t it = find_interval(d.tab, m);
    const uint32_t v0 = d.tab.ivoff[it], v1 = d.tab.ivoff[it + 1];
    const float off = (float)(m % d.bl) / (float)d.sr;
    float acc = 0.0f;
    for (uint32_t v = v0; v < v1; ++v) {
        const float4 n = d.tab.voices[v];   // (hz, vel, env_t, rel_t)
        const float env_time = n.z + off;
        const float env = n.w == 0.0f ? apply_ads(d.adsr, env_time) : apply_r_rt(d.adsr, env_time, n.w);
        float s = 0.0f;
        const float vel = n.y * env * d.amp_multiplier;
        s += wavetable_act(d.wt, n.x, env_time + n.w) * vel;
        acc += s;
    }
    return acc;
}
// one voice's contribution to frame m (extensions.rs:556-570): the same f32 operations as sampsyn_frame's loop body
TD_DEV float sampsyn_voice(const SampsynDesc& d, const float4 n, float off) {   // n = (hz, vel, env_t at the block start, rel_t)
    const float env_time = n.z + off;
    const float env = n.w == 0.0f ? apply_ads(d.adsr, env_time) : apply_r_rt(d.adsr, env_time, n.w);
    float s = 0.0f;
    const float vel = n.y * env * d.amp_multiplier;
    s += wavetable_act(d.wt, n.x, env_time + n.w) * vel;
    return s;
}
TD_DEV void sampsyn_block(const SampsynDesc& d, uint32_t bx, uint32_t M) {
    // (the costliest tiles first, as in k_synth: IntervalTab::tile_order)
    const uint32_t tile = d.tab.tile_order ? ((const uint32_t TD_CONST*)(const TD_CONST char*)d.tab.tile_order)[bx] : bx;
    const uint32_t m0 = tile * kTileFrames + 2 * threadIdx.x;
    const uint32_t m1 = m0 + kTileFrames / 2;
    // A wave's frames nearly always lie in ONE interval: the voice records then come in through scalar loads, once per wave,
    // and the per-frame interval walk (two dependent loads and a loop per frame) is paid once.  Frames at or beyond M are
    // computed on clamped indices and not stored.  Same f32 operations per frame and voice, in the same order: bit-exact.
    const uint32_t mc[4] = {min(m0, M - 1u), min(m0 + 1u, M - 1u), min(m1, M - 1u), min(m1 + 1u, M - 1u)};
    uint32_t it[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) it[e] = find_interval(d.tab, mc[e]);
    const uint32_t it0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)it[0]);
    float v[4];
    if (__all((it[0] == it0 && it[1] == it0 && it[2] == it0 && it[3] == it0) ? 1 : 0)) {
        const uint32_t TD_CONST* off_c = (const uint32_t TD_CONST*)(const TD_CONST char*)d.tab.ivoff;
        const uint32_t v0 = off_c[it0], v1 = off_c[it0 + 1u];
        typedef float f4c __attribute__((ext_vector_type(4)));
        const f4c TD_CONST* vc = (const f4c TD_CONST*)(const TD_CONST char*)d.tab.voices;
        float off[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { off[e] = (float)(mc[e] % d.bl) / (float)d.sr; v[e] = 0.0f; }
#pragma unroll 2   // (two voices' table gathers in flight: 65 -> 61 us on config 4)
        for (uint32_t vi = v0; vi < v1; ++vi) {
            const f4c q = vc[vi];
            const float4 n = make_float4(q.x, q.y, q.z, q.w);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += sampsyn_voice(d, n, off[e]);
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = sampsyn_frame(d, mc[e]);
    }
    if (m0 < M) store_pair(d.out, m0, M, epilogue4(make_float4(v[0], v[0], m0 + 1 < M ? v[1] : 0.0f, m0 + 1 < M ? v[1] : 0.0f), d.pg));
    if (m1 < M) store_pair(d.out, m1, M, epilogue4(make_float4(v[2], v[2], m1 + 1 < M ? v[3] : 0.0f, m1 + 1 < M ? v[3] : 0.0f), d.pg));
}
__global__ __launch_bounds__(kThreads) void k_sampsyn(const SampsynDesc* __restrict__ descs, uint32_t M) {
    sampsyn_block(descs[blockIdx.y], blockIdx.x, M);
}

// ------------------------------------------------------------------------------------------------
// k_adsr: envelope-follower vertex (extensions.rs:593-651)
// ------------------------------------------------------------------------------------------------
// the vertex' gain for frame m: lerp(1.0, adsr_vel, wet) (extensions.rs:636-645); 1.0 for a frame the reference leaves
// untouched (the `continue` at extensions.rs:632-635: x * 1.0f == x bit for bit)
TD_DEV float adsr_vel(const AdsrVDesc& d, uint32_t m) {
    const uint32_t it = find_interval(d.tab, m);
    const float4 p = d.tab.voices[2 * it], g = d.tab.voices[2 * it + 1];   // (t_off, vel, release_val, skip)
    if (p.w != 0.0f) return 1.0f;
    const float offset = (float)(m % d.bl) / (float)d.sr;
    float pvel, gvel;
    if (d.use_off) {
        pvel = p.z == 0.0f ? apply_ads(d.conf, p.x + offset) * p.y : apply_r(d.conf, p.x + offset, p.z) * p.y;
        gvel = g.z == 0.0f ? apply_ads(d.conf, g.x + offset) * g.y : apply_r(d.conf, g.x + offset, g.z) * g.y;
    } else {
        pvel = apply_adsr(d.conf, p.x + offset) * p.y;
        gvel = apply_adsr(d.conf, g.x + offset) * g.y;
    }
    const float maxmul = d.use_max ? 1.0f : 0.0f;
    const float minmul = 1.0f - maxmul;
    const float av = fmaxf(pvel, gvel) * maxmul + fminf(pvel, gvel) * minmul;
    return lerpf(1.0f, av, d.wet);
}
TD_DEV float2 adsr_frame(const AdsrVDesc& d, uint32_t m, float2 x) {
    const float vel = adsr_vel(d, m);
    return make_float2(x.x * vel, x.y * vel);
}
// ------------------------------------------------------------------------------------------------
// k_adsr_env: the per-frame gain of an Adsr vertex that is read through by its consumer (InTerm kind 5)
// ------------------------------------------------------------------------------------------------
// env[m] = adsr_vel(d, m) for every frame of the chunk -- a function of the frame and the vertex' event tables only, so
// vertices with identical tables and parameters (the 84 envelope stages of a deep chain) share ONE buffer and ONE launch,
// and their consumers multiply by a 4-byte load instead of evaluating ~150 instructions per frame.
// A lane owns kEnvRun consecutive frames.  What adsr_vel pays per frame -- the interval walk, two voice records, IEEE
// divisions -- is paid once per run where the run lies in one interval (intervals start at block starts and event
// frames, so nearly always):
//  * divisions by the conf's segment lengths and by (float)sr as (float)((double)n * rcp), rcp = 1.0 / (double)x from the
//    host.  That IS the IEEE f32 quotient: the product is within 1.5 * 2^-53 of n / x, and the quotient of two f32 is
//    never closer than 2^-49 (relative) to a rounding boundary of a normal f32 result; quotients too small for that
//    argument (below 1e-30: envelope times never get there) take the division.  n / 0 and 0 / 0 give +-inf / NaN as the
//    division does.
//  * the piece of the curve (attack / decay / sustain ramp / release) is found for the run's first and last envelope time
//    and, if the same, applied as straight code.
// Confs whose levels could reach the `res <= -1.0` escape (adsr.rs:62-69,75-86), negative or NaN times and runs that
// straddle an interval start take adsr_vel frame by frame.  Same f32 operations in the same order either way: bit-exact.
constexpr int kEnvRun = 8;
TD_DEV float fdiv_rcp(float n, float x, double rcp) {
    const float q = (float)((double)n * rcp);
    if (fabsf(q) < 1.0e-30f && q != 0.0f) return n / x;
    return q;
}
// The conf's scalars as opaque registers: read through the descriptor, the per-lane piece selects below are folded by
// the compiler into INDEXED loads of a copy in scratch memory.
TD_DEV float opaque_s(float x) {
    asm volatile("" : "+v"(x));
    return x;
}
TD_DEV double opaque_s(double x) {
    asm volatile("" : "+v"(x));
    return x;
}
struct AdsrRunConsts {
    float A, D, S, R, AD, ADS, std_v, att_v, dec_v, sus_v, rel_v, dv0, dv1, dv2, dv3, srf;
    double rA, rD, rS, rR, rsr;
};
TD_DEV AdsrRunConsts adsr_run_consts(const AdsrVDesc& d) {
    const AdsrConfD& c = d.conf;
    AdsrRunConsts u;
    u.A = opaque_s(c.attack_sec); u.D = opaque_s(c.decay_sec); u.S = opaque_s(c.sustain_sec); u.R = opaque_s(c.release_sec);
    u.AD = opaque_s(c.attack_sec + c.decay_sec);
    u.ADS = opaque_s(c.attack_sec + c.decay_sec + c.sustain_sec);
    u.std_v = opaque_s(c.std_vel); u.att_v = opaque_s(c.attack_vel); u.dec_v = opaque_s(c.decay_vel);
    u.sus_v = opaque_s(c.sustain_vel); u.rel_v = opaque_s(c.release_vel);
    u.dv0 = opaque_s(c.attack_vel - c.std_vel); u.dv1 = opaque_s(c.decay_vel - c.attack_vel);
    u.dv2 = opaque_s(c.sustain_vel - c.decay_vel); u.dv3 = opaque_s(c.release_vel - c.sustain_vel);
    u.srf = opaque_s((float)d.sr);
    u.rA = opaque_s(d.rcp[0]); u.rD = opaque_s(d.rcp[1]); u.rS = opaque_s(d.rcp[2]); u.rR = opaque_s(d.rcp[3]);
    u.rsr = opaque_s(d.rcp[4]);
    return u;
}
struct AdsrPiece { float a, dv, s1, s2, s3, x; double rcp; int k; };   // value = a + q dv, q = (((t - s1) - s2) - s3) / x [min 1: k == 3]
TD_DEV int adsr_piece_index(const AdsrRunConsts& u, float t) {   // the `t <= ...` chain of adsr.rs:46-60 (NaN: 3)
    return t <= u.A ? 0 : t <= u.AD ? 1 : t <= u.ADS ? 2 : 3;
}
// The four pieces as a table in LDS, built once per workgroup (adsr_piece_table) and indexed per lane: written as a chain of
// selects on the lane's piece index the compiler builds the same table itself -- in scratch memory, whose per-wave
// allocation held the kernel at a quarter of its wave slots.
struct __attribute__((aligned(16))) AdsrPieceRow { float a, dv, s1, s2, s3, x; double rcp; };
TD_DEV void adsr_piece_table(const AdsrRunConsts& u, AdsrPieceRow* tab) {   // (one thread; straight-line stores)
    tab[0] = AdsrPieceRow{u.std_v, u.dv0, 0.0f, 0.0f, 0.0f, u.A, u.rA};
    tab[1] = AdsrPieceRow{u.att_v, u.dv1, u.A, 0.0f, 0.0f, u.D, u.rD};
    tab[2] = AdsrPieceRow{u.dec_v, u.dv2, u.A, u.D, 0.0f, u.S, u.rS};
    tab[3] = AdsrPieceRow{u.sus_v, u.dv3, u.A, u.D, u.S, u.R, u.rR};
}
TD_DEV AdsrPiece adsr_piece(const AdsrPieceRow* tab, int k) {
    const AdsrPieceRow r = tab[k];
    AdsrPiece p;
    p.k = k;
    p.a = r.a; p.dv = r.dv; p.s1 = r.s1; p.s2 = r.s2; p.s3 = r.s3; p.x = r.x; p.rcp = r.rcp;
    return p;
}
// mode 0: apply_adsr (adsr.rs:75-86), 1: apply_ads (adsr.rs:62-69): beyond the sustain ramp the level is sustain_vel
TD_DEV float adsr_piece_value(const AdsrPiece& p, float t, int mode, float sustain_vel) {
    if (mode == 1 && p.k == 3) return sustain_vel;
    float q = fdiv_rcp(((t - p.s1) - p.s2) - p.s3, p.x, p.rcp);
    if (p.k == 3) q = fminf(q, 1.0f);
    return p.a + q * p.dv;
}
struct AdsrVoiceRun { float t0, vel, rel; int mode; bool same; AdsrPiece pc; };   // mode 2: apply_r from `rel` (adsr.rs:71-73)
TD_DEV AdsrVoiceRun adsr_voice_run(const AdsrRunConsts& u, const AdsrPieceRow* tab, bool use_off, float4 v, float off_first, float off_last, bool& ok) {
    AdsrVoiceRun r;
    r.t0 = v.x; r.vel = v.y; r.rel = v.z;
    r.mode = use_off ? (v.z == 0.0f ? 1 : 2) : 0;
    const float tf = v.x + off_first, tl = v.x + off_last;
    ok = ok && tf >= 0.0f && tl >= tf;   // (NaN fails both)
    const int kf = adsr_piece_index(u, tf), kl = adsr_piece_index(u, tl);
    r.same = kf == kl;
    r.pc = adsr_piece(tab, kf);
    return r;
}
TD_DEV float adsr_voice_value(const AdsrRunConsts& u, const AdsrPieceRow* tab, const AdsrVoiceRun& r, float offset) {
    const float t = r.t0 + offset;
    if (r.mode == 2) return (r.rel + fminf(fdiv_rcp(t, u.R, u.rR), 1.0f) * (u.rel_v - r.rel)) * r.vel;
    if (r.same) return adsr_piece_value(r.pc, t, r.mode, u.sus_v) * r.vel;
    return adsr_piece_value(adsr_piece(tab, adsr_piece_index(u, t)), t, r.mode, u.sus_v) * r.vel;
}
TD_DEV void adsr_env_block(const AdsrVDesc& d, uint32_t bx, uint32_t M) {
    __shared__ AdsrPieceRow ptab[4];
    if (threadIdx.x == 0u) adsr_piece_table(adsr_run_consts(d), ptab);
    __syncthreads();
    const uint32_t m0 = (bx * kThreads + threadIdx.x) * (uint32_t)kEnvRun;
    float sq = 0.0f;   // sum of the squared gains of the lane's frames (-> AdsrVDesc::env_tile)
    if (m0 < M) {
    float* const out = d.env;
    const uint32_t mlast = min(m0 + (uint32_t)kEnvRun - 1u, M - 1u);
    const uint32_t it = find_interval(d.tab, m0);
    const uint32_t TD_GLOBAL* st = reinterpret_cast<const uint32_t TD_GLOBAL*>((const TD_GLOBAL char*)d.tab.istart);
    bool fast = d.tame != 0u && mlast - m0 == (uint32_t)kEnvRun - 1u && (it + 1u >= d.tab.n_int || st[it + 1u] > mlast);
    const uint32_t i0 = m0 % d.bl;
    fast = fast && i0 + (uint32_t)kEnvRun <= d.bl;   // (no block start inside the run: intervals begin there anyway)
    const AdsrRunConsts u = adsr_run_consts(d);
    AdsrVoiceRun pr{}, gr{};
    if (fast) {
        const float4 p = d.tab.voices[2u * it], g = d.tab.voices[2u * it + 1u];   // (t_off, vel, release_val, skip)
        fast = p.w == 0.0f;
        const float of = fdiv_rcp((float)i0, u.srf, u.rsr), ol = fdiv_rcp((float)(i0 + (uint32_t)kEnvRun - 1u), u.srf, u.rsr);
        pr = adsr_voice_run(u, ptab, d.use_off != 0u, p, of, ol, fast);
        gr = adsr_voice_run(u, ptab, d.use_off != 0u, g, of, ol, fast);
    }
    if (fast) {
        const float maxmul = d.use_max ? 1.0f : 0.0f, minmul = 1.0f - maxmul;
#pragma unroll 1
        for (uint32_t q = 0; q < (uint32_t)kEnvRun; q += 4u) {
            float v[4];
#pragma unroll
            for (uint32_t e = 0; e < 4u; ++e) {
                const float offset = fdiv_rcp((float)(i0 + q + e), u.srf, u.rsr);   // (float)(i % bl) / (float)sr
                const float pvel = adsr_voice_value(u, ptab, pr, offset), gvel = adsr_voice_value(u, ptab, gr, offset);
                const float av = fmaxf(pvel, gvel) * maxmul + fminf(pvel, gvel) * minmul;
                v[e] = lerpf(1.0f, av, d.wet);
            }
            sq += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            gstore4(out + m0 + q, make_float4(v[0], v[1], v[2], v[3]));
        }
    } else {
#pragma unroll 1
        for (uint32_t m = m0; m <= mlast; ++m) {
            const float v = adsr_vel(d, m);
            sq += v * v;
            out[m] = v;
        }
    }
    }
    // one wave = 512 consecutive frames = half a wave-tile of k_band_chain: the table holds the MEAN SQUARE of every half
    static_assert(kEnvRun * 64 * 2 == kTileFrames, "two waves of k_adsr_env cover one wave-tile");
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
    const uint32_t w0 = (bx * kThreads + (threadIdx.x & ~63u)) * (uint32_t)kEnvRun;   // the wave's first frame
    if ((threadIdx.x & 63u) == 0u && w0 < M && d.env_tile)
        d.env_tile[w0 / (uint32_t)(kEnvRun * 64)] = sq / (float)min((uint32_t)(kEnvRun * 64), M - w0);
}
__global__ __launch_bounds__(kThreads) void k_adsr_env(const AdsrVDesc* __restrict__ descs, uint32_t M) {
    adsr_env_block(descs[blockIdx.y], blockIdx.x, M);
}

// ------------------------------------------------------------------------------------------------
// k_sources: the launches of one level that read no edge buffer, as ONE grid
// ------------------------------------------------------------------------------------------------
// A level's source vertices (affine Synth, wavetable voice, SampleLerp) and the envelope buffers of the Adsr vertices (k_adsr_env)
// depend on nothing on the device and on one another: queued one behind the other on a stream each pays its own ramp and
// tail -- the envelope launch 18 us for 12 MB.  Here their workgroups are the parts of one 1-D grid, the longest-running
// family first (its slow tiles first inside it: IntervalTab::tile_order), the short ones filling its tail (SourceKind's
// order); every workgroup runs the block function of its family unchanged (same values as the separate launches by construction).
// KINDS: the families compiled in (the register budget is that of the largest); MINB: workgroups per CU to allocate for.
template <uint32_t KINDS, int MINB>
__global__ __launch_bounds__(kThreads, MINB) void k_sources(const SynthDesc* __restrict__ sd, const SampsynDesc* __restrict__ yd,
                                                            const LerpDesc* __restrict__ ld, const AdsrVDesc* __restrict__ ed,
                                                            const SourceGrid G, uint32_t M) {
    // (the descriptor arrays are kernel arguments of their own, `__restrict__` like the families' own kernels': read through a
    // pointer taken from a table in the argument block they were vector loads -- the envelope part's table build 25 dependent ones)
    const uint32_t b = blockIdx.x;
    if (G.zero_n16) {   // (one store per thread at most for a deep chain's 3.8 MB: ahead of the block's own work, off its critical path)
        uint4* const z = G.zero;
        for (uint32_t i = b * kThreads + threadIdx.x; i < G.zero_n16; i += gridDim.x * kThreads) z[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    if ((KINDS & (1u << SRC_SYNTH_AFFINE)) && b < G.end[SRC_SYNTH_AFFINE]) {
        const uint32_t by = b / G.gx[SRC_SYNTH_AFFINE];
        synth_affine_block(sd[by], b - by * G.gx[SRC_SYNTH_AFFINE], M);
    } else if ((KINDS & (1u << SRC_SAMPSYN)) && b < G.end[SRC_SAMPSYN]) {
        const uint32_t local = b - G.end[SRC_SAMPSYN - 1], by = local / G.gx[SRC_SAMPSYN];
        sampsyn_block(yd[by], local - by * G.gx[SRC_SAMPSYN], M);
    } else if ((KINDS & (1u << SRC_LERP)) && b < G.end[SRC_LERP]) {
        const uint32_t local = b - G.end[SRC_LERP - 1], by = local / G.gx[SRC_LERP];
        sample_lerp_block(ld[by], local - by * G.gx[SRC_LERP], M);
    } else if (KINDS & (1u << SRC_ENV)) {
        const uint32_t local = b - G.end[SRC_ENV - 1], by = local / G.gx[SRC_ENV];
        adsr_env_block(ed[by], local - by * G.gx[SRC_ENV], M);
    }
}

template <int TMODE>
__global__ __launch_bounds__(kThreads) void k_adsr(const AdsrVDesc* __restrict__ descs, uint32_t M) {
    const AdsrVDesc& d = descs[blockIdx.y];
    const uint32_t m0 = blockIdx.x * kTileFrames + 2 * threadIdx.x;
    const uint32_t m1 = m0 + kTileFrames / 2;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    sum_terms<TMODE>(term_tab(d.ins), d.k, m0, m1, M, a0, a1);
    if (m0 < M) {
        float2 a = adsr_frame(d, m0, make_float2(a0.x, a0.y));
        float2 b = (m0 + 1 < M) ? adsr_frame(d, m0 + 1, make_float2(a0.z, a0.w)) : make_float2(0.f, 0.f);
        store_pair(d.out, m0, M, epilogue4(make_float4(a.x, a.y, b.x, b.y), d.pg));
    }
    if (m1 < M) {
        float2 a = adsr_frame(d, m1, make_float2(a1.x, a1.y));
        float2 b = (m1 + 1 < M) ? adsr_frame(d, m1 + 1, make_float2(a1.z, a1.w)) : make_float2(0.f, 0.f);
        store_pair(d.out, m1, M, epilogue4(make_float4(a.x, a.y, b.x, b.y), d.pg));
    }
}

// ------------------------------------------------------------------------------------------------
// k_band_pass: exact sequential one-pole pair (extensions.rs:654-689)
// ------------------------------------------------------------------------------------------------
// One workgroup per vertex walks the chunk tile by tile:
//   A (256 lanes)  summed input tile -> LDS
//   B (lanes 0..3) the four recurrences  y += gamma * (x - y)  (low L, low R, high L, high R), in order
//   C (256 lanes)  cut / pass combination (incl. quirk Q7: right pass uses the LEFT cut), epilogue, store
// The recurrence is strictly sequential in the reference; this form keeps it bit-exact.
__global__ __launch_bounds__(kThreads) void k_band_pass(const BandDesc* __restrict__ descs, uint32_t M) {
    const BandDesc& d = descs[blockIdx.x];
    __shared__ float xs[kTileFrames * 2];
    __shared__ float ys[kTileFrames * 4];
    const uint32_t lane = threadIdx.x;
    float y = 0.0f;
    const uint32_t c = lane & 3u;
    const float gam = (c & 2u) ? d.hgamma : d.lgamma;
    bool first = d.state->first != 0 || d.first_override != 0u;
    if (lane < 4) y = reinterpret_cast<const float*>(d.state)[lane];
    const float lmul = d.lgamma == 0.0f ? 0.0f : 1.0f;
    const float hmul = d.hgamma == 0.0f ? 0.0f : 1.0f;
    const float pass_mul = d.pass ? 1.0f : 0.0f;
    const float cut_mul = 1.0f - pass_mul;
    for (uint32_t base = 0; base < M; base += kTileFrames) {
        const uint32_t n_tile = min((uint32_t)kTileFrames, M - base);
        // phase A
        {
            float4 a0, a1;
            const uint32_t m0 = base + 2 * lane, m1 = m0 + kTileFrames / 2;
            sum_inputs_pairs(d.ins, d.k, d.term_mode, m0, m1, M, a0, a1);
            reinterpret_cast<float4*>(xs)[lane] = a0;
            reinterpret_cast<float4*>(xs)[lane + kTileFrames / 4] = a1;
        }
        __syncthreads();
        // phase B
        if (lane < 4) {
            if (first) { y = xs[c & 1u]; first = false; }   // extensions.rs:664-670: seed from buf[0]
            const uint32_t ch = c & 1u;
            uint32_t n = 0;
            for (; n + 8 <= n_tile; n += 8) {
                float x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) x[u] = xs[(n + u) * 2 + ch];
#pragma unroll
                for (int u = 0; u < 8; ++u) { y = y + gam * (x[u] - y); ys[(n + u) * 4 + c] = y; }
            }
            for (; n < n_tile; ++n) { y = y + gam * (xs[n * 2 + ch] - y); ys[n * 4 + c] = y; }
        }
        __syncthreads();
        // phase C
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t f = 2 * lane + h * (kTileFrames / 2);   // frame pair within the tile
            const uint32_t m = base + f;
            if (m < M) {
                float o[4];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float l = xs[(f + e) * 2], r = xs[(f + e) * 2 + 1];
                    const float4 s = reinterpret_cast<const float4*>(ys)[f + e];   // ll, lr, hl, hr
                    const float cutl = (lmul * s.x + hmul * (l - s.z)) * 0.5f;
                    const float cutr = (lmul * s.y + hmul * (r - s.w)) * 0.5f;
                    const float passl = l - cutl;
                    const float passr = r - cutl;   // extensions.rs:685 (Q7)
                    o[2 * e] = cutl * cut_mul + passl * pass_mul;
                    o[2 * e + 1] = cutr * cut_mul + passr * pass_mul;
                }
                store_pair(d.out, m, M, epilogue4(make_float4(o[0], o[1], o[2], o[3]), d.pg));
            }
        }
        __syncthreads();
    }
    if (lane < 4) reinterpret_cast<float*>(d.state)[lane] = y;
    if (lane == 0) d.state->first = first ? 1u : 0u;
}

// ------------------------------------------------------------------------------------------------
// k_band_spec / k_band_fix: the same filter, exact and parallel (see BandSpecDesc in kernels.h)
// ------------------------------------------------------------------------------------------------
template <int J>
TD_DEV float quad_bcast(float v) {   // value of lane J of this lane's quad (DPP quad_perm, no LDS)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), J * 0x55, 0xF, 0xF, true));
}
struct BandCoef { float lmul, hmul, pass_mul, cut_mul; };
TD_DEV BandCoef band_coef(float lgamma, float hgamma, uint32_t pass) {
    BandCoef k;
    k.lmul = lgamma == 0.0f ? 0.0f : 1.0f;
    k.hmul = hgamma == 0.0f ? 0.0f : 1.0f;
    k.pass_mul = pass ? 1.0f : 0.0f;
    k.cut_mul = 1.0f - k.pass_mul;
    return k;
}
// one output frame from the quad's four states and the input frame (extensions.rs:682-687, quirk Q7)
TD_DEV float2 band_out(const BandCoef& k, float l, float r, float ll, float lr, float hl, float hr) {
    const float cutl = (k.lmul * ll + k.hmul * (l - hl)) * 0.5f;
    const float cutr = (k.lmul * lr + k.hmul * (r - hr)) * 0.5f;
    const float passl = l - cutl;
    const float passr = r - cutl;
    return make_float2(cutl * k.cut_mul + passl * k.pass_mul, cutr * k.cut_mul + passr * k.pass_mul);
}

constexpr uint32_t kNoJob = 0xFFFFFFFFu;   // BandSpecDesc::seg_job: the segment lies in no parked stretch
__global__ __launch_bounds__(kThreads) void k_band_spec(const BandSpecDesc* __restrict__ descs, uint32_t M) {
    const BandSpecDesc& d = descs[blockIdx.y];
    if (blockIdx.x == 0 && threadIdx.x < 40u) d.stats[threadIdx.x] = 0u;   // k_band_fix's counters and ticket [0..7]; verdict and fill claims [32..33]
    const uint32_t c = threadIdx.x & 3u;                       // chain: 0 low L, 1 low R, 2 high L, 3 high R
    const uint32_t seg_raw = blockIdx.x * (kThreads / 4) + (threadIdx.x >> 2);
    if (blockIdx.x * (kThreads / 4) >= d.nseg) return;         // whole workgroup beyond this vertex' segments
    // quads past the last segment shadow it (cross-lane reductions below need every lane of the wave) and
    // store nothing
    const bool live = seg_raw < d.nseg;
    const uint32_t seg = live ? seg_raw : d.nseg - 1u;
    if (live && c == 0u) d.seg_job[seg_raw] = kNoJob;   // (k_band_fix parks stretches; whoever fills them finds every other segment marked free)
    const uint32_t ch = c & 1u;
    const float gam = (c & 2u) ? d.hgamma : d.lgamma;
    const BandCoef kf = band_coef(d.lgamma, d.hgamma, d.pass);
    const float* __restrict__ qf = reinterpret_cast<const float*>(d.xq4);   // the planar-in-4 input: the only copy (round 6)
    const uint32_t start = seg * d.S;
    const uint32_t end = min(start + d.S, M);
    // exact state at the chunk's first frame: carried, or seeded from buf[0] (extensions.rs:664-670)
    const float y_true0 = (d.state->first || d.first_override) ? q4_chan(qf, 0u, ch) : gload1(reinterpret_cast<const float*>(d.state) + c);
    const uint32_t M2 = ((M + 3u) & ~3u) >> 1;               // 16-byte words of the planar copy (padded to whole fours)
    // warm-up: recurrence only, fed from the planar-in-4 copy: lane c of the quad loads 16-byte word c of an
    // 8-frame group = {L 0..3 | R 0..3 | L 4..7 | R 4..7}, so step j of the group needs register j & 3 of lane
    // (j >> 2) * 2 + ch -- a DPP quad_perm ([0,1,0,1] / [2,3,2,3]) applied right in the subtract's operand.
    // One 64-byte fetch per quad per 8 steps, three dependent VALU instructions per step and nothing else.
    // n and start are multiples of 32 (S and W are), so the loop runs in batches of 32 frames: four fetches
    // issued one whole batch ahead -- ~32 steps of dependent VALU cover an Infinity Cache / L2 round trip.
    const float4* __restrict__ q4 = reinterpret_cast<const float4*>(d.xq4);
    auto fetchq = [&](uint32_t frame) -> float4 {            // frame is a multiple of 8
        return gload4(q4 + min((frame >> 1) + c, M2 - 1u));  // (prefetches may run past the last stepped frame)
    };
    // The eight steps are written as one instruction block: the compiler's DPP hazard rule also pads a DPP
    // instruction whose NON-DPP operand (y, just written by the previous add) is fresh -- 3.3 ns of s_nop per
    // 7.7 ns step.  The rule that matters in hardware is about the DPP-read register; those are load results,
    // and the one s_nop at the top covers a compiler copy placed right before the block.
#define TD_BAND_S(X, PERM)                                                                          \
    "v_sub_f32_dpp %1, " X ", %0 quad_perm:" PERM " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"      \
    "v_mul_f32 %1, %2, %1\n"                                                                        \
    "v_add_f32 %0, %0, %1\n"
#define TD_BAND_STEP8(A)                                                                            \
    {                                                                                               \
        const float4 a = A;                                                                         \
        float t_;                                                                                   \
        asm volatile("s_nop 1\n"                                                                    \
                     TD_BAND_S("%3", "[0,1,0,1]") TD_BAND_S("%4", "[0,1,0,1]")                      \
                     TD_BAND_S("%5", "[0,1,0,1]") TD_BAND_S("%6", "[0,1,0,1]")                      \
                     TD_BAND_S("%3", "[2,3,2,3]") TD_BAND_S("%4", "[2,3,2,3]")                      \
                     TD_BAND_S("%5", "[2,3,2,3]") TD_BAND_S("%6", "[2,3,2,3]")                      \
                     : "+v"(y), "=&v"(t_)                                                           \
                     : "v"(gam), "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w));                           \
    }
    // Warm-up length.  The long warm-up W covers a full-scale tail decaying to the denormal floor (needed when
    // the segment sits in or right after a silence or a held constant, where trajectories park instead of
    // meeting).  Otherwise the short window Ws = 40 / gamma is enough if the energy the input feeds in INSIDE
    // the window dominates, at the segment start, what is left of everything before it: a state is an
    // exponentially weighted memory of the input, so with E(blocks) = max_b peak_b * e^(-gamma * distance of
    // block b to the segment start), the wrong part of the guess is at most A = E(blocks of the Ws frames before
    // the window) and the state itself about B = E(blocks of the window).  A <= live_thr * B (1e-9: below half an
    // ulp with a margin for states much smaller than the input peak) means the guess error ends far below the
    // state's last bit.  (A window that merely decays does NOT qualify -- error and state shrink together -- a
    // window in which the next sound starts does, however many decades the level swings in between.)  Block
    // values come for free from the k_sum launch that materialised the input; held-constant blocks (-1) count as
    // no energy and must not lie in the last 20 / gamma frames.  A wrong pick only costs a repair in k_band_fix,
    // never exactness.
    uint32_t my_w = d.W;
    // (the workgroup's 64 segments are neighbours, their block ranges overlap almost entirely: staged in LDS once)
    __shared__ float pk_l[1024];
    const uint32_t nw_ = d.Ws / 256u, s256 = d.S / 256u;
    const uint32_t seg_first = blockIdx.x * (kThreads / 4), seg_last = min(seg_first + kThreads / 4 - 1u, d.nseg - 1u);
    const uint32_t pk_base = seg_first * s256 > 2u * nw_ ? seg_first * s256 - 2u * nw_ : 0u;
    const uint32_t pk_cnt = seg_last * s256 - pk_base;
    const bool pk_lds = pk_cnt <= 1024u && d.Ws < d.W;
    if (pk_lds)
        for (uint32_t i = threadIdx.x; i < pk_cnt; i += kThreads) pk_l[i] = d.blk_peaks[pk_base + i];
    // ... and so are the block responses of the warm-up guess: the quick windows of the workgroup's 64 neighbouring
    // segments end at consecutive blocks, each looks Kmax blocks back (<= 63 + 256 blocks of 4 doubles)
    __shared__ double resp_l[(kThreads / 4 + 256) * 4];
    const uint32_t Kmax = max(d.Kl, d.Kh), wq_blk = d.Wq >> 8, wq2_blk = max(d.Wq, d.Wq2) >> 8;
    uint32_t rs_base = 0u, rs_cnt = 0u;
    if (d.resp && seg_last * s256 > wq_blk) {
        const uint32_t hi_blk = seg_last * s256 - wq_blk;                                  // one past the last block any window needs
        const uint32_t lo_seg_blk = seg_first * s256 > wq2_blk ? seg_first * s256 - wq2_blk : 0u;
        rs_base = lo_seg_blk > Kmax ? lo_seg_blk - Kmax : 0u;
        rs_cnt = min(hi_blk - rs_base, (uint32_t)(kThreads / 4 + 256));
        for (uint32_t i = threadIdx.x; i < rs_cnt * 4u; i += kThreads) resp_l[i] = d.resp[rs_base * 4u + i];
    }
    __syncthreads();
    auto blk_peak = [&](uint32_t b) -> float { return pk_lds ? pk_l[b - pk_base] : d.blk_peaks[b]; };
    // window_live(w, strict): may the warm-up be the w frames before the segment?  (quad-uniform)  strict: the energy
    // criterion above, for a warm-up that starts from a wrong value; a warm-up that starts from the block-response guess
    // already carries everything before the window exactly and only needs the window to be alive (no parked stretch).
    auto window_live = [&](uint32_t w, float thr) -> bool {
        if (!(start > w)) return false;
        const uint32_t nw = w / 256u, b1 = start / 256u, b0 = b1 - nw, bA = b0 > nw ? b0 - nw : 0u;
        // (near the chunk start the history before the window is the carried state itself, decayed to here)
        const float A0 = bA == 0u ? fabsf(y_true0) * __expf(-d.gmin * (float)start) : 0.0f;
        float A = 0.0f, B = 0.0f, tail_min = 1.0f;
        uint32_t b = bA + c;
        for (; b < b0; b += 4u) A = fmaxf(A * d.decay4, blk_peak(b));
        // lane-local reference point: block b - 4; bring A forward as the window is walked
        for (; b < b1; b += 4u) {
            const float pk = blk_peak(b);
            A *= d.decay4;
            B = fmaxf(B * d.decay4, pk);
            if (b + d.post_blocks >= b1) tail_min = fminf(tail_min, pk);
        }
        // the block maxima are now referred to block b - 4 (this lane's last); refer them to the segment start
        float tail = 1.0f;
        for (uint32_t r = b - 4u + 1u; r < b1; ++r) tail *= d.decay1;
        B *= tail;
        A = fmaxf(A * tail, A0);
        A = fmaxf(fmaxf(quad_bcast<0>(A), quad_bcast<1>(A)), fmaxf(quad_bcast<2>(A), quad_bcast<3>(A)));
        B = fmaxf(fmaxf(quad_bcast<0>(B), quad_bcast<1>(B)), fmaxf(quad_bcast<2>(B), quad_bcast<3>(B)));
        tail_min = fminf(fminf(quad_bcast<0>(tail_min), quad_bcast<1>(tail_min)), fminf(quad_bcast<2>(tail_min), quad_bcast<3>(tail_min)));
        return tail_min >= 0.0f && B >= 1e-30f && A <= B * thr;
    };
    if (d.Ws < d.W) {
        // with block responses the quick window is tried first (its own, shorter, liveness windows), then the short one
        // What a warm-up from the guess still has to outlast is the rounding residue of louder times: the true f32 state
        // carries ~2^-24 of everything it ever held, decaying like the state itself, and exact arithmetic knows nothing
        // of it.  Quick window: the energy from before it is small against the energy inside (quick_thr, ~2^24 x the
        // garbage-guess threshold); medium window: long enough for 8 decades of level drop (measured on the 84-stage
        // chain); both need the window alive.
        if (d.resp && d.Wq < d.Ws && window_live(d.Wq, d.quick_thr)) my_w = d.Wq;
        else if (d.resp && d.Wq2 < d.Ws && window_live(d.Wq2, 3.0e38f)) my_w = d.Wq2;
        else if (window_live(d.Ws, d.live_thr)) my_w = d.Ws;
    }
    // Every lane of the wave walks the same number of steps -- the longest warm-up any of its quads picked (a
    // longer warm-up than needed never hurts) -- over its OWN window [start - w, start).  Quads whose window is
    // cut short by the chunk start begin at frame 0 with the exact state and simply finish earlier.
    // (a quad whose window the chunk start cuts short asks for no more than the frames it has: with segments of several
    // thousand frames the first wave's early quads would otherwise send its later ones all the way back to frame 0)
    uint32_t w_wave = min(my_w, start);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) w_wave = max(w_wave, (uint32_t)__shfl_xor((int)w_wave, off, 64));
    const uint32_t my_begin = start > w_wave ? start - w_wave : 0u;
    uint32_t n = my_begin;
    // exact state at frame 0, constant chain, or a guess.  With block responses (BandRespParam) the guess is the
    // exact-arithmetic state at the window start -- the responses of the last K blocks chained in double -- which
    // the true f32 trajectory only leaves by its own accumulated rounding (a few ulp): the quick warm-up Wq then only
    // has to let the two trajectories MEET (a gap of an ulp closes with probability gamma per step), not to forget a
    // wrong starting value first.  Without them the guess is the input frame itself.
    float y;
    if (my_begin == 0u || gam == 0.0f) {
        y = y_true0;
    } else if (d.resp && (my_begin & 255u) == 0u && ((c & 2u) ? d.Kh : d.Kl) != 0u) {   // (K == 0: a fast smoother, no responses)
        const double A = (c & 2u) ? d.Ah : d.Al;
        const uint32_t K = (c & 2u) ? d.Kh : d.Kl;
        const uint32_t nblk = my_begin >> 8, first = nblk > K ? nblk - K : 0u;
        double yd = first == 0u ? (double)y_true0 : 0.0;
        if (first >= rs_base && nblk <= rs_base + rs_cnt) {
            const double* rl = resp_l + (size_t)(first - rs_base) * 4u + c;
            uint32_t b = first;
            for (; b + 4u <= nblk; b += 4u, rl += 16) {   // (four independent LDS reads per trip)
                const double r0 = rl[0], r1 = rl[4], r2 = rl[8], r3 = rl[12];
                yd = yd * A + r0; yd = yd * A + r1; yd = yd * A + r2; yd = yd * A + r3;
            }
            for (; b < nblk; ++b, rl += 4) yd = yd * A + rl[0];
        } else {   // (a window other than the quick one: straight from memory)
            const double* rg = d.resp + (size_t)first * 4u + c;
            uint32_t b = first;
            for (; b + 4u <= nblk; b += 4u, rg += 16) {
                const double r0 = rg[0], r1 = rg[4], r2 = rg[8], r3 = rg[12];
                yd = yd * A + r0; yd = yd * A + r1; yd = yd * A + r2; yd = yd * A + r3;
            }
            for (; b < nblk; ++b, rg += 4) yd = yd * A + rg[0];
        }
        y = (float)yd;
    } else {
        y = q4_chan(qf, my_begin, ch);
    }
    if (n + 32u <= start && (n & 31u) == 0u) {
        // three register sets in rotation (no copies): each batch of 32 steps runs on loads issued two batches
        // (~0.5 us) earlier -- every quad streams its own window, mostly out of L2
        float4 a0 = fetchq(n), a1 = fetchq(n + 8u), a2 = fetchq(n + 16u), a3 = fetchq(n + 24u);
        float4 b0 = fetchq(n + 32u), b1 = fetchq(n + 40u), b2 = fetchq(n + 48u), b3 = fetchq(n + 56u);
        while (n + 96u <= start) {
            const float4 c0 = fetchq(n + 64u), c1 = fetchq(n + 72u), c2 = fetchq(n + 80u), c3 = fetchq(n + 88u);
            TD_BAND_STEP8(a0) TD_BAND_STEP8(a1) TD_BAND_STEP8(a2) TD_BAND_STEP8(a3)
            n += 32u;
            a0 = fetchq(n + 64u); a1 = fetchq(n + 72u); a2 = fetchq(n + 80u); a3 = fetchq(n + 88u);
            TD_BAND_STEP8(b0) TD_BAND_STEP8(b1) TD_BAND_STEP8(b2) TD_BAND_STEP8(b3)
            n += 32u;
            b0 = fetchq(n + 64u); b1 = fetchq(n + 72u); b2 = fetchq(n + 80u); b3 = fetchq(n + 88u);
            TD_BAND_STEP8(c0) TD_BAND_STEP8(c1) TD_BAND_STEP8(c2) TD_BAND_STEP8(c3)
            n += 32u;
        }
        if (n + 32u <= start) {
            TD_BAND_STEP8(a0) TD_BAND_STEP8(a1) TD_BAND_STEP8(a2) TD_BAND_STEP8(a3)
            n += 32u;
        }
        if (n + 32u <= start) {
            TD_BAND_STEP8(b0) TD_BAND_STEP8(b1) TD_BAND_STEP8(b2) TD_BAND_STEP8(b3)
            n += 32u;
        }
    }
    if ((n & 7u) == 0u)
        for (; n + 8u <= start; n += 8u) { TD_BAND_STEP8(fetchq(n)) }
#undef TD_BAND_STEP8
#undef TD_BAND_S
    for (; n < start; ++n) y = y + gam * (q4_chan(qf, n, ch) - y);
    if (live) d.seg_start[seg * 4u + c] = y;
    // the segment itself: recurrence + output
    const float x_first = q4_chan(qf, start, ch);
    // Fast form (every quad of the wave owns a whole segment): the wave alternates between
    //   (a) 32 steps of the bare recurrence, like the warm-up, each lane dropping its state after every step
    //       into LDS (one ds_write per step; quad slots padded to 528 B so a wave's 64 lanes hit 64 banks), and
    //   (b) the output of those 32 frames x 16 segments with all 64 lanes: frame-parallel, coalesced 256-byte
    //       runs of input and output instead of one 8-byte store per quad per step.
    // The "input constant / all zero" flags of the segments fall out of (b) as ballots.
    constexpr uint32_t kQStride = 132;   // floats per quad slot: 32 frames x 4 states + 4 pad
    __shared__ __attribute__((aligned(16))) float ys_l[(kThreads / 4) * kQStride];
    __shared__ float2 xf_l[kThreads / 4];
    // Every quad takes its whole 32-frame pieces this way -- the chunk's last segment too, which is rarely a whole one (round 6:
    // it used to send its whole wave through the per-step form below, ~150 ns a frame: unnoticed behind a 256-frame segment's
    // warm-up, the longest thing in the launch behind a 4 096-frame one).  The wave runs as many pieces as its longest quad has;
    // a quad with fewer keeps stepping on clamped loads and stores nothing, its state is taken where its own pieces end.
    const uint32_t whole_q = (end - start) & ~31u;   // (quad-uniform; quads past the last segment mirror it)
    uint32_t len_w = whole_q;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) len_w = max(len_w, (uint32_t)__shfl_xor((int)len_w, off, 64));
    bool same = true, zero = true;
    uint32_t n_done = 0u;                            // frames of this quad's segment the piece form has covered
    if (len_w != 0u) {
        const uint32_t quad = threadIdx.x >> 2, lane = threadIdx.x & 63u, wq0 = (threadIdx.x >> 6) * 16u;
        float* myq = ys_l + quad * kQStride + c;
        const float xr_first = quad_bcast<1>(x_first);
        if (c == 0u) xf_l[quad] = make_float2(x_first, xr_first);
        uint32_t same_mask = 0xFFFFu, zero_mask = 0xFFFFu;   // bit q: segment of the wave's quad q
        const uint32_t wave_seg0 = blockIdx.x * (kThreads / 4) + wq0;
#define TD_BAND_S1(X, PERM, J)                                                                                  \
    asm volatile("v_sub_f32_dpp %1, %3, %0 quad_perm:" PERM " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"        \
                 "v_mul_f32 %1, %2, %1\n"                                                                       \
                 "v_add_f32 %0, %0, %1\n"                                                                       \
                 : "+v"(y), "=&v"(t_) : "v"(gam), "v"(X));                                                      \
    myq[(J) * 4] = y;
#define TD_BAND_STEP8W(A, J0)                                                                                   \
    {                                                                                                           \
        const float4 a = A;                                                                                     \
        float t_;                                                                                               \
        asm volatile("s_nop 1" ::: "memory");                                                                   \
        TD_BAND_S1(a.x, "[0,1,0,1]", (J0) + 0) TD_BAND_S1(a.y, "[0,1,0,1]", (J0) + 1)                           \
        TD_BAND_S1(a.z, "[0,1,0,1]", (J0) + 2) TD_BAND_S1(a.w, "[0,1,0,1]", (J0) + 3)                           \
        TD_BAND_S1(a.x, "[2,3,2,3]", (J0) + 4) TD_BAND_S1(a.y, "[2,3,2,3]", (J0) + 5)                           \
        TD_BAND_S1(a.z, "[2,3,2,3]", (J0) + 6) TD_BAND_S1(a.w, "[2,3,2,3]", (J0) + 7)                           \
    }
        float4 a0 = fetchq(start), a1 = fetchq(start + 8u), a2 = fetchq(start + 16u), a3 = fetchq(start + 24u);
        // (b)'s input frames (interleaved copy) of the lane's eight (segment, frame) slots: loaded a whole 32-frame piece
        // ahead, like the recurrence's own input -- the eight loads of a piece would otherwise be exposed one L2 round trip
        // after the other between the dependent chains
        // A lone wave issues one instruction per ~2.1 ns whatever its kind (tools/ubench/issue_rate.hip), so (b) costs what
        // it COUNTS: everything that does not change from piece to piece -- the slots' frame offsets, LDS addresses and
        // first frames, the descriptor's pointers and pan / gain (read through `d` they would be re-fetched by scalar loads
        // after every global store, which may alias the descriptor for all the compiler knows) -- is worked out once, and
        // the segments' "constant" / "zero" verdicts are eight ballots after the last piece instead of two per slot.
        float2* __restrict__ const out_p = d.out;
        const PanGain pg = d.pg;
        const uint32_t S = d.S;
        uint32_t m0_[8];          // frame of slot i at p = 0 (quads past the end mirror the last segment)
        uint32_t slot_end[8];     // one past the last frame slot i's segment takes this way (0: a quad past the last segment)
        const float* ys_rd[8];    // the slot's four states in the wave's LDS staging
        float2 xo[8], x0r[8];
        uint32_t not_same = 0u, not_zero = 0u;   // bit i: slot i
#pragma unroll
        for (uint32_t i = 0; i < 8u; ++i) {
            const uint32_t idx = i * 64u + lane, sq = idx >> 5, j = idx & 31u, sraw = wave_seg0 + sq;
            const uint32_t s0_ = min(sraw, d.nseg - 1u) * S;
            m0_[i] = s0_ + j;
            slot_end[i] = sraw < d.nseg ? s0_ + ((min(s0_ + S, M) - s0_) & ~31u) : 0u;
            ys_rd[i] = ys_l + (wq0 + sq) * kQStride + j * 4u;
            xo[i] = q4_frame(qf, min(m0_[i], M - 1u));
        }
        float y_cap = y;
        for (uint32_t p = 0; p < len_w; p += 32u) {
            TD_BAND_STEP8W(a0, 0) TD_BAND_STEP8W(a1, 8) TD_BAND_STEP8W(a2, 16) TD_BAND_STEP8W(a3, 24)
            if (p + 32u == whole_q) y_cap = y;   // (this quad's own pieces end here)
            if (p + 32u < len_w) {   // the next 32 frames' input flies during (b)
                a0 = fetchq(start + p + 32u); a1 = fetchq(start + p + 40u);
                a2 = fetchq(start + p + 48u); a3 = fetchq(start + p + 56u);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
#pragma unroll
            for (uint32_t i = 0; i < 8u; ++i) {
                const uint32_t m = m0_[i] + p;
                const float2 x = xo[i];
                if (p + 32u < len_w) xo[i] = q4_frame(qf, min(m + 32u, M - 1u));
                const float4 s = *reinterpret_cast<const float4*>(ys_rd[i]);
                if (p == 0u) x0r[i] = xf_l[wq0 + ((i * 64u + lane) >> 5)];
                const float2 x0 = x0r[i];
                const bool sm = __float_as_uint(x.x) == __float_as_uint(x0.x) && __float_as_uint(x.y) == __float_as_uint(x0.y);
                const bool zr = x.x == 0.0f && x.y == 0.0f;
                if (m < slot_end[i]) {
                    not_same |= (sm ? 0u : 1u) << i;
                    not_zero |= (zr ? 0u : 1u) << i;
                    gstore2(out_p + m, epilogue(band_out(kf, x.x, x.y, s.x, s.y, s.z, s.w), pg));
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        }
#undef TD_BAND_STEP8W
#undef TD_BAND_S1
#pragma unroll
        for (uint32_t i = 0; i < 8u; ++i) {   // slot i: lanes 0..31 hold segment 2 i of the wave, lanes 32..63 segment 2 i + 1
            const unsigned long long bs = __ballot(((not_same >> i) & 1u) ? 1 : 0), bz = __ballot(((not_zero >> i) & 1u) ? 1 : 0);
            if ((uint32_t)bs != 0u) same_mask &= ~(1u << (2u * i));
            if ((uint32_t)(bs >> 32) != 0u) same_mask &= ~(2u << (2u * i));
            if ((uint32_t)bz != 0u) zero_mask &= ~(1u << (2u * i));
            if ((uint32_t)(bz >> 32) != 0u) zero_mask &= ~(2u << (2u * i));
        }
        y = y_cap;
        n_done = whole_q;
        const uint32_t q = quad & 15u;
        same = ((same_mask >> q) & 1u) != 0u;    // (so far: the frames below go on from here, per channel)
        zero = ((zero_mask >> q) & 1u) != 0u;
    }
    // ... and what is left of the segment -- fewer than 32 frames, or all of it in a chunk of a few frames -- step by step
    auto step = [&](uint32_t m, float l, float r) {
        if (m >= end) return;
        const float x = ch ? r : l;
        same = same && (__float_as_uint(x) == __float_as_uint(x_first));
        zero = zero && (x == 0.0f);
        y = y + gam * (x - y);
        const float2 o = band_out(kf, l, r, quad_bcast<0>(y), quad_bcast<1>(y), quad_bcast<2>(y), quad_bcast<3>(y));
        if (c == 0u && live) gstore2(d.out + m, epilogue(o, d.pg));
    };
    for (n = start + n_done; n < end; ++n) { const float2 x = q4_frame(qf, n); step(n, x.x, x.y); }
    if (live) d.seg_final[seg * 4u + c] = y;
    const float s0 = quad_bcast<0>(same ? 1.0f : 0.0f), s1 = quad_bcast<1>(same ? 1.0f : 0.0f);
    const float z0 = quad_bcast<0>(zero ? 1.0f : 0.0f), z1 = quad_bcast<1>(zero ? 1.0f : 0.0f);
    const float xr_first = quad_bcast<1>(x_first);
    if (c == 0u && live) {
        d.seg_flags[seg] = ((s0 != 0.0f && s1 != 0.0f) ? 1u : 0u) | ((z0 != 0.0f && z1 != 0.0f) ? 2u : 0u);
        d.seg_x0[seg] = make_float2(x_first, xr_first);
    }
}

// One workgroup per vertex, 16 waves.  Rounds of
//   Phase A (1024 lanes): bitmap + ordered list (LDS) of the segments whose entry state differs from the
//            predecessor's exit state ("events");
//   Phase B: the waves take the events round-robin and repair each one's cascade INDEPENDENTLY, assuming the
//            predecessor's stored exit state is the true one -- true for the first event, and for every
//            event that no earlier cascade reaches.  A cascade never enters a segment another event owns.
// until Phase A finds nothing.  Every repair keeps the arrays honest -- seg_start = the entry state the
// segment's stored output and seg_final were computed from -- so the final all-clear Phase A IS the proof
// (induction from the exactly-known first segment); an optimistic repair that started from a stale state
// shows up as a mismatch in the next round.  The lowest event of a round is always repaired for good, so
// the rounds terminate; in practice events are far apart (one per silent stretch) and two rounds do.
constexpr int kFixThreads = 1024, kFixWaves = kFixThreads / 64;
constexpr uint32_t kFixMaxSegs = 131072;   // LDS bitmap capacity (host picks S accordingly)
constexpr uint32_t kFixMaxEvents = 2048;   // events repaired per round (the rest wait for the next round)

// One event: wave-wide.  Lanes 0..3 = the four chains on the TRUE trajectory, lanes 4..7 re-run the
// speculative one from the entry state pass 1 used (as soon as the two are bit-identical the rest of the
// segment -- output and exit state -- is already right; checked every 8 steps); the other lanes idle along
// as copies of quad 0.  xs / ys: this wave's LDS staging for 64 frames of input / state.
TD_DEV void band_fix_cascade(const BandSpecDesc& d, uint32_t M, uint32_t seg, uint32_t lo, uint32_t hi, const uint32_t* bitmap,
                             float2* xs, float* ys, uint32_t& recomputed, uint32_t& parked_segs) {
    const uint32_t lane = threadIdx.x & 63u, c = lane & 3u, ch = c & 1u;
    const float gam = (c & 2u) ? d.hgamma : d.lgamma;
    const BandCoef kf = band_coef(d.lgamma, d.hgamma, d.pass);
    // (read once: through `d` they would be re-fetched by scalar loads after every global store of the loops below)
    const float* __restrict__ const qf = reinterpret_cast<const float*>(d.xq4);   // (the planar-in-4 input)
    float2* const out_p = d.out;
    const PanGain pg = d.pg;
    const uint32_t S = d.S;
    const uint32_t* const seg_flags = d.seg_flags;
    uint32_t* Su = reinterpret_cast<uint32_t*>(d.seg_start);
    uint32_t* Fu = reinterpret_cast<uint32_t*>(d.seg_final);
    uint4* S4 = reinterpret_cast<uint4*>(d.seg_start);
    uint4* F4 = reinterpret_cast<uint4*>(d.seg_final);
    const uint2* X0 = reinterpret_cast<const uint2*>(d.seg_x0);
    const float* xsf = reinterpret_cast<const float*>(xs);
    auto flagged = [&](uint32_t s) { return ((bitmap[(s - lo) >> 5] >> ((s - lo) & 31u)) & 1u) != 0u; };   // lo <= s < hi
    // exit state of the predecessor (lanes 0..3 own every store to seg_final this wave makes)
    float y = __shfl(__uint_as_float(Fu[(seg - 1u) * 4u + c]), (int)c, 64);
    bool first_round = true;
    for (;;) {
        const uint32_t start = seg * S, end = min(start + S, M), len = end - start;
        // everything this round needs from global memory, issued together (one round trip)
        const uint32_t flags = seg_flags[seg];
        const uint2 x0 = X0[seg];
        const uint32_t su = Su[seg * 4u + c];
        const uint32_t fu_old = Fu[seg * 4u + c];
        if (!first_round) {
            if (flagged(seg)) break;   // another event's segment: its own cascade (this round or the next) takes over
            if (__any((lane < 4u && su != __float_as_uint(y)) ? 1 : 0) == 0) break;   // entered with exactly the new state
        }
        first_round = false;
        if (lane < 4u) Su[seg * 4u + c] = __float_as_uint(y);   // the entry state everything below is computed from
        uint32_t last = seg;   // last segment covered by this round
        uint32_t job = kNoJob;
        // A segment that an earlier repair left (partly) to a pending job has stored output that is not all
        // in memory yet: "the rest already stands" cannot be used there, it is stepped through (or parked anew).
        const bool had_job = d.seg_job[seg] != kNoJob;
        float yy = (lane >= 4u && lane < 8u) ? __uint_as_float(su) : y;
        uint32_t n = 0;
        bool parked = false, coalesced = false;
        while (n < len && !parked && !coalesced) {
            // the next 64 frames of input (most repairs park or coalesce within the first few frames)
            const uint32_t cl = min(64u, len - n);
            if (lane < cl) xs[lane] = q4_frame(qf, start + n + lane);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            uint32_t k = 0;
            while (k < cl && !parked && !coalesced) {
                // A run of bit-identical input frames on which all eight trajectories (the true one and the speculative
                // twin) stand still -- the silence in front of a hit inside a mixed segment, which no parking covers:
                // one step proves it for the whole run, the states of the run are filled in by all lanes.
                if (!flags) {   // (a segment of constant / all-zero input parks instead, below)
                    const uint2 xk = reinterpret_cast<const uint2*>(xs)[k], xl = reinterpret_cast<const uint2*>(xs)[min(lane, cl - 1u)];
                    const unsigned long long same = __ballot((lane >= k && lane < cl && xl.x == xk.x && xl.y == xk.y) ? 1 : 0) >> k;
                    const uint32_t run = same == ~0ull ? 64u : (uint32_t)__ffsll((long long)~same) - 1u;   // (frames k .. k + run - 1)
                    if (run >= 16u) {
                        const float yn = yy + gam * (xsf[2u * k + ch] - yy);
                        if (__all((lane >= 8u || __float_as_uint(yn) == __float_as_uint(yy)) ? 1 : 0)) {
                            const float yc = __shfl(yy, (int)c, 64);   // the true state of chain c, in every lane
                            for (uint32_t f = lane >> 2; f < run; f += 16u) ys[(k + f) * 4u + c] = yc;
                            k += run;
                            continue;
                        }
                    }
                }
                const uint32_t nb = min(8u, cl - k);
                float xv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) xv[u] = (uint32_t)u < nb ? xsf[2u * (k + u) + ch] : 0.0f;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if ((uint32_t)u >= nb || parked) continue;
                    const float yn = yy + gam * (xv[u] - yy);
                    const bool still = __float_as_uint(yn) == __float_as_uint(yy);
                    yy = yn;
                    if (lane < 4u) ys[(k + u) * 4u + c] = yy;
                    if (!flags || !__all((lane >= 4u || still) ? 1 : 0)) continue;
                    // All four chains sit on a fixed point of this frame's input.  It stays one while the input
                    // is (A) bit-identical, or (B) any-signed zero with every moving chain non-zero (x - y is
                    // then the same for +0 and -0).  The state is parked for the rest of this segment and for
                    // every following segment of the same class (up to the next event's segment).
                    const bool zero_ok = (flags & 2u) && __all((lane >= 4u || gam == 0.0f || yy != 0.0f) ? 1 : 0);
                    const bool const_ok = (flags & 1u) != 0u;
                    if (!zero_ok && !const_ok) continue;
                    uint32_t slot = 0;
                    if (lane == 0u) slot = atomicAdd(&d.stats[3], 1u);
                    slot = (uint32_t)__shfl((int)slot, 0, 64);
                    if (slot >= d.nseg) {   // job table full (only after many superseded rounds): keep stepping
                        if (lane == 0u) atomicSub(&d.stats[3], 1u);
                        continue;
                    }
                    uint32_t e = seg + 1u;
                    for (;;) {   // extend over following segments, 64 at a time
                        const uint32_t s2 = e + lane;
                        bool ok = false;
                        if (s2 < hi && !flagged(s2)) {
                            const uint32_t f2 = seg_flags[s2];
                            if (zero_ok) ok = (f2 & 2u) != 0u;
                            else { const uint2 x2 = X0[s2]; ok = (f2 & 1u) && x2.x == x0.x && x2.y == x0.y; }
                        }
                        const unsigned long long m = __ballot(ok ? 1 : 0);
                        const uint32_t run = m == ~0ull ? 64u : (uint32_t)__ffsll((long long)~m) - 1u;
                        e += run;
                        if (run < 64u) break;
                    }
                    const float y0 = quad_bcast<0>(yy), y1 = quad_bcast<1>(yy), y2 = quad_bcast<2>(yy), y3 = quad_bcast<3>(yy);
                    if (lane == 0u) {
                        BandJob j;
                        j.begin = start + n + k + (uint32_t)u + 1u;
                        j.end = min(e * S, M);
                        j.y[0] = y0; j.y[1] = y1; j.y[2] = y2; j.y[3] = y3;
                        j.pad[0] = j.pad[1] = 0u;
                        d.jobs[slot] = j;
                    }
                    job = slot;
                    parked_segs += e - seg;
                    last = e - 1u;
                    parked = true;
                    k += (uint32_t)u + 1u;
                }
                if (!parked) {
                    k += nb;
                    const float twin = __shfl_xor(yy, 4, 64);
                    coalesced = !had_job && __all((lane >= 8u || __float_as_uint(twin) == __float_as_uint(yy)) ? 1 : 0) != 0;
                }
            }
            // outputs of the k frames stepped in this batch
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            if (lane < k) {
                const float2 x = xs[lane];
                const float4 s = reinterpret_cast<const float4*>(ys)[lane];
                out_p[start + n + lane] = epilogue(band_out(kf, x.x, x.y, s.x, s.y, s.z, s.w), pg);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            n += k;
        }
        y = __shfl(yy, (int)c, 64);   // true state, every lane
        if (coalesced && !parked) {   // the stored exit state (and output) of this segment stands: cascade over
            ++recomputed;
            break;
        }
        recomputed += last - seg + 1u;
        const uint32_t fu_cmp = last == seg ? fu_old : Fu[last * 4u + c];
        const bool changed = __any((lane < 4u && fu_cmp != __float_as_uint(y)) ? 1 : 0) != 0;
        if (parked) {
            // the parked state is the exit state of every covered segment and the entry state of all but the first
            const uint4 yb = make_uint4(__float_as_uint(quad_bcast<0>(y)), __float_as_uint(quad_bcast<1>(y)),
                                        __float_as_uint(quad_bcast<2>(y)), __float_as_uint(quad_bcast<3>(y)));
            for (uint32_t s2 = seg + lane; s2 <= last; s2 += 64u) {
                d.seg_job[s2] = job;
                F4[s2] = yb;
                if (s2 > seg) S4[s2] = yb;
            }
        } else {
            if (lane < 4u) Fu[seg * 4u + c] = __float_as_uint(y);
            if (lane == 0u) d.seg_job[seg] = kNoJob;
        }
        seg = last + 1u;
        if (!changed || seg >= hi) break;   // (a change that reaches the end of the range is the next pass's event)
    }
}

// Rounds over the segments [lo, hi) (whole workgroup; hi - lo <= kFixMaxSegs).
struct BandFixLds {
    uint32_t bitmap[kFixMaxSegs / 32];
    uint32_t events[kFixMaxEvents];
    uint32_t wave_cnt[kFixWaves];
    uint32_t n_events;
    float2 xs[kFixWaves][64];
    __attribute__((aligned(16))) float ys[kFixWaves][64 * 4];
};
// Returns whether anything had to be repaired (same value in every thread).
TD_DEV bool band_fix_range(const BandSpecDesc& d, uint32_t M, uint32_t lo, uint32_t hi, BandFixLds& L) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t nwords = (hi - lo + 31u) / 32u;
    const uint4* S4 = reinterpret_cast<const uint4*>(d.seg_start);
    const uint4* F4 = reinterpret_cast<const uint4*>(d.seg_final);
    uint32_t cascades = 0, rec = 0, park = 0;
    bool any = false;
    for (;;) {
        if (tid == 0) L.n_events = 0;
        // ---- Phase A: mismatch bitmap (each wave owns two whole words per pass; four passes' loads in flight)
        bool saw = false;
        for (uint32_t base = lo; base < hi; base += 4u * kFixThreads) {
            uint4 a[4], p[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t s = base + (uint32_t)u * kFixThreads + tid;
                if (s > 0u && s < hi) { a[u] = S4[s]; p[u] = F4[s - 1u]; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t s = base + (uint32_t)u * kFixThreads + tid;
                const bool mis = s > 0u && s < hi && (a[u].x != p[u].x || a[u].y != p[u].y || a[u].z != p[u].z || a[u].w != p[u].w);
                const unsigned long long m = __ballot(mis ? 1 : 0);
                saw = saw || m != 0ull;
                if (lane == 0u && s < hi) {
                    L.bitmap[(s - lo) >> 5] = (uint32_t)m;
                    L.bitmap[((s - lo) >> 5) + 1u] = (uint32_t)(m >> 32);
                }
            }
        }
        // the all-clear -- nearly every call of nearly every render -- leaves here, without the list's scans and barriers
        if (!__syncthreads_or(saw ? 1 : 0)) break;
        // ---- ... and the ordered event list: one bitmap word per lane, block-wide exclusive scan of the counts
        for (uint32_t wbase = 0; wbase < nwords; wbase += kFixThreads) {
            const uint32_t wi = wbase + tid;
            uint32_t word = wi < nwords ? L.bitmap[wi] : 0u;
            const uint32_t cnt = (uint32_t)__popc(word);
            uint32_t inc = cnt;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t v = (uint32_t)__shfl_up((int)inc, off, 64);
                if (lane >= (uint32_t)off) inc += v;
            }
            if (lane == 63u) L.wave_cnt[wave] = inc;
            __syncthreads();
            uint32_t slot = L.n_events + inc - cnt;
            for (uint32_t w = 0; w < wave; ++w) slot += L.wave_cnt[w];
            while (word) {
                const uint32_t b = (uint32_t)__ffs((int)word) - 1u;
                word &= word - 1u;
                if (slot < kFixMaxEvents) L.events[slot] = lo + wi * 32u + b;
                ++slot;
            }
            __syncthreads();
            if (tid == 0) {
                uint32_t t = 0;
                for (int w = 0; w < kFixWaves; ++w) t += L.wave_cnt[w];
                L.n_events += t;
            }
            __syncthreads();
        }
        const uint32_t nev_total = L.n_events;
        if (nev_total == 0u) break;
        any = true;
        // ---- Phase B: independent cascades
        const uint32_t nev = min(nev_total, kFixMaxEvents);
        for (uint32_t i = wave; i < nev; i += kFixWaves) {
            band_fix_cascade(d, M, L.events[i], lo, hi, L.bitmap, L.xs[wave], L.ys[wave], rec, park);
            ++cascades;
        }
        // the next Phase A reads what the other waves of THIS workgroup stored: same CU, same L1 -- a
        // workgroup-scope fence (an agent-scope one would write the whole dirty L2 back every round)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __syncthreads();
    }
    if (lane == 0u && cascades) { atomicAdd(&d.stats[0], cascades); atomicAdd(&d.stats[1], rec); atomicAdd(&d.stats[2], park); }
    return any;
}

// Output of the parked stretches recorded by k_band_fix: the four states are constants, every frame's output follows from
// its own input frame -- fully parallel.  seg_job maps a segment to the job covering it.  Tiles are claimed from a counter
// (stats[33]) by whoever takes part.
TD_DEV void band_fill_tiles(const BandSpecDesc& d, uint32_t M, uint32_t* claim_s) {
    const BandCoef kf = band_coef(d.lgamma, d.hgamma, d.pass);
    constexpr uint32_t kPer = 8u;            // frames per thread and claim
    const uint32_t span = kPer * blockDim.x;
    const uint32_t n_claims = (uint32_t)(((uint64_t)M + span - 1u) / span);
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0u) *claim_s = atomicAdd(&d.stats[33], 1u);
        __syncthreads();
        const uint32_t t = *claim_s;
        if (t >= n_claims) return;
#pragma unroll
        for (uint32_t q = 0; q < kPer; ++q) {
            const uint64_t m64 = (uint64_t)t * span + q * blockDim.x + threadIdx.x;   // (64-bit: M may sit near 2^32)
            if (m64 >= M) break;
            const uint32_t m = (uint32_t)m64;
            const uint32_t j = d.seg_job[m / d.S];
            if (j == kNoJob) continue;
            const BandJob jb = d.jobs[j];
            if (m < jb.begin || m >= jb.end) continue;
            const float2 x = q4_frame(reinterpret_cast<const float*>(d.xq4), m);
            d.out[m] = epilogue(band_out(kf, x.x, x.y, jb.y[0], jb.y[1], jb.y[2], jb.y[3]), d.pg);
        }
    }
}
constexpr uint32_t kFillHelperPolls = 4096;   // a helper that has not seen the verdict by then (~ms) leaves: the others fill

// grid (G + H, vertices).  Workgroup g < G first settles its own slice of the segments (cascades stop at the slice
// end, the slice's first event trusts whatever exit state its left neighbour shows at the time), then the
// workgroup that finishes LAST (ticket in stats[4]; no spinning) checks the slice borders and, if one fails,
// repeats the rounds over all segments; the slices' all-clears plus the border check (or that global all-clear)
// are the proof for the whole vertex.  stats[] is zeroed by k_band_spec.
// That last workgroup then gives the verdict in stats[32] (a cache line of its own: the cascades' counters are busy): 1 = no parked stretch (nearly always), 2 = parked stretches
// recorded -- their output is filled in right here instead of by a launch of its own (84 launches of config 4's 341):
// by that workgroup and by the H helper workgroups (blockIdx.x >= G), which have done nothing but poll that word since they
// started.  Tiles are claimed from a counter, so the fill is complete when the last workgroup's loop ends whether or
// not a helper ever shows up (no assumption on dispatch order or residency); a helper that never sees a verdict leaves.
__global__ __launch_bounds__(kFixThreads) void k_band_fix(const BandSpecDesc* __restrict__ descs, uint32_t M, uint32_t G) {
    const BandSpecDesc& d = descs[blockIdx.y];
    __shared__ BandFixLds L;
    __shared__ uint32_t ticket_s;
    const uint32_t tid = threadIdx.x;
    if (blockIdx.x >= G) {   // helper
        if (tid == 0u) {
            uint32_t v = 0u;
            for (uint32_t i = 0; i < kFillHelperPolls; ++i) {
                v = __hip_atomic_load(&d.stats[32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v) break;
                __builtin_amdgcn_s_sleep(64);
            }
            ticket_s = v;
        }
        __syncthreads();
        if (ticket_s != 2u) return;
        if (tid == 0u) __threadfence();   // acquire the jobs: ONE lane's fence (it empties this CU's L1), the barrier hands it on
        __syncthreads();
        band_fill_tiles(d, M, &ticket_s);
        return;
    }
    const uint32_t per = (((d.nseg + G - 1u) / G) + 63u) & ~63u;
    const uint32_t lo = min(blockIdx.x * per, d.nseg), hi = min(lo + per, d.nseg);
    if (G > 1u) {
        const bool any = lo < hi && band_fix_range(d, M, lo, hi, L);
        if (any) __threadfence();   // release what this slice's repairs stored (and the jobs they parked)
        __syncthreads();
        if (tid == 0) ticket_s = atomicAdd(&d.stats[4], any ? 0x10001u : 1u);
        __syncthreads();
        if ((ticket_s & 0xFFFFu) != G - 1u) return;
        // Last workgroup.  If no slice had anything to repair, the slices' own all-clear checks -- together
        // they compared every segment with its predecessor, and nothing was modified -- are the proof already.
        if ((ticket_s >> 16) != 0u || any) {
            __threadfence();   // acquire the other slices' repairs
            // Inside a slice nothing changed after its own all-clear (no other slice writes there), so what is
            // left to prove is the 15 slice borders: each slice's first segment against its left neighbour's exit
            // state as it stands NOW.  Only if one of them fails do the rounds run again, over all segments.
            if (tid == 0) ticket_s = 0u;
            __syncthreads();
            if (tid > 0u && tid < G) {
                const uint32_t s = tid * per;
                if (s < d.nseg) {
                    const uint4 a = reinterpret_cast<const uint4*>(d.seg_start)[s], p = reinterpret_cast<const uint4*>(d.seg_final)[s - 1u];
                    if (a.x != p.x || a.y != p.y || a.z != p.z || a.w != p.w) atomicOr(&ticket_s, 1u);
                }
            }
            __syncthreads();
            if (ticket_s != 0u) band_fix_range(d, M, 0u, d.nseg, L);
        }
    } else {
        band_fix_range(d, M, 0u, d.nseg, L);
    }
    __syncthreads();
    if (tid < 4u) reinterpret_cast<uint32_t*>(d.state)[tid] = reinterpret_cast<const uint32_t*>(d.seg_final)[(d.nseg - 1u) * 4u + tid];
    if (tid == 0u) d.state->first = 0u;
    // ---- the verdict, and the parked stretches' output
    if (tid == 0u) ticket_s = __hip_atomic_load(&d.stats[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const bool jobs = ticket_s != 0u;
    if (jobs && tid == 0u) __threadfence();   // (release this workgroup's own jobs -- stored before the barrier above --, acquire everybody else's)
    __syncthreads();
    if (tid == 0u) __hip_atomic_store(&d.stats[32], jobs ? 2u : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (jobs) band_fill_tiles(d, M, &ticket_s);
}

// ------------------------------------------------------------------------------------------------
// k_band_scan: band_pass_gen as a blocked affine scan -- tolerance class; one launch per vertex or per chain of
// band-pass vertices (BandScanDesc, kernels.h)
// ------------------------------------------------------------------------------------------------
// (granule_store / granule_load: the inter-workgroup hand-off words, defined with the helpers at the top of the file)
TD_DEV double dsel4(const double v[4], uint32_t c) { return c == 0u ? v[0] : c == 1u ? v[1] : c == 2u ? v[2] : v[3]; }
constexpr uint32_t kScanSpinLimit = 4096;   // polls (~1 us each) before a predecessor is recomputed instead of awaited

template <int TMODE, int NF>
__global__ __launch_bounds__(kThreads, NF == 16 ? 3 : 4) void k_band_scan(const BandScanDesc* __restrict__ descs, uint32_t M) {
    constexpr int NP = NF / 2;                          // frame pairs (16-byte words) per lane
    constexpr uint32_t TILE = (uint32_t)NF * kThreads;  // frames per workgroup
    const BandScanDesc& d = descs[blockIdx.y];
    if (blockIdx.x >= d.n_tiles) return;
    // tile staging, lane-major with one pad word per lane run: the coalesced side (word q * 256 + tid) and the lane side
    // (words tid * NP .. + NP - 1) are both conflict-free
    __shared__ float4 xt[kThreads * (NP + 1)];
    __shared__ double wtot[kThreads / 64][4];
    __shared__ double carry_s[4];
    __shared__ float st_l[kScanMaxStages][5];       // tile 0: every stage's carried state {y[4], first} as the launch found it
    __shared__ float yinit_s[4];
    __shared__ uint32_t pb[kScanMaxK * 8];          // the predecessors' granule values, oldest first
    __shared__ uint32_t have_s[kScanMaxK / 32];     // bit p: predecessor p's eight granules are in pb
    __shared__ uint32_t all_s, tile_s;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t n_stages = d.n_stages;
    const bool chain = n_stages > 1u;
    const uint32_t dflags = (uint32_t)__builtin_amdgcn_readfirstlane((int)d.flags);   // (read once: see k_band_chain)
    const BandStageDesc TD_CONST* const stages = (const BandStageDesc TD_CONST*)(const TD_CONST char*)d.stages;   // (uniform: scalar loads)
    // Tile number: a ticket.  A workgroup only ever waits for LOWER tiles, and whoever drew a lower ticket is running -- no
    // assumption about the order workgroups are dispatched in, nor about how many of them the device holds at once.
    if (tid == 0u) tile_s = atomicAdd(d.ticket, 1u);
    __syncthreads();
    const uint32_t tile = tile_s;
    if (tile == 0u) {
        // The carried states, before any workgroup can have replaced them: the last tile stores a stage's new state only
        // after it has seen the word set below.
        for (uint32_t i = tid; i < n_stages * 5u; i += kThreads) {
            const uint32_t s = i / 5u, e = i - 5u * s;
            const uint32_t TD_GLOBAL* sw = reinterpret_cast<const uint32_t TD_GLOBAL*>((const TD_GLOBAL char*)stages[s].state);
            st_l[s][e] = __uint_as_float((e == 4u && stages[s].first_override) ? 1u : sw[e]);   // (e == 4: `first`; set_time since the vertex last ran)
        }
        __syncthreads();
        if (tid == 0u) __hip_atomic_store((gu32)(TD_GLOBAL char*)(d.ticket + 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const TermTab ins = term_tab(d.ins);
    const uint32_t k = d.k;
    auto slot = [](uint32_t p) { return p + p / (uint32_t)NP; };
    const uint32_t tile0 = tile * TILE, mlast = M - 1u, mf = tile0 + (uint32_t)NF * tid;

    float4 x[NP];             // the lane's NF consecutive frames: a stage's input, then its output
    float nz_e = 0.0f;        // (band_mode 2) the lane's share of the launch's estimated deviation energy (BandScanDesc::noise)
    double excl[4], xw[4], B[4];
    // ---- the first vertex' input terms for tile tt -> lane-consecutive frames
    auto load_tile = [&](uint32_t tt) {
        const uint32_t t0 = tt * TILE;
#pragma unroll
        for (int r = 0; r < NP / 2; ++r) {
            const uint32_t m0 = t0 + (uint32_t)(2 * r) * 512u + 2u * tid, m1 = m0 + 512u;
            float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
            sum_terms<TMODE>(ins, k, m0, m1, M, a0, a1);   // sum_inputs (extensions.rs:310-319): zero, += in edge order
            xt[slot((uint32_t)(2 * r) * 256u + tid)] = a0;
            xt[slot((uint32_t)(2 * r + 1) * 256u + tid)] = a1;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NP; ++j) x[j] = xt[tid * (uint32_t)(NP + 1) + (uint32_t)j];
    };
    // ---- stage parameters (uniform)
    const BandStageDesc TD_CONST* sp = stages;
    float lgam = 0.f, hgam = 0.f;
    double al = 0.0, ah = 0.0, gl = 0.0, gh = 0.0, awl = 0.0, awh = 0.0;
    bool need_r = true;
    auto set_stage = [&](uint32_t s) {
        sp = stages + s;
        lgam = sp->lgamma; hgam = sp->hgamma;
        al = 1.0 - (double)lgam; ah = 1.0 - (double)hgam; gl = (double)lgam; gh = (double)hgam;
        awl = sp->aw[0]; awh = sp->aw[1];
        // (the right-channel smoothers always run here: a `pass` vertex' right output is r - cutl, but the reference's
        // expression is cutr * 0 + (r - cutl) * 1 -- a right smoother gone non-finite does reach the output, as NaN)
        need_r = true;
    };
    // ---- zero-state responses of x: of the lane's run (b), of the wave up to the lane (excl), of the tile up to the
    // wave (xw), of the whole tile (B).  Used for the workgroup's own tile and, should a predecessor of a single vertex
    // fail to publish in time, for that predecessor's tile: identical arithmetic, identical values.
    // need_r: the right-channel smoothers (chains 1, 3) run (always, in this kernel).
    auto pass1 = [&](uint32_t s, bool is_tile0) {
        double b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
        if (need_r) {
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const float4 v = x[j];
                b0 = __builtin_fma(b0, al, gl * (double)v.x);
                b1 = __builtin_fma(b1, al, gl * (double)v.y);
                b2 = __builtin_fma(b2, ah, gh * (double)v.x);
                b3 = __builtin_fma(b3, ah, gh * (double)v.y);
                b0 = __builtin_fma(b0, al, gl * (double)v.z);
                b1 = __builtin_fma(b1, al, gl * (double)v.w);
                b2 = __builtin_fma(b2, ah, gh * (double)v.z);
                b3 = __builtin_fma(b3, ah, gh * (double)v.w);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const double xa = (double)x[j].x, xb = (double)x[j].z;
                b0 = __builtin_fma(b0, al, gl * xa);
                b2 = __builtin_fma(b2, ah, gh * xa);
                b0 = __builtin_fma(b0, al, gl * xb);
                b2 = __builtin_fma(b2, ah, gh * xb);
            }
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) {   // inclusive scan over the wave: b_i += a^(NF * 2^q) b_(i - 2^q)
            const uint32_t dd = 1u << q;
            const double pl = sp->ap[0][q], ph = sp->ap[1][q];
            const double t0 = __shfl_up(b0, dd, 64), t2 = __shfl_up(b2, dd, 64);
            if (lane >= dd) {
                b0 = __builtin_fma(t0, pl, b0);
                b2 = __builtin_fma(t2, ph, b2);
            }
            if (need_r) {
                const double t1 = __shfl_up(b1, dd, 64), t3 = __shfl_up(b3, dd, 64);
                if (lane >= dd) {
                    b1 = __builtin_fma(t1, pl, b1);
                    b3 = __builtin_fma(t3, ph, b3);
                }
            }
        }
        excl[0] = __shfl_up(b0, 1u, 64); excl[2] = __shfl_up(b2, 1u, 64);
        excl[1] = 0.0; excl[3] = 0.0;
        if (need_r) { excl[1] = __shfl_up(b1, 1u, 64); excl[3] = __shfl_up(b3, 1u, 64); }
        if (lane == 0u) excl[0] = excl[1] = excl[2] = excl[3] = 0.0;
        if (lane == 63u) { wtot[wave][0] = b0; wtot[wave][1] = b1; wtot[wave][2] = b2; wtot[wave][3] = b3; }
        if (is_tile0 && tid == 0u) {
            // state at the chunk's first frame: carried, or seeded from buf[0] (extensions.rs:664-670).  Only tile 0's own
            // workgroup ever computes tile 0 (it is the one predecessor that is never recomputed).
            const bool first = __float_as_uint(st_l[s][4]) != 0u;
            yinit_s[0] = first ? x[0].x : st_l[s][0];
            yinit_s[1] = first ? x[0].y : st_l[s][1];
            yinit_s[2] = first ? x[0].x : st_l[s][2];
            yinit_s[3] = first ? x[0].y : st_l[s][3];
            if (first) {   // a constant chain (gamma 0) keeps its seed for good
                float* sf = reinterpret_cast<float*>(sp->state);
                if (lgam == 0.0f) { sf[0] = yinit_s[0]; sf[1] = yinit_s[1]; }
                if (hgam == 0.0f) { sf[2] = yinit_s[2]; sf[3] = yinit_s[3]; }
            }
        }
        __syncthreads();
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (uint32_t w = 0; w < (uint32_t)(kThreads / 64); ++w) {
            if (w == wave) { xw[0] = acc[0]; xw[1] = acc[1]; xw[2] = acc[2]; xw[3] = acc[3]; }
            acc[0] = __builtin_fma(acc[0], awl, wtot[w][0]);
            acc[1] = __builtin_fma(acc[1], awl, wtot[w][1]);
            acc[2] = __builtin_fma(acc[2], awh, wtot[w][2]);
            acc[3] = __builtin_fma(acc[3], awh, wtot[w][3]);
        }
        B[0] = acc[0]; B[1] = acc[1]; B[2] = acc[2]; B[3] = acc[3];
    };
    auto half_of = [&](uint32_t q) -> uint32_t {   // granule q of a tile: chain q / 2, low / high word of its double
        const unsigned long long u = (unsigned long long)__double_as_longlong(dsel4(B, q >> 1));
        return (q & 1u) ? (uint32_t)(u >> 32) : (uint32_t)u;
    };
    bool state_may_be_written = tile == 0u;   // (this workgroup has seen "tile 0 has read the carried states")
    // the first stage at which this tile's response or entry state was not finite (uniform), per channel: the left smoothers
    // (chains 0, 2) and the right ones (chains 1, 3) -- the reference's cutl depends on the left pair only, cutr on the right
    // (extensions.rs:674-687), so a right-channel NaN must not reach the left output or the left states
    uint32_t poisoned_l = n_stages, poisoned_r = n_stages;

    const double* pw_cur = nullptr;
    double pwl = 1.0, pwh = 1.0;   // (1 - gamma)^(NF lane), low / high
    for (uint32_t s = 0; s < n_stages; ++s) {
        set_stage(s);
        if (sp->pw != pw_cur) {   // (uniform; the stages of a chain of identical filters share one table: loaded once)
            pw_cur = sp->pw;
            pwl = pw_cur[lane];
            pwh = pw_cur[64u + lane];
        }
        // the next link's envelope gains for the lane's frames: issued now, used after pass 2
        const float* env_pre = (s + 1u < n_stages && sp->n_post) ? (sp->post[0].env ? sp->post[0].env : (sp->n_post > 1u ? sp->post[1].env : nullptr)) : nullptr;
        float4 envv[NP / 2];
#pragma unroll
        for (int q = 0; q < NP / 2; ++q)
            envv[q] = (env_pre && mf + 4u * (uint32_t)q < M) ? gload4(env_pre + mf + 4u * (uint32_t)q) : make_float4(0.f, 0.f, 0.f, 0.f);
        unsigned long long* const sync = sp->sync;
        const uint32_t n_pred = min(tile, sp->K), first_pred = tile - n_pred;
        // Work list of the stage: the own tile; then -- a single vertex only, and only if a predecessor did not publish
        // in time -- the missing predecessors and the own tile once more (its frames were dropped to make room).
        uint32_t cur = tile, p_cur = 0u, p_next = 0u;
        int phase = 0;
        for (;;) {
            if (s == 0u) load_tile(cur);   // (later stages of a chain: the frames are in registers)
            pass1(s, cur == 0u);
            if (phase == 2) break;
            if (phase == 0) {
                if (tid < 8u) {   // publish: the state this tile leaves behind when entered with zero state (tile 0: with the true state)
                    const uint32_t c = tid >> 1;
                    if (tile == 0u) {
                        const double v = __builtin_fma((double)yinit_s[c], sp->at[c >> 1], dsel4(B, c));
                        const unsigned long long u = (unsigned long long)__double_as_longlong(v);
                        granule_store(sync + tid, (tid & 1u) ? (uint32_t)(u >> 32) : (uint32_t)u);
                    } else {
                        granule_store(sync + (size_t)tile * 8u + tid, half_of(tid));
                    }
                }
                if (n_pred == 0u) break;
                if (dflags & 2u) {   // (timing experiments only: no look-back -- wrong results)
                    for (uint32_t i = tid; i < n_pred * 8u; i += kThreads) pb[i] = 0u;
                    break;
                }
                if (wave == 0u) {
                    const unsigned long long* g0 = sync + (size_t)first_pred * 8u;
                    const uint32_t n8 = n_pred * 8u;
                    if (lane < kScanMaxK / 32u) have_s[lane] = 0u;
                    const bool forced = !chain && (dflags & 1u) != 0u;
                    bool done = false;
                    if (!forced) {
                        for (uint32_t spin = 0;; ++spin) {
                            bool ok = true;
                            for (uint32_t idx = lane; idx < n8; idx += 64u) {
                                const unsigned long long g = granule_load(g0 + idx);
                                const bool t = (uint32_t)(g >> 32) == 1u;
                                if (t) pb[idx] = (uint32_t)g;
                                ok = ok && t;
                            }
                            if (__all(ok ? 1 : 0)) { done = true; break; }
                            if (!chain && spin >= kScanSpinLimit) break;   // (a chain's waits are safe without bound: tickets)
                            __builtin_amdgcn_s_sleep(2);
                        }
                    }
                    if (!done) {
                        // which predecessors are complete (lanes 8 j .. 8 j + 7 read tile j of a group of eight)
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                        if (!forced)
                            for (uint32_t base = 0; base < n8; base += 64u) {
                                const uint32_t idx = base + lane;
                                bool t = false;
                                if (idx < n8) {
                                    const unsigned long long g = granule_load(g0 + idx);
                                    t = (uint32_t)(g >> 32) == 1u;
                                    if (t) pb[idx] = (uint32_t)g;
                                }
                                const unsigned long long bal = __ballot(t ? 1 : 0);
                                const uint32_t p = base / 8u + lane;
                                if (lane < 8u && p < n_pred && ((bal >> (8u * lane)) & 0xFFull) == 0xFFull)
                                    atomicOr(&have_s[p >> 5], 1u << (p & 31u));
                            }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                        // tile 0 depends on nobody and is the one tile nobody else can compute (it folds the carried
                        // state): waited for without bound
                        if (first_pred == 0u && !(have_s[0] & 1u)) {
                            for (;;) {
                                bool t = true;
                                if (lane < 8u) {
                                    const unsigned long long g = granule_load(g0 + lane);
                                    t = (uint32_t)(g >> 32) == 1u;
                                    if (t) pb[lane] = (uint32_t)g;
                                }
                                if (__all(t ? 1 : 0)) break;
                                __builtin_amdgcn_s_sleep(8);
                            }
                            if (lane == 0u) atomicOr(&have_s[0], 1u);
                        }
                    }
                    if (lane == 0u) all_s = done ? 1u : 0u;
                }
                __syncthreads();
                if (all_s) break;
                phase = 1;
            } else {   // phase 1: a predecessor's response, computed here
                if (tid < 8u) pb[p_cur * 8u + tid] = half_of(tid);
            }
            // next missing predecessor (uniform), or back to the own tile
            while (p_next < n_pred && ((have_s[p_next >> 5] >> (p_next & 31u)) & 1u)) ++p_next;
            if (p_next < n_pred) { p_cur = p_next++; cur = first_pred + p_cur; }
            else { phase = 2; cur = tile; }
        }
        __syncthreads();   // pb complete
        if (tid < 4u) {
            // state entering the tile: C = sum_j a_tile^(j-1) B_(tile-j), oldest first (tile 0: the carried / seeded state)
            const double a = sp->at[tid >> 1];
            double C = 0.0;
            if (tile == 0u) C = (double)yinit_s[tid];
            else
                for (uint32_t p0 = 0; p0 < n_pred; p0 += 16u) {   // (the LDS reads of a group in flight together, then the dependent chain)
                    double bv[16];
#pragma unroll
                    for (uint32_t e = 0; e < 16u; ++e) {
                        const uint32_t p = min(p0 + e, n_pred - 1u);
                        const unsigned long long u = (unsigned long long)pb[p * 8u + 2u * tid] | ((unsigned long long)pb[p * 8u + 2u * tid + 1u] << 32);
                        bv[e] = __longlong_as_double((long long)u);
                    }
#pragma unroll
                    for (uint32_t e = 0; e < 16u; ++e)
                        if (p0 + e < n_pred) C = __builtin_fma(C, a, bv[e]);
                }
            carry_s[tid] = C;
        }
        __syncthreads();
        // (x - x == 0 only for finite x: the tile's own response and the state entering it -- once either is not finite the
        // reference's state stays NaN for the rest of the chunk, see the end of the kernel)
        if (poisoned_l == n_stages &&
            (!(B[0] - B[0] == 0.0) || !(B[2] - B[2] == 0.0) || !(carry_s[0] - carry_s[0] == 0.0) || !(carry_s[2] - carry_s[2] == 0.0)))
            poisoned_l = s;
        if (poisoned_r == n_stages &&
            (!(B[1] - B[1] == 0.0) || !(B[3] - B[3] == 0.0) || !(carry_s[1] - carry_s[1] == 0.0) || !(carry_s[3] - carry_s[3] == 0.0)))
            poisoned_r = s;
        // entry state of the lane's run, exact arithmetic rounded once: excl + a^(NF lane) (xw + a_wave^wave C)
        double awpl = 1.0, awph = 1.0;
        for (uint32_t w = 0; w < wave; ++w) { awpl *= awl; awph *= awh; }
        float y0 = (float)__builtin_fma(pwl, __builtin_fma(awpl, carry_s[0], xw[0]), excl[0]);
        float y1 = (float)__builtin_fma(pwl, __builtin_fma(awpl, carry_s[1], xw[1]), excl[1]);
        float y2 = (float)__builtin_fma(pwh, __builtin_fma(awph, carry_s[2], xw[2]), excl[2]);
        float y3 = (float)__builtin_fma(pwh, __builtin_fma(awph, carry_s[3], xw[3]), excl[3]);
        if (d.noise) {   // (uniform; band_mode 2, single vertices: the estimate of k_band_chain<.., true>, both channels' smoothers)
            const float x0l = x[0].x, x0r = x[0].y;
            const float a0 = fabsf(y0), a1 = fabsf(y1), a2 = fabsf(y2), a3 = fabsf(y3);
            const float vl = __builtin_fmaf(sp->nzv[0], y0 * y0, sp->nzv[1] * (y2 * y2));
            const float vr = __builtin_fmaf(sp->nzv[0], y1 * y1, sp->nzv[1] * (y3 * y3));
            const float ol = (fabsf(x0l - y0) < sp->nzk[0] * a0 ? sp->nzs[0] * a0 : 0.0f) + (fabsf(x0l - y2) < sp->nzk[1] * a2 ? sp->nzs[1] * a2 : 0.0f);
            const float orr = (fabsf(x0r - y1) < sp->nzk[0] * a1 ? sp->nzs[0] * a1 : 0.0f) + (fabsf(x0r - y3) < sp->nzk[1] * a3 ? sp->nzs[1] * a3 : 0.0f);
            const float el = __builtin_fmaf(ol, ol, vl), er = __builtin_fmaf(orr, orr, vr);
            // a `pass` vertex' two outputs both carry the LEFT cut (extensions.rs:685); a `cut` vertex' right output its own
            nz_e += mf < M ? (float)NF * (sp->pass ? el : 0.5f * (el + er)) : 0.0f;
        }
        // the lane's frames in the reference's own arithmetic (extensions.rs:671-688); x becomes the vertex' output.
        // Pan / gain steps a vertex skips (flags) are skipped by uniform branches, not computed and masked.
        const BandCoef kf = band_coef(lgam, hgam, sp->pass);
        PanGain pg;
        pg.l_amp = sp->pg.l_amp; pg.r_amp = sp->pg.r_amp; pg.gain = sp->pg.gain; pg.flags = sp->pg.flags;
        auto epi4 = [](float4 v, const PanGain& g4) {
            if (g4.flags & 1u) { v.x *= g4.l_amp; v.y *= g4.r_amp; v.z *= g4.l_amp; v.w *= g4.r_amp; }
            if (g4.flags & 2u) { v.x *= g4.gain; v.y *= g4.gain; v.z *= g4.gain; v.w *= g4.gain; }
            return v;
        };
        float f0 = 0.f, f1 = 0.f, f2 = 0.f, f3 = 0.f;
        bool has_fin = false;
        const bool fin_here = mlast - tile0 < TILE;   // (uniform: the chunk's last frame lies in this tile)
        if (need_r) {
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const float4 v = x[j];
                y0 = y0 + lgam * (v.x - y0); y1 = y1 + lgam * (v.y - y1);
                y2 = y2 + hgam * (v.x - y2); y3 = y3 + hgam * (v.y - y3);
                const float2 oa = band_out(kf, v.x, v.y, y0, y1, y2, y3);
                if (fin_here && mf + 2u * (uint32_t)j == mlast) { f0 = y0; f1 = y1; f2 = y2; f3 = y3; has_fin = true; }
                y0 = y0 + lgam * (v.z - y0); y1 = y1 + lgam * (v.w - y1);
                y2 = y2 + hgam * (v.z - y2); y3 = y3 + hgam * (v.w - y3);
                const float2 ob = band_out(kf, v.z, v.w, y0, y1, y2, y3);
                if (fin_here && mf + 2u * (uint32_t)j + 1u == mlast) { f0 = y0; f1 = y1; f2 = y2; f3 = y3; has_fin = true; }
                x[j] = epi4(make_float4(oa.x, oa.y, ob.x, ob.y), pg);
            }
        } else {
            // pass: cut_mul = 0, pass_mul = 1 -> out = (l - cutl, r - cutl) with cutl = (lmul ll + hmul (l - hl)) 0.5; for
            // finite values the same bits as the reference's `cutl * cut_mul + passl * pass_mul` up to the sign of a zero
            const bool lo_on = lgam != 0.0f, hi_on = hgam != 0.0f;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const float4 v = x[j];
                y0 = y0 + lgam * (v.x - y0);
                y2 = y2 + hgam * (v.x - y2);
                const float ca = ((lo_on ? y0 : 0.0f) + (hi_on ? v.x - y2 : 0.0f)) * 0.5f;
                if (fin_here && mf + 2u * (uint32_t)j == mlast) { f0 = y0; f2 = y2; has_fin = true; }
                y0 = y0 + lgam * (v.z - y0);
                y2 = y2 + hgam * (v.z - y2);
                const float cb = ((lo_on ? y0 : 0.0f) + (hi_on ? v.z - y2 : 0.0f)) * 0.5f;
                if (fin_here && mf + 2u * (uint32_t)j + 1u == mlast) { f0 = y0; f2 = y2; has_fin = true; }
                x[j] = epi4(make_float4(v.x - ca, v.y - ca, v.z - cb, v.w - cb), pg);
            }
        }
        if (has_fin) {   // the lane that holds the chunk's last frame carries the state over (constant chains: pass1)
            if (!state_may_be_written) {
                while (__hip_atomic_load((gu32)(TD_GLOBAL char*)(d.ticket + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u)
                    __builtin_amdgcn_s_sleep(8);
                state_may_be_written = true;
            }
            float* sf = reinterpret_cast<float*>(sp->state);
            if (lgam != 0.0f) { sf[0] = f0; if (need_r) sf[1] = f1; }
            if (hgam != 0.0f) { sf[2] = f2; if (need_r) sf[3] = f3; }
            sp->state->first = 0u;
        }
        if (s + 1u < n_stages) {
            // The links to the next band-pass vertex: an Adsr vertex multiplies by its gain of the frame (k_adsr_env), then
            // pan / gain; a single-input Sum is pan / gain only.  Every link and the next vertex start with their own
            // sum_inputs `0.0 + x`, whose only effect is -0 -> +0; a zero stays a zero through the multiplies in between, so
            // ONE `0.0 + x` at the end leaves the same bits.  Frames at or beyond M become 0, as sum_terms leaves them.
            const uint32_t np = sp->n_post;
            for (uint32_t p = 0; p < np; ++p) {
                const float* env = sp->post[p].env;
                PanGain lp;
                lp.l_amp = sp->post[p].pg.l_amp; lp.r_amp = sp->post[p].pg.r_amp; lp.gain = sp->post[p].pg.gain; lp.flags = sp->post[p].pg.flags;
                if (env) {
#pragma unroll
                    for (int q = 0; q < NP / 2; ++q) {
                        // (mf is a multiple of NF: 16-byte aligned; the buffer holds at least frames + 3 gains)
                        const float4 e = env == env_pre ? envv[q]
                                       : mf + 4u * (uint32_t)q < M ? gload4(env + mf + 4u * (uint32_t)q) : make_float4(0.f, 0.f, 0.f, 0.f);
                        const float4 a = x[2 * q], b = x[2 * q + 1];
                        x[2 * q] = make_float4(a.x * e.x, a.y * e.x, a.z * e.y, a.w * e.y);
                        x[2 * q + 1] = make_float4(b.x * e.z, b.y * e.z, b.z * e.w, b.w * e.w);
                    }
                }
                if (lp.flags) {
#pragma unroll
                    for (int j = 0; j < NP; ++j) x[j] = epi4(x[j], lp);
                }
            }
#pragma unroll
            for (int j = 0; j < NP; ++j) x[j] = add4(make_float4(0.f, 0.f, 0.f, 0.f), x[j]);
            if (tile0 + TILE > M) {   // (uniform: only the chunk's last tile has frames beyond M)
#pragma unroll
                for (int j = 0; j < NP; ++j) x[j] = zero_tail(x[j], mf + 2u * (uint32_t)j, M);
            }
        }
    }
    if (d.noise) {   // (band_mode 2) the tile's share of the launch's estimate, for k_band_audit
        __shared__ float nzw[kThreads / 64];
        float e = (nz_e == nz_e) ? nz_e : 0.0f;   // (NaN: a tile the reference turns NaN as well)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) e += __shfl_xor(e, off, 64);
        if (lane == 0u) nzw[wave] = e;
        __syncthreads();
        if (tid == 0u) d.noise[tile] = (nzw[0] + nzw[1]) + (nzw[2] + nzw[3]);
    }
    if (d.poison) {
        // ---- once NaN, always NaN (the reference's smoother state never recovers; the look-back forgets a tile after K tiles):
        // every tile says at which stage it went non-finite and learns the same of ALL earlier tiles (lower tickets: their
        // holders are running and get here without waiting for anybody above them, so the wait needs no bound).
        __shared__ uint32_t pz[kThreads / 64][2];
        if (tid == 0u) granule_store(d.poison + tile, poisoned_l | (poisoned_r << 16));   // (n_stages <= 128)
        uint32_t pl = n_stages, pr = n_stages;
        (void)for_lower_granules(d.poison, tile, 0xFFFFFFFFu, [&pl, &pr](uint32_t, uint32_t v) { pl = min(pl, v & 0xFFFFu); pr = min(pr, v >> 16); });
        // (EARLIER tiles only: inside the tile that goes non-finite itself the arithmetic carries the NaN from the frame it
        // appears at -- the frames before it stay what they are -- and into the states it stores)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            pl = min(pl, (uint32_t)__shfl_xor((int)pl, off, 64));
            pr = min(pr, (uint32_t)__shfl_xor((int)pr, off, 64));
        }
        if (lane == 0u) { pz[wave][0] = pl; pz[wave][1] = pr; }
        __syncthreads();
        pl = min(min(pz[0][0], pz[1][0]), min(pz[2][0], pz[3][0]));
        pr = min(min(pz[0][1], pz[1][1]), min(pz[2][1], pz[3][1]));
        if (pl < n_stages || pr < n_stages) {   // (rare)
            const float qnan = __uint_as_float(0x7FC00000u);
            // out_l = cutl cut_mul + (l - cutl) pass_mul,  out_r = cutr cut_mul + (r - cutl) pass_mul  (extensions.rs:682-687, one
            // of the two factors 0, the other 1): a non-finite cutl turns BOTH outputs NaN whatever the vertex passes, a
            // non-finite cutr the right one only
            const bool left_bad = pl < n_stages, right_bad = left_bad || pr < n_stages;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                if (left_bad) { x[j].x = qnan; x[j].z = qnan; }
                if (right_bad) { x[j].y = qnan; x[j].w = qnan; }
            }
            if (mlast - tile0 < TILE && wave == 0u) {   // (the tile with the chunk's last frame: the states carried out of the chunk)
                for (uint32_t s = min(pl, pr) + lane; s < n_stages; s += 64u) {
                    float* sf = reinterpret_cast<float*>(stages[s].state);
                    const bool rr = stages[s].pass == 0u;
                    if (s >= pl) {
                        if (stages[s].lgamma != 0.0f) sf[0] = qnan;
                        if (stages[s].hgamma != 0.0f) sf[2] = qnan;
                    }
                    if (s >= pr && rr) {
                        if (stages[s].lgamma != 0.0f) sf[1] = qnan;
                        if (stages[s].hgamma != 0.0f) sf[3] = qnan;
                    }
                }
            }
        }
    }
    // the last vertex' output, back through the staging for coalesced stores
#pragma unroll
    for (int j = 0; j < NP; ++j) xt[tid * (uint32_t)(NP + 1) + (uint32_t)j] = x[j];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NP; ++q) store_pair(d.out, tile0 + 2u * ((uint32_t)q * 256u + tid), M, xt[slot((uint32_t)q * 256u + tid)]);
}

// ------------------------------------------------------------------------------------------------
// k_band_chain: a chain of `pass` band-pass vertices in ONE launch (BandScanDesc with n_stages >= 2)
// ------------------------------------------------------------------------------------------------
// The frames of a tile -- NF * 256 consecutive ones, NF per lane -- stay in registers from stage to stage; the links
// between two vertices (envelope gain, pan / gain) are applied in between.  Per stage:
//   z[n]   the smoothers' zero-state response inside the lane's run, the reference's own step  z + gamma (x - z)  (f32)
//   scan   of the runs' responses over the wave (double), the four waves' totals combined through LDS (barrier 1); the
//          tile's total published as four granules by wave 0
//   C      = sum_j a_tile^j B_(tile - 1 - j) over the K preceding tiles: lane j of wave 0 reads predecessor j's granules
//          (spinning until they are tagged), multiplies by its power, the wave adds up (double); to LDS (barrier 2)
//   y[n]   = z[n] + (1 - gamma)^(n + 1) c, c the lane's entry state (exact arithmetic, rounded once) -- no second
//          dependent walk -- then the vertex' output  (l - cut, r - cut), cut = (lmul low + hmul (l - high)) / 2.
// A `pass` vertex' right-channel smoothers reach no output (extensions.rs:685, quirk Q7) and are not run.
// (Tried and dropped, both bit-identical: every WAVE handing over for itself -- no barriers, but 2 816 pollers with a
// four times deeper look-back: 0.71 ms for BASELINE config 4's 84 stages against 0.60; a wave owning two wave-tiles half
// a timeline apart so that one's hand-off passes under the other's arithmetic -- 256 registers, two waves per SIMD: 1.22 ms.)
// GUARD (engine option "band_mode" 2): the launch also estimates how far its output lies from the reference's f32 trajectory
// -- the one thing this class gives up -- and leaves the estimate, per wave-tile, to k_band_audit (BandScanDesc::noise).  The
// reference's smoother rounds its state once per frame; against exact arithmetic those roundings accumulate to (a) a random
// walk held in check by the filter's own decay, variance ulp^2 / 12 per step over 1 / (gamma (2 - gamma)) steps, while the
// state moves by more than its own ulp per frame, and (b) where it does not -- a held level, silence with an offset, a
// very slow ramp: |gamma (x - y)| below a few ulp(y) -- a standing offset of up to ulp(y) / (2 gamma): the f32 state parks
// short of its target and the exact-arithmetic one does not.  Both are functions of the state's LEVEL, which every lane has
// in its entry state: one sample per 16 frames and smoother.  The estimate rides through the chain like the signal does
// (static pan / gain of vertices and links, the envelope links' gain of the lane's first frame) as a variance and an offset.
template <int TMODE, bool GUARD>
__global__ __launch_bounds__(kThreads, 3) void k_band_chain(const BandScanDesc* __restrict__ descs, uint32_t M, uint32_t chains_in_x) {
    constexpr int NF = 16, NP = NF / 2;
    constexpr uint32_t WT = (uint32_t)NF * 64u;         // frames per wave-tile
    // Several chains in one launch (a batch of projects): the chain index is the FAST grid dimension, so the workgroups
    // the device holds at a time belong to all of them -- chains are independent of each other and each one's tiles move in
    // lockstep from hop to hop; side by side on a CU, one's arithmetic runs under another's hop.
    const BandScanDesc& d = descs[chains_in_x ? blockIdx.x : blockIdx.y];
    if ((chains_in_x ? blockIdx.y : blockIdx.x) >= d.n_tiles) return;
    __shared__ float4 xt[kThreads * (NP + 1)];          // staging, one quarter per wave: lane-major, one pad word per lane run
    __shared__ float st_l[kScanMaxStages][5];           // tile 0: every stage's carried state {y[4], first} as the launch found it
    __shared__ double wtot[kThreads / 64][2];
    __shared__ double carry_s[2];
    __shared__ uint32_t tile_s;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));   // (wave: a scalar, its branches are branches)
    const uint32_t n_stages = d.n_stages;
    const BandStageDesc TD_CONST* const stages = (const BandStageDesc TD_CONST*)(const TD_CONST char*)d.stages;   // (uniform: scalar loads)
    if (tid == 0u) tile_s = atomicAdd(d.ticket, 1u);
    __syncthreads();
    const uint32_t tile = tile_s;
    __shared__ float nzx_s[16];   // (GUARD, BandScanDesc::nz_probe) the tile's sixteen sample energies
    if (GUARD && d.nz_probe) {    // (uniform) the tile's share of k_sine_probe's work: a sample every 256 frames, sixteen in the tile
        const float e = probe_energy(*d.nz_probe, tile * 16u + (tid >> 4), tid & 15u, M);
        if ((tid & 15u) == 0u) nzx_s[tid >> 4] = e;   // (read at the chain's end: barriers in between)
    }
    if (tile == 0u) {
        // The carried states, before anybody can have replaced them: the wave holding the chunk's last frame stores a
        // stage's new state only after it has seen the word set below.
        for (uint32_t i = tid; i < n_stages * 5u; i += kThreads) {
            const uint32_t s = i / 5u, e = i - 5u * s;
            const uint32_t TD_GLOBAL* sw = reinterpret_cast<const uint32_t TD_GLOBAL*>((const TD_GLOBAL char*)stages[s].state);
            st_l[s][e] = __uint_as_float((e == 4u && stages[s].first_override) ? 1u : sw[e]);   // (e == 4: `first`; set_time since the vertex last ran)
        }
        __syncthreads();
        if (tid == 0u) __hip_atomic_store((gu32)(TD_GLOBAL char*)(d.ticket + 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const uint32_t wt = tile * 4u + wave;               // this wave's tile
    const SumDesc TD_CONST* const nd = (const SumDesc TD_CONST*)(const TD_CONST char*)d.norm;   // (nullptr: no Normalize vertex behind the chain)
    // (its carried max, read before anything else: the tile holding the chunk's end replaces it once every tile has published)
    const float norm_init = nd ? (nd->use_init ? nd->init_max : gload1(&nd->state->max)) : 0.0f;
    const uint32_t wt0 = wt * WT, mlast = M - 1u, mf = wt0 + (uint32_t)NF * lane;
    float4* const xw4 = xt + wave * 64u * (uint32_t)(NP + 1);   // the wave's quarter of the staging
    auto slot = [](uint32_t p) { return p + p / (uint32_t)NP; };
    // ---- the first vertex' input terms for this wave's frames, coalesced (16-byte word q * 64 + lane), then lane-consecutive
    float4 x[NP];
    {
        const TermTab ins = term_tab(d.ins);
        const uint32_t k = d.k;
        PanGain pre;
        pre.l_amp = d.pre.l_amp; pre.r_amp = d.pre.r_amp; pre.gain = d.pre.gain; pre.flags = d.pre.flags;
        PanGain pre2;
        pre2.l_amp = d.pre2.l_amp; pre2.r_amp = d.pre2.r_amp; pre2.gain = d.pre2.gain; pre2.flags = d.pre2.flags;
        const bool pre_on = (pre.flags | pre2.flags) != 0u;
#pragma unroll
        for (int r = 0; r < NP / 2; ++r) {
            const uint32_t m0 = wt0 + (uint32_t)(2 * r) * 128u + 2u * lane, m1 = m0 + 128u;
            float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
            sum_terms<TMODE>(ins, k, m0, m1, M, a0, a1);   // sum_inputs (extensions.rs:310-319): zero, += in edge order
            if (pre_on) { a0 = epilogue4(epilogue4(a0, pre), pre2); a1 = epilogue4(epilogue4(a1, pre), pre2); }   // (the Sum vertex in front, the stage behind it: their pan / gain)
            xw4[slot((uint32_t)(2 * r) * 64u + lane)] = a0;
            xw4[slot((uint32_t)(2 * r + 1) * 64u + lane)] = a1;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // (the wave's own LDS words: ordered within the wave)
#pragma unroll
        for (int j = 0; j < NP; ++j) x[j] = xw4[lane * (uint32_t)(NP + 1) + (uint32_t)j];
    }
    bool state_may_be_written = tile == 0u;
    // ---- the right channel's only way into a `pass` vertex' output other than r - cutl: cutr * 0 (extensions.rs:686-687), NaN
    // once a right smoother is not finite -- which it is from the first non-finite right INPUT frame on, for good, in every
    // stage from there (the NaN output is the next stage's input), or from the chunk's start when a carried right state is
    // not.  The right smoothers are not run here; the tile says where its right input first goes non-finite (one granule,
    // published now, read by every later tile at the chain's end).
    __shared__ uint32_t rfirst_s[kThreads / 64];   // (read again at the chain's end: nothing is kept in a register across the stages)
    {
        uint32_t rf = 0xFFFFFFFFu;
#pragma unroll
        for (int j = NP - 1; j >= 0; --j) {
            const uint32_t m = mf + 2u * (uint32_t)j;
            if (m + 1u < M && !(x[j].w - x[j].w == 0.0f)) rf = m + 1u;
            if (m < M && !(x[j].y - x[j].y == 0.0f)) rf = m;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) rf = min(rf, (uint32_t)__shfl_xor((int)rf, off, 64));
        if (tile == 0u) {   // carried right states (a vertex about to be re-seeded from buf[0] carries none); a stage per lane:
            bool bad = false;   // tile 0 heads every dependency chain of the launch -- nothing serial in front of its first stage
            for (uint32_t s = lane; s < n_stages; s += 64u) {
                const BandStageDesc* const q = d.stages + s;   // (per-lane addresses: vector loads)
                if (__float_as_uint(st_l[s][4]) != 0u) continue;
                bad = bad || (q->lgamma != 0.0f && !(st_l[s][1] - st_l[s][1] == 0.0f)) || (q->hgamma != 0.0f && !(st_l[s][3] - st_l[s][3] == 0.0f));
            }
            if (__any(bad ? 1 : 0)) rf = 0u;
        }
        if (lane == 0u) rfirst_s[wave] = rf;
        __syncthreads();
        if (tid == 0u) granule_store(d.rpoison + tile, min(min(rfirst_s[0], rfirst_s[1]), min(rfirst_s[2], rfirst_s[3])));
    }
    uint32_t poisoned_at = n_stages;   // (wave 0) the first stage at which this tile's totals or entry state were not finite
    const double* pw_cur = nullptr;
    const double* pk_cur = nullptr;
    double pwl = 1.0, pwh = 1.0;     // (1 - gamma)^(NF lane)
    double w16l = 0.0, w16h = 0.0;   // (1 - gamma)^(NF ((lane % 16) + 1)): what the row before still weighs at this lane
    double w32l = 0.0, w32h = 0.0;   // (1 - gamma)^(NF ((lane % 32) + 1)): ... the half wave before
    double pkl = 0.0, pkh = 0.0;     // (1 - gamma)^(NF 256 lane): the weight of the tile lane + 1 tiles back
    const bool tail = wt0 + WT > M;  // (uniform: only the chunk's last wave-tiles have frames beyond M)
    const bool fin_here = mlast >= wt0 && mlast - wt0 < WT;
    // (timing experiments, flags bit 2: the wave's clock at eight points of every stage, written over its frames of the
    // output instead of the result -- tools/band_chain_phases.py)
    // (the descriptor's flag word, read ONCE: tested inside the poll loop -- between atomic loads the compiler must assume memory
    // changes -- it was a vector load + s_waitcnt vmcnt(0) in front of every stage's granule loads: a second, serialised memory
    // round trip per stage)
    const uint32_t dflags = (uint32_t)__builtin_amdgcn_readfirstlane((int)d.flags);
    const bool prof = (dflags & 4u) != 0u;
    unsigned long long* const pb = reinterpret_cast<unsigned long long*>(d.out + wt0);
    if (prof && lane == 0u && wt0 + WT <= M) {   // where the wave runs: HW_ID (wave / SIMD / CU / SE) and XCC_ID, and its tile
        pb[WT - 1u] = ((unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) << 32) |
                      (unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));
        pb[WT - 2u] = tile;
    }
    auto stamp = [&](uint32_t s, uint32_t k) { if (prof && lane == 0u && s < WT / 8u && wt0 + WT <= M) pb[s * 8u + k] = __builtin_readcyclecounter(); };
    // pan / gain over the lane's frames, each behind ONE uniform branch (written as per-element conditions they become
    // a multiply AND a select per element whatever the flags are; the empty asm keeps the branch a branch)
    auto pan_gain = [&x](float l_amp, float r_amp, float gain, uint32_t flags) {
        if (flags & 1u) {
            asm volatile("");
#pragma unroll
            for (int j = 0; j < NP; ++j) { x[j].x *= l_amp; x[j].y *= r_amp; x[j].z *= l_amp; x[j].w *= r_amp; }
        }
        if (flags & 2u) {
            asm volatile("");
#pragma unroll
            for (int j = 0; j < NP; ++j) { x[j].x *= gain; x[j].y *= gain; x[j].z *= gain; x[j].w *= gain; }
        }
    };

    // What a stage needs before its first instruction -- the two gammas, the table pointers, the first envelope link -- is
    // read ONE STAGE AHEAD, in one batch of scalar loads: read where it is used, behind `&&` and `if`, it was five dependent
    // scalar-cache round trips in front of every stage's recurrence.
    // (GUARD: the estimate's seven coefficients too -- they sit on a cache line of their own at the descriptor's end, and read
    // where they are used they were a scalar-cache MISS in front of every stage's output phase: 0.3 us per stage)
    struct StageHead { float lgam, hgam; const double* pw; const double* pk; uint32_t n_post; const float* env0; const float* env1;
                       float nzv0, nzv1, nzs0, nzs1, nzk0, nzk1; const float* envt; };
    auto head_of = [&](uint32_t s) {
        const BandStageDesc TD_CONST* const q = stages + s;
        if (GUARD) return StageHead{q->lgamma, q->hgamma, q->pw, q->pk, q->n_post, q->post[0].env, q->post[1].env,
                                    q->nzv[0], q->nzv[1], q->nzs[0], q->nzs[1], q->nzk[0], q->nzk[1], q->envt};
        return StageHead{q->lgamma, q->hgamma, q->pw, q->pk, q->n_post, q->post[0].env, q->post[1].env, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, nullptr};
    };
    const float* env_kept = nullptr;                                  // the envelope whose gains for this lane's frames sit in env_row
    float4* const env_row = xt + tid * (uint32_t)(NP + 1);           // (the lane's own row of the staging: one pad word per row, conflict-free)
    StageHead head = head_of(0u);
    // (GUARD) the lane's estimate: variance (one accumulator per smoother, added up at the end) and offset of the deviation at
    // its frames.  Every VALU instruction of this loop costs ~12 ticks of a stage's ~6 900 (three waves of a SIMD move in
    // lockstep), so the estimate is pared down: the static gains between a vertex and the chain's end are folded into its
    // coefficients by the host (only the envelope links' gain is applied here, in the stages that have one), the pair of
    // smoothers is one packed operand, and the parked test of the faster smoother is dropped where the slower one's
    // offset dwarfs it (host: nzk[1] 0).
    f32x2 nz_v2 = {0.0f, 0.0f};
    float nz_off = 0.0f;
    const float* nz_envt = nullptr;   // the envelope link whose gain over this wave-tile nz_e2 (mean square) / nz_e1 (RMS) hold
    float nz_e2 = 1.0f, nz_e1 = 1.0f;
    for (uint32_t s = 0; s < n_stages; ++s) {
        const BandStageDesc TD_CONST* const sp = stages + s;
        stamp(s, 0u);
        const StageHead next_head = head_of(min(s + 1u, n_stages - 1u));
        const float lgam = head.lgam, hgam = head.hgam;
        if (head.pw != pw_cur) {   // (uniform; the stages of a chain of identical filters share their tables: loaded once)
            pw_cur = head.pw;
            pwl = pw_cur[lane];
            pwh = pw_cur[64u + lane];
            w16l = pw_cur[(lane & 15u) + 1u]; w16h = pw_cur[64u + (lane & 15u) + 1u];
            w32l = pw_cur[(lane & 31u) + 1u]; w32h = pw_cur[64u + (lane & 31u) + 1u];
        }
        if (head.pk != pk_cur) {
            pk_cur = head.pk;
            pkl = pk_cur[lane]; pkh = pk_cur[kScanMaxK + lane];
        }
        const float* env_pre = ((s + 1u < n_stages || nd) && head.n_post) ? (head.env0 ? head.env0 : (head.n_post > 1u ? head.env1 : nullptr)) : nullptr;
        // ---- zero-state responses inside the lane's run
        // (the {low, high} smoother pair of a frame is one packed operand from here to the output)
        f32x2 z[NF];
        {
            const f32x2 gam = {lgam, hgam};
            f32x2 a = {0.0f, 0.0f};
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                a = __builtin_elementwise_fma(gam, f32x2{x[j].x, x[j].x} - a, a); z[2 * j] = a;
                a = __builtin_elementwise_fma(gam, f32x2{x[j].z, x[j].z} - a, a); z[2 * j + 1] = a;
            }
        }
        stamp(s, 1u);
        // ---- inclusive scan over the wave, on the VALU: b_i += a^(NF 2^q) b_(i - 2^q) inside the rows of 16 (a lane
        // without a source adds a^.. * 0), then the row before (rows 1, 3), then the half wave before (rows 2, 3)
        double b0 = (double)z[NF - 1].x, b2 = (double)z[NF - 1].y;
        b0 = __builtin_fma(dpp_f64<kDppRowShr + 1, 0xF>(b0), sp->ap[0][0], b0); b2 = __builtin_fma(dpp_f64<kDppRowShr + 1, 0xF>(b2), sp->ap[1][0], b2);
        b0 = __builtin_fma(dpp_f64<kDppRowShr + 2, 0xF>(b0), sp->ap[0][1], b0); b2 = __builtin_fma(dpp_f64<kDppRowShr + 2, 0xF>(b2), sp->ap[1][1], b2);
        b0 = __builtin_fma(dpp_f64<kDppRowShr + 4, 0xF>(b0), sp->ap[0][2], b0); b2 = __builtin_fma(dpp_f64<kDppRowShr + 4, 0xF>(b2), sp->ap[1][2], b2);
        b0 = __builtin_fma(dpp_f64<kDppRowShr + 8, 0xF>(b0), sp->ap[0][3], b0); b2 = __builtin_fma(dpp_f64<kDppRowShr + 8, 0xF>(b2), sp->ap[1][3], b2);
        b0 = __builtin_fma(dpp_f64<kDppRowBcast15, 0xA>(b0), w16l, b0);         b2 = __builtin_fma(dpp_f64<kDppRowBcast15, 0xA>(b2), w16h, b2);
        b0 = __builtin_fma(dpp_f64<kDppRowBcast31, 0xC>(b0), w32l, b0);         b2 = __builtin_fma(dpp_f64<kDppRowBcast31, 0xC>(b2), w32h, b2);
        const double e0 = dpp_f64<kDppWaveShr1, 0xF>(b0), e2 = dpp_f64<kDppWaveShr1, 0xF>(b2);   // the wave's response up to the lane's run
        if (lane == 63u) { wtot[wave][0] = b0; wtot[wave][1] = b2; }
        stamp(s, 2u);
        __syncthreads();   // barrier 1: the four waves' totals
        stamp(s, 3u);
        // the next link's envelope gains for the lane's frames: issued now -- the look-back's round trip lies ahead, and the
        // eight registers are not held through the recurrence and the scan --, used after the output
        float4 envv[NP / 2];
        // (wave 0 issues them BEHIND its serial section: in front of it they are four more loads the hand-off's own loads queue
        // behind -- vmcnt counts in order: 0.331 -> 0.321 ms)
        // The links of a chain mostly share ONE envelope (BASELINE config 4: all 84): the lane's sixteen gains are read from
        // global memory when the buffer changes and kept in the lane's own row of the staging area -- idle between the input
        // phase and the end phase -- from where every later stage takes them with four LDS reads: no vector-memory loads (and
        // none of the in-order waits they put in front of the hand-off's own loads) in the stage loop.
        auto load_env = [&]() {
            if (env_pre && env_pre == env_kept) {   // (uniform)
#pragma unroll
                for (int q = 0; q < NP / 2; ++q) envv[q] = env_row[q];
            } else {
#pragma unroll
                for (int q = 0; q < NP / 2; ++q)
                    envv[q] = (env_pre && mf + 4u * (uint32_t)q < M) ? gload4(env_pre + mf + 4u * (uint32_t)q) : make_float4(0.f, 0.f, 0.f, 0.f);
                if (env_pre) {
#pragma unroll
                    for (int q = 0; q < NP / 2; ++q) env_row[q] = envv[q];
                    env_kept = env_pre;
                }
            }
        };
        if (wave != 0u) load_env();
        double xw0 = 0.0, xw2 = 0.0, awp0 = 1.0, awp2 = 1.0;   // the tile's response up to this wave; a_wave^wave
        for (uint32_t w = 0; w < wave; ++w) {   // (uniform trip count)
            awp0 *= sp->aw[0]; awp2 *= sp->aw[1];
            xw0 = __builtin_fma(xw0, sp->aw[0], wtot[w][0]);
            xw2 = __builtin_fma(xw2, sp->aw[1], wtot[w][1]);
        }
        if (wave == 0u) {
            // The tile's serial section -- the other three waves wait at barrier 2 for it -- goes ahead of the other tiles' waves
            // on this SIMD (0.340 -> 0.331 ms on config 4: the first change to this kernel's arithmetic side that moved it)
            __builtin_amdgcn_s_setprio(3);
            // the whole tile's response (wave 0: xw is still 0), then the state at the chunk's first frame: carried, or seeded
            // from buf[0] (extensions.rs:664-670)
            double T0 = 0.0, T2 = 0.0;
#pragma unroll
            for (uint32_t w = 0; w < 4u; ++w) {
                T0 = __builtin_fma(T0, sp->aw[0], wtot[w][0]);
                T2 = __builtin_fma(T2, sp->aw[1], wtot[w][1]);
            }
            double C0 = 0.0, C2 = 0.0;
            uint32_t c_lane = 63u;   // the lane that ends up with the state entering the tile
            unsigned long long* const sync = sp->sync;
            if (tile == 0u) {
                const bool first = __float_as_uint(st_l[s][4]) != 0u;
                const float x00 = __shfl(x[0].x, 0, 64);
                const float yi0 = first ? x00 : st_l[s][0], yi2 = first ? x00 : st_l[s][2];
                if (first && lane == 0u) {   // a constant chain (gamma 0) keeps its seed for good
                    float* sf = reinterpret_cast<float*>(sp->state);
                    if (lgam == 0.0f) sf[0] = yi0;
                    if (hgam == 0.0f) sf[2] = yi2;
                }
                C0 = (double)yi0; C2 = (double)yi2;
                T0 = __builtin_fma(C0, sp->at[0], T0);   // (what this tile leaves behind, entered with the TRUE state)
                T2 = __builtin_fma(C2, sp->at[1], T2);
            }
            if (lane < 4u) {   // publish: granule q = chain (q >> 1) low / high word
                const unsigned long long u = (unsigned long long)__double_as_longlong(lane < 2u ? T0 : T2);
                granule_store(sync + (size_t)tile * 8u + lane, (lane & 1u) ? (uint32_t)(u >> 32) : (uint32_t)u);
            }
            stamp(s, 4u);
            if (tile != 0u) {
                const uint32_t n_pred = min(tile, sp->K);
                for (uint32_t base = 0; base < n_pred; base += 64u) {   // (one trip unless the look-back is deeper than 64 tiles)
                    const uint32_t j = base + lane;
                    const bool mine = j < n_pred;
                    const unsigned long long* g = sync + (size_t)(tile - 1u - (mine ? j : 0u)) * 8u;
                    unsigned long long g0 = 0, g1 = 0, g2 = 0, g3 = 0;
                    bool ok = !mine;

                    for (;;) {
                        if (dflags & 2u) break;   // (timing experiments only: no look-back -- wrong results)
                        if (!ok) {   // (a lane whose four words have arrived reads no more: the queue is left to the late ones)
                            g0 = granule_load(g); g1 = granule_load(g + 1); g2 = granule_load(g + 2); g3 = granule_load(g + 3);
                            ok = (uint32_t)(g0 >> 32) == 1u && (uint32_t)(g1 >> 32) == 1u && (uint32_t)(g2 >> 32) == 1u && (uint32_t)(g3 >> 32) == 1u;
                        }
                        if (__all(ok ? 1 : 0)) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (mine) {
                        const double B0 = __longlong_as_double((long long)((g0 & 0xFFFFFFFFull) | (g1 << 32)));
                        const double B2 = __longlong_as_double((long long)((g2 & 0xFFFFFFFFull) | (g3 << 32)));
                        // (tiles 65 + lane back -- a look-back deeper than 64 tiles, cut-offs below ~5 Hz -- fetch their weights here)
                        C0 = __builtin_fma(base ? pk_cur[64u + lane] : pkl, B0, C0);
                        C2 = __builtin_fma(base ? pk_cur[kScanMaxK + 64u + lane] : pkh, B2, C2);
                    }
                }
                // the wave's sum into lane 63: running sums inside the rows, then row / half-wave totals passed on
                C0 += dpp_f64<kDppRowShr + 1, 0xF>(C0); C2 += dpp_f64<kDppRowShr + 1, 0xF>(C2);
                C0 += dpp_f64<kDppRowShr + 2, 0xF>(C0); C2 += dpp_f64<kDppRowShr + 2, 0xF>(C2);
                C0 += dpp_f64<kDppRowShr + 4, 0xF>(C0); C2 += dpp_f64<kDppRowShr + 4, 0xF>(C2);
                C0 += dpp_f64<kDppRowShr + 8, 0xF>(C0); C2 += dpp_f64<kDppRowShr + 8, 0xF>(C2);
                if (n_pred > 16u) {   // (uniform; up to 16 predecessors sit in row 0: its total is in lane 15, the other rows add zeros)
                    C0 += dpp_f64<kDppRowBcast15, 0xA>(C0); C2 += dpp_f64<kDppRowBcast15, 0xA>(C2);
                    C0 += dpp_f64<kDppRowBcast31, 0xC>(C0); C2 += dpp_f64<kDppRowBcast31, 0xC>(C2);
                } else {
                    c_lane = 15u;
                }
            }
            // (x - x == 0 only for finite x: the tile's response -- every lane's run feeds it -- and, in lane 63, the state entering it)
            if (poisoned_at == n_stages && __any((!(T0 - T0 == 0.0) || !(T2 - T2 == 0.0) || !(C0 - C0 == 0.0) || !(C2 - C2 == 0.0)) ? 1 : 0)) poisoned_at = s;
            if (lane == c_lane) { carry_s[0] = C0; carry_s[1] = C2; }
            // (the last stage: the tile's verdict is final -- out it goes now, a stage's output phase ahead of the gather at the end)
            if (s + 1u == n_stages && lane == 0u) granule_store(d.poison + tile, poisoned_at);
            stamp(s, 5u);
            __builtin_amdgcn_s_setprio(0);
            load_env();
        }
        __syncthreads();   // barrier 2: the state entering the tile
        stamp(s, 6u);
        // the lane's entry state, exact arithmetic rounded once: e + a^(NF lane) (xw + a_wave^wave C)
        const f32x2 c = {(float)__builtin_fma(pwl, __builtin_fma(awp0, carry_s[0], xw0), e0),
                         (float)__builtin_fma(pwh, __builtin_fma(awp2, carry_s[1], xw2), e2)};
        const f32x2 TD_CONST* const pn = (const f32x2 TD_CONST*)(const TD_CONST char*)sp->pn;
        // what this vertex' two smoothers add, from the level of the lane's entry state (host: all 0 for a constant chain, and
        // for every stage but the first of a run of identical filters, which stands for the run)
        if (GUARD && (head.nzk0 != 0.0f || head.nzv0 != 0.0f || head.nzv1 != 0.0f)) {   // (uniform)
            asm volatile("");
            const float x0 = x[0].x;
            nz_v2 = __builtin_elementwise_fma(f32x2{head.nzv0, head.nzv1}, c * c, nz_v2);
            const float al = fabsf(c.x);
            nz_off = __builtin_fmaf(fabsf(x0 - c.x) < head.nzk0 * al ? head.nzs0 : 0.0f, al, nz_off);   // parked: |gamma (x - y)| below 4 ulp(y)
            if (head.nzk1 != 0.0f) {   // (uniform)
                asm volatile("");
                const float ah = fabsf(c.y);
                nz_off = __builtin_fmaf(fabsf(x0 - c.y) < head.nzk1 * ah ? head.nzs1 : 0.0f, ah, nz_off);
            }
        }
        // ---- output (extensions.rs:682-687 with cut_mul 0, pass_mul 1), then the vertex' pan / gain
        const bool lo_on = lgam != 0.0f, hi_on = hgam != 0.0f;
        if (fin_here) {   // the lane that holds the chunk's last frame carries the state over (one wave-tile of the chunk)
            asm volatile("");
            const uint32_t nl = mlast - mf;   // (< NF on that lane only)
            f32x2 f = {0.f, 0.f};
#pragma unroll
            for (int n = 0; n < NF; ++n)
                if (nl == (uint32_t)n) f = __builtin_elementwise_fma(pn[n], c, z[n]);
            const bool has_fin = nl < (uint32_t)NF;
            if (has_fin) {
                if (!state_may_be_written) {
                    while (__hip_atomic_load((gu32)(TD_GLOBAL char*)(d.ticket + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u)
                        __builtin_amdgcn_s_sleep(8);
                }
                float* sf = reinterpret_cast<float*>(sp->state);
                if (lo_on) sf[0] = f.x;
                if (hi_on) sf[2] = f.y;
                sp->state->first = 0u;
            }
            state_may_be_written = state_may_be_written || __any(has_fin ? 1 : 0) != 0;
        }
        if (lo_on && hi_on) {
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const float4 v = x[j];
                const f32x2 ya = __builtin_elementwise_fma(pn[2 * j], c, z[2 * j]), yb = __builtin_elementwise_fma(pn[2 * j + 1], c, z[2 * j + 1]);
                // (l - cut, r - cut), cut = (low + (l - high)) / 2: the halving is exact, so it rides in the subtraction's fma
                const float sa = ya.x + (v.x - ya.y), sb = yb.x + (v.z - yb.y);
                const f32x2 mh = {-0.5f, -0.5f};
                const f32x2 oa = __builtin_elementwise_fma(f32x2{sa, sa}, mh, f32x2{v.x, v.y}), ob = __builtin_elementwise_fma(f32x2{sb, sb}, mh, f32x2{v.z, v.w});
                x[j] = make_float4(oa.x, oa.y, ob.x, ob.y);
            }
        } else {
            asm volatile("");
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const float4 v = x[j];
                const f32x2 ya = __builtin_elementwise_fma(pn[2 * j], c, z[2 * j]), yb = __builtin_elementwise_fma(pn[2 * j + 1], c, z[2 * j + 1]);
                const float ca = ((lo_on ? ya.x : 0.0f) + (hi_on ? v.x - ya.y : 0.0f)) * 0.5f;
                const float cb = ((lo_on ? yb.x : 0.0f) + (hi_on ? v.z - yb.y : 0.0f)) * 0.5f;
                x[j] = make_float4(v.x - ca, v.y - ca, v.z - cb, v.w - cb);
            }
        }
        pan_gain(sp->pg.l_amp, sp->pg.r_amp, sp->pg.gain, sp->pg.flags);
        stamp(s, 7u);
        if (s + 1u < n_stages || nd) {
            // The links to the next band-pass vertex (after the last stage: to the Normalize vertex): an Adsr vertex multiplies by its gain of the frame (k_adsr_env), then
            // pan / gain; a single-input Sum is pan / gain only.  Every link and the next vertex start with their own
            // sum_inputs `0.0 + x`, whose only effect is -0 -> +0: a zero's sign moves no value anywhere in the next stage,
            // and what reads the chain's output starts with a `0.0 + x` of its own -- none here (tolerance class).
            // Frames at or beyond M become 0, as sum_terms leaves them.
            const uint32_t np = sp->n_post;
            for (uint32_t p = 0; p < np; ++p) {
                const float* env = sp->post[p].env;
                if (env) {
                    if (env != env_pre) {   // (a second envelope link of the same hop: not prefetched)
                        asm volatile("");
#pragma unroll
                        for (int q = 0; q < NP / 2; ++q)
                            envv[q] = mf + 4u * (uint32_t)q < M ? gload4(env + mf + 4u * (uint32_t)q) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                    for (int q = 0; q < NP / 2; ++q) {
                        const float4 e = envv[q];
                        const float4 a = x[2 * q], b = x[2 * q + 1];
                        x[2 * q] = make_float4(a.x * e.x, a.y * e.x, a.z * e.y, a.w * e.y);
                        x[2 * q + 1] = make_float4(b.x * e.z, b.y * e.z, b.z * e.w, b.w * e.w);
                    }
                }
                pan_gain(sp->post[p].pg.l_amp, sp->post[p].pg.r_amp, sp->post[p].pg.gain, sp->post[p].pg.flags);
            }
            if (tail) {
#pragma unroll
                for (int j = 0; j < NP; ++j) x[j] = zero_tail(x[j], mf + 2u * (uint32_t)j, M);
            }
            // (GUARD) the estimate goes through the envelope link like the frames do -- by the link's RMS gain over this wave's
            // 1 024 frames, ONE scalar load (AdsrVDesc::env_tile).  (Taken from the per-frame gains the link loop above holds in
            // registers, the three instructions cost the loop 0.25 us per stage, wherever they were put.)
            if (GUARD && head.envt) {
                if (head.envt != nz_envt) {   // (uniform; the links of a chain mostly share ONE envelope: read once, like the power tables)
                    nz_envt = head.envt;
                    const float TD_CONST* const et = (const float TD_CONST*)(const TD_CONST char*)head.envt;
                    const uint32_t h0 = 2u * (uint32_t)__builtin_amdgcn_readfirstlane((int)wt);   // (the table holds mean squares per 512 frames)
                    nz_e2 = wt0 + WT / 2u < M ? 0.5f * (et[h0] + et[h0 + 1u]) : et[h0];
                    nz_e1 = __builtin_sqrtf(nz_e2);
                }
                nz_v2 *= f32x2{nz_e2, nz_e2};
                nz_off *= nz_e1;
            }
        }
        head = next_head;
    }
    if (prof) return;
    // ---- the chain's end.  Once NaN, always NaN: the reference's smoother state never recovers, the look-back above forgets a
    // tile after K tiles -- so every tile says at which stage it first went non-finite (one granule) and learns the same of ALL
    // earlier tiles: if one did, every frame here is NaN from that stage on, and so are the states carried out of the chunk.
    // With a Normalize vertex behind the chain (extensions.rs:310-329; fresh-render form as in k_norm1, its block is this wave's
    // 1 024 frames: buf = 0.0 + x; max = buf_max.max(max) over the blocks so far; buf * (1.0 / max); pan / gain; the sink's
    // quantiser when it is the output) the tile's maximum goes out beside that granule and ONE gather brings both in.
    __shared__ uint32_t pz[kThreads / 64], pt[kThreads / 64];
    __shared__ float nwm[kThreads / 64], npm[kThreads / 64];
    __shared__ float nz_w[kThreads / 64], nz_p[kThreads / 64];   // (GUARD) the waves' energies; the gather's partial sums
    const float qnan = __uint_as_float(0x7FC00000u);
    // (the end phase computes its frame indices and addresses afresh, from a lane number and descriptor pointers the compiler
    // cannot see through: shared with the input phase's identical expressions they were kept -- i.e. spilled to scratch, the
    // stage loop runs at the register cap -- across all the stages: ~5 us per launch)
    const BandScanDesc* dl = &d;
    asm volatile("" : "+v"(dl));
    const SumDesc TD_CONST* ndl = nd;
    asm volatile("" : "+v"(ndl));
    uint32_t lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const uint32_t mf_e = wt0 + (uint32_t)NF * lane_e;
    float4* const xw4_e = xt + wave * 64u * (uint32_t)(NP + 1);
    {   // the right channel from the first frame at which any tile's right input went non-finite (see rpoison above)
        uint32_t r_from = min(min(rfirst_s[0], rfirst_s[1]), min(rfirst_s[2], rfirst_s[3]));   // the tile's own
        (void)for_lower_granules(dl->rpoison, tile, 0xFFFFFFFFu, [&r_from](uint32_t, uint32_t v) { r_from = min(r_from, v); });
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) r_from = min(r_from, (uint32_t)__shfl_xor((int)r_from, off, 64));
        if (lane_e == 0u) pz[wave] = r_from;
        __syncthreads();
        r_from = min(min(pz[0], pz[1]), min(pz[2], pz[3]));
        __syncthreads();   // (pz is used again below)
        if (r_from <= mlast) {   // (rare)
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const uint32_t m = mf_e + 2u * (uint32_t)j;
                if (m >= r_from) x[j].y = qnan;
                if (m + 1u >= r_from) x[j].w = qnan;
            }
            if (fin_here) {   // the right states carried out of the chunk: every stage's (the NaN output of one is the input of the next)
                for (uint32_t s = lane_e; s < n_stages; s += 64u) {
                    float* sf = reinterpret_cast<float*>(stages[s].state);
                    if (stages[s].lgamma != 0.0f) sf[1] = qnan;
                    if (stages[s].hgamma != 0.0f) sf[3] = qnan;
                }
            }
        }
    }
    if (ndl) {
        float pk = 0.0f;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            x[j] = add4(make_float4(0.f, 0.f, 0.f, 0.f), x[j]);
            const uint32_t m = mf_e + 2u * (uint32_t)j;
            if (m < M) pk = fmaxf(fmaxf(pk, fabsf(x[j].x)), fabsf(x[j].y));   // (absmaxlen's fold: NaNs never win)
            if (m + 1u < M) pk = fmaxf(fmaxf(pk, fabsf(x[j].z)), fabsf(x[j].w));
        }
        pk = wave_max(pk);
        if (lane_e == 0u) nwm[wave] = pk;
        __syncthreads();
    }
    if (tid == 0u) {
        if (ndl) {
            asm volatile("" ::"v"(norm_init));   // (the carried max has been READ before this tile counts as published)
            granule_store(ndl->sync + tile, __float_as_uint(fmaxf(fmaxf(nwm[0], nwm[1]), fmaxf(nwm[2], nwm[3]))));
            if (tile == 0u) {
                ndl->init_copy[0] = norm_init;
                ndl->init_copy[1] = ndl->state->scan_max;
            }
        }
    }
    uint32_t pmin = n_stages, ptile = 0xFFFFFFFFu;   // the earliest stage an earlier tile went non-finite at; the first such tile
    float pm = 0.0f;                                 // the largest block peak of the earlier tiles
    {
        // (lower TICKETS: their holders are running and reach this point without waiting for anybody above them -- the
        // wait needs no bound, like the look-back's)
        const uint32_t ns = n_stages;
        if (ndl) (void)for_lower_granules2(dl->poison, ndl->sync, tile, 0xFFFFFFFFu, [&](uint32_t idx, uint32_t p, uint32_t v) {
                pmin = min(pmin, p);
                if (p < ns) ptile = min(ptile, idx);
                pm = fmaxf(pm, __uint_as_float(v)); });
        else (void)for_lower_granules(dl->poison, tile, 0xFFFFFFFFu, [&](uint32_t idx, uint32_t p) { pmin = min(pmin, p); if (p < ns) ptile = min(ptile, idx); });
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        pmin = min(pmin, (uint32_t)__shfl_xor((int)pmin, off, 64));
        ptile = min(ptile, (uint32_t)__shfl_xor((int)ptile, off, 64));
    }
    pm = wave_max(pm);
    if (lane_e == 0u) { pz[wave] = pmin; pt[wave] = ptile; npm[wave] = pm; }
    __syncthreads();
    pmin = min(min(pz[0], pz[1]), min(pz[2], pz[3]));
    ptile = min(min(pt[0], pt[1]), min(pt[2], pt[3]));
    pm = fmaxf(fmaxf(npm[0], npm[1]), fmaxf(npm[2], npm[3]));
    const bool poisoned = pmin < n_stages;
    if (poisoned) {   // (rare)
#pragma unroll
        for (int j = 0; j < NP; ++j) x[j] = make_float4(qnan, qnan, qnan, qnan);
        if (fin_here) {
            for (uint32_t s = pmin + lane_e; s < n_stages; s += 64u) {
                float* sf = reinterpret_cast<float*>(stages[s].state);
                if (stages[s].lgamma != 0.0f) sf[0] = qnan;
                if (stages[s].hgamma != 0.0f) sf[2] = qnan;
            }
        }
        if (ndl) {
            // this tile's blocks hold nothing but NaN: no peak; and what the tiles after the first poisoned one published
            // was measured on frames that are NaN in truth: the maximum so far is that of the tiles up to it
            __syncthreads();
            if (lane_e == 0u) nwm[wave] = 0.0f;
            float pm2 = 0.0f;
            const uint32_t upto = min(ptile + 1u, tile);
            (void)for_lower_granules(ndl->sync, upto, 0xFFFFFFFFu, [&pm2](uint32_t, uint32_t v) { pm2 = fmaxf(pm2, __uint_as_float(v)); });
            pm2 = wave_max(pm2);
            if (lane_e == 0u) npm[wave] = pm2;
            __syncthreads();
            pm = fmaxf(fmaxf(npm[0], npm[1]), fmaxf(npm[2], npm[3]));
        }
    }
    if (ndl) {
        if (lane_e == 0u && wt0 < M) ndl->peaks[wt] = nwm[wave];
        float run = fmaxf(pm, norm_init);                                   // max_(b-1) entering the tile's first block
        for (uint32_t w = 0; w <= wave; ++w) run = fmaxf(nwm[w], run);      // *max = buf_max.max(*max)
        const float r = 1.0f / run;
        if (GUARD) {   // the estimate goes through the Normalize vertex like the frames do: 1 / max, its pan / gain (largest channel)
            const float gq = fabsf(r) * dl->nz_end;
            const float nz_var = nz_v2.x + nz_v2.y;
            float e = mf_e < M ? (float)NF * __builtin_fmaf(nz_off * gq, nz_off * gq, nz_var * gq * gq) : 0.0f;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) e += __shfl_xor(e, off, 64);
            unsigned long long* const nzs = dl->nz_sync;
            if (nzs) {
                // ... and what the graph's probed sine vertices measured at their own outputs (k_sine_probe, earlier on the stream;
                // BandScanDesc::nz_extra), through the same 1 / max: this wave-tile's share of its probe group's energy
                if (dl->nz_probe) {   // (uniform) measured by this tile at its start: samples 4 wave .. 4 wave + 3 are this wave-tile's
                    float xe = lane_e < 4u ? nzx_s[wave * 4u + lane_e] * dl->nz_xg2[0] : 0.0f;
                    xe += __shfl_xor(xe, 1, 64);
                    xe += __shfl_xor(xe, 2, 64);
                    e += xe * gq * gq;   // (lanes 0 .. 3 hold the sum; lane 0 is the one that counts below)
                } else if (dl->nz_extra[0] && wt0 < M) {   // (uniform)
                    const uint32_t cnt = dl->nz_xcnt, i0 = wt * cnt + lane_e;   // this wave-tile's samples: cnt <= 64, one a lane
                    const uint32_t ns = (M + (uint32_t)kTileFrames / cnt - 1u) / ((uint32_t)kTileFrames / cnt);
                    float xe = 0.0f;
                    if (lane_e < cnt && i0 < ns) {
                        xe = gload1(dl->nz_extra[0] + i0) * dl->nz_xg2[0];
                        if (dl->nz_extra[1]) xe += gload1(dl->nz_extra[1] + i0) * dl->nz_xg2[1];
                    }
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) xe += __shfl_xor(xe, off, 64);
                    e += xe * gq * gq;
                }
                // The graph's only guarded launch: the verdict right here (kernels.h BandScanDesc::nz_sync).  Every tile leaves its
                // energy as ONE granule -- a plain tagged store: 2 813 atomic adds to one word took 60 us of the launch -- and the
                // tile with the last ticket, which every other ticket holder is running ahead of or beside, gathers them.
                if (lane_e == 0u) nz_w[wave] = (wt0 < M && e == e) ? e : 0.0f;   // (NaN: a tile the reference turns NaN as well)
                __syncthreads();
                const float mine = (nz_w[0] + nz_w[1]) + (nz_w[2] + nz_w[3]);
                if (threadIdx.x == 0u) granule_store(nzs + tile, __float_as_uint(mine));
                if (tile + 1u == dl->n_tiles) {
                    float sum = 0.0f;
                    (void)for_lower_granules(nzs, tile, 0xFFFFFFFFu, [&sum](uint32_t, uint32_t v) { sum += __uint_as_float(v); });
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
                    if (lane_e == 0u) nz_p[wave] = sum;
                    __syncthreads();
                    if (threadIdx.x == 0u) {
                        const float ms = (((nz_p[0] + nz_p[1]) + (nz_p[2] + nz_p[3])) + mine) * dl->nz_scale;
                        __hip_atomic_store((gu32)(TD_GLOBAL char*)(dl->nz_host + 1), __float_as_uint(sqrtf(ms)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        if (!(ms <= dl->nz_thr2)) __hip_atomic_store((gu32)(TD_GLOBAL char*)dl->nz_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
            } else if (lane_e == 0u && wt0 < M) {
                dl->noise[wt] = e;
            }
        }
        PanGain npg;
        npg.l_amp = ndl->pg.l_amp; npg.r_amp = ndl->pg.r_amp; npg.gain = ndl->pg.gain; npg.flags = ndl->pg.flags;
#pragma unroll
        for (int j = 0; j < NP; ++j)
            xw4_e[lane_e * (uint32_t)(NP + 1) + (uint32_t)j] = epilogue4(make_float4(x[j].x * r, x[j].y * r, x[j].z * r, x[j].w * r), npg);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        float2* const nout = ndl->out;
        void* const npcm = ndl->pcm;
        const uint32_t nq = ndl->qmode;
        const float namp = ndl->amplitude;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const uint32_t m = wt0 + 2u * ((uint32_t)q * 64u + lane_e);
            const float4 v = xw4_e[slot((uint32_t)q * 64u + lane_e)];
            if (nout) store_pair(nout, m, M, v);
            if (nq) store_quant_pair(npcm, nq, m, M, v, namp);
        }
        // the wave holding the chunk's last frame has the running max of the chunk's last block
        if (fin_here && lane_e == 0u) const_cast<NormState*>((const NormState*)ndl->state)->max = run;
        return;
    }
    if (GUARD) {
        float e = mf_e < M ? (float)NF * __builtin_fmaf(nz_off, nz_off, nz_v2.x + nz_v2.y) : 0.0f;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) e += __shfl_xor(e, off, 64);
        if (lane_e == 0u && wt0 < M) dl->noise[wt] = e;
    }
    // the last vertex' output, back through the wave's staging for coalesced stores
#pragma unroll
    for (int j = 0; j < NP; ++j) xw4_e[lane_e * (uint32_t)(NP + 1) + (uint32_t)j] = x[j];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
#pragma unroll
    for (int q = 0; q < NP; ++q) store_pair(dl->out, wt0 + 2u * ((uint32_t)q * 64u + lane_e), M, xw4_e[slot((uint32_t)q * 64u + lane_e)]);
}

// ------------------------------------------------------------------------------------------------
// k_band_audit: the guard's verdict (AuditHead / AuditDesc, kernels.h) -- one workgroup per graph of the submission
// ------------------------------------------------------------------------------------------------
// Per scan launch: sum over its wave-tiles of  energy x (static gain to the output)^2 x (1 / running max of the Normalize
// vertex on the way, at the tile's first frame)^2.  The running max of block b is max(carried max, peaks[0 .. b])
// (extensions.rs:321-329): segment maxima of the peak table, their prefix, then a short walk inside the tile's segment.
// Launches add up as amplitudes (the same rounding pattern may reach the output twice).  NaN energies (tiles the
// reference turns NaN as well) count as 0, infinite ones trip the guard.
constexpr uint32_t kAuditBlocks = 8192;   // running maxima held in LDS (a longer chunk of shorter blocks walks the peak table per entry)
__global__ __launch_bounds__(kThreads) void k_band_audit(const AuditHead* __restrict__ heads) {
    const AuditHead h = heads[blockIdx.x];
    __shared__ float seg_max[kThreads], seg_pre[kThreads + 1], part[kThreads];
    __shared__ float runmax[kAuditBlocks];   // max(carried max, peaks[0 .. b]) of the Normalize vertex whose table was read last
    const uint32_t tid = threadIdx.x;
    float amp = 0.0f;   // (thread 0) sum over the launches of sqrt(energy at the output)
    const float* have = nullptr;   // the peak table `runmax` / `seg_pre` stand for (k_sine_probe's sources of one graph share one)
    const float* have_init = nullptr;
    for (uint32_t i = 0; i < h.n; ++i) {
        const AuditDesc d = h.descs[i];
        const uint32_t seg = d.peaks ? (d.nb + (uint32_t)kThreads - 1u) / (uint32_t)kThreads : 0u;
        const bool in_lds = d.nb <= kAuditBlocks;
        if (d.peaks && !(d.peaks == have && d.init_copy == have_init)) {
            __syncthreads();
            float m = 0.0f;
            if (in_lds) {   // the table through LDS: coalesced loads, every thread then walks its own segment there
                for (uint32_t b = tid; b < d.nb; b += (uint32_t)kThreads) runmax[b] = gload1(d.peaks + b);
                __syncthreads();
                for (uint32_t b = tid * seg; b < min(d.nb, (tid + 1u) * seg); ++b) m = fmaxf(m, runmax[b]);
            } else {
                for (uint32_t b = tid * seg; b < min(d.nb, (tid + 1u) * seg); ++b) m = fmaxf(m, gload1(d.peaks + b));
            }
            // seg_pre[q] = max(carried max, peaks of the segments before q): an inclusive scan of the segment maxima, shifted by one
            for (uint32_t off = 1u; off < (uint32_t)kThreads; off <<= 1) {
                seg_max[tid] = m;
                __syncthreads();
                if (tid >= off) m = fmaxf(m, seg_max[tid - off]);
                __syncthreads();
            }
            seg_max[tid] = m;
            __syncthreads();
            const float init = gload1(d.init_copy);
            seg_pre[tid] = tid ? fmaxf(init, seg_max[tid - 1u]) : init;
            if (in_lds) {
                float run = seg_pre[tid];
                for (uint32_t b = tid * seg; b < min(d.nb, (tid + 1u) * seg); ++b) { run = fmaxf(run, runmax[b]); runmax[b] = run; }
            }
            __syncthreads();
            have = d.peaks;
            have_init = d.init_copy;
        }
        float e = 0.0f;
        const uint32_t lg_tf = 31u - (uint32_t)__clz((int)max(d.tile_frames, 1u));
        const uint32_t bl_sh = (d.bl & (d.bl - 1u)) == 0u ? 31u - (uint32_t)__clz((int)max(d.bl, 1u)) : 0xFFFFFFFFu;   // (block length a power of two: a shift)
        auto inv_max2 = [&](uint32_t frame) {   // (1 / the Normalize vertex' running max at `frame`)^2
            const uint32_t b0 = min(d.nb - 1u, bl_sh != 0xFFFFFFFFu ? frame >> bl_sh : frame / d.bl);
            float run;
            if (in_lds) {
                run = runmax[b0];
            } else {
                const uint32_t sg = b0 / seg;
                run = seg_pre[sg];
                for (uint32_t b = sg * seg; b <= b0; ++b) run = fmaxf(run, gload1(d.peaks + b));
            }
            const float r = 1.0f / run;
            return r * r;
        };
        // k_sine_probe's samples four at a time where four consecutive ones lie in one block (60 s at 1 024-frame blocks: 11 000
        // samples, one 16-byte load and one look-up per four); everything else entry by entry
        uint32_t w_done = 0u;
        if (d.sampled && d.peaks && d.bl % (4u * d.tile_frames) == 0u && (reinterpret_cast<uintptr_t>(d.noise) & 15u) == 0u) {
            const uint32_t nq = d.n_wt / 4u;
            for (uint32_t q = tid; q < nq; q += (uint32_t)kThreads) {
                const float4 v = gload4(d.noise + 4u * q);
                const float sum = (fmaxf(v.x, 0.0f) + fmaxf(v.y, 0.0f)) + (fmaxf(v.z, 0.0f) + fmaxf(v.w, 0.0f));   // (NaN -> 0)
                e += sum * inv_max2((4u * q) << lg_tf);
            }
            w_done = nq * 4u;
        }
        for (uint32_t w = w_done + tid; w < d.n_wt; w += (uint32_t)kThreads) {
            float v = fmaxf(gload1(d.noise + w), 0.0f);   // (NaN -> 0)
            // (a scan launch's tile goes through the running max at its first frame; a probe's sample through that at its own frame)
            if (d.peaks) v *= inv_max2(d.sampled ? probe_frame(w, lg_tf) : w * d.tile_frames);
            e += v;
        }
        part[tid] = e;
        __syncthreads();
        for (uint32_t st = (uint32_t)kThreads / 2u; st > 0u; st >>= 1) {
            if (tid < st) part[tid] += part[tid + st];
            __syncthreads();
        }
        if (tid == 0u) amp += sqrtf(part[0]) * fabsf(d.gain);
        __syncthreads();
    }
    if (tid == 0u) {
        const float ms = amp * amp / (float)max(h.frames, 1u);   // estimated mean square of the deviation over the chunk's frames
        __hip_atomic_store((gu32)(TD_GLOBAL char*)(h.host_word + 1), __float_as_uint(sqrtf(ms)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (!(ms <= h.thr2)) __hip_atomic_store((gu32)(TD_GLOBAL char*)h.host_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
void launch_band_audit(const AuditHead* heads, int n_heads, hipStream_t s) {
    if (n_heads > 0) hipLaunchKernelGGL(k_band_audit, dim3((uint32_t)n_heads), dim3(kThreads), 0, s, heads);
}

// ------------------------------------------------------------------------------------------------
// k_resample: build-defined windowed-sinc resampler (specification in DESIGN.md "Resampler"; the oracle
// implements the same arithmetic: taps in order k = 0..255 through BOTH neighbouring phases, f32 accumulate, then
// (1 - a) y0 + a y1)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_resample(ResampleDesc d) {
    for (uint64_t j = (uint64_t)blockIdx.x * kThreads + threadIdx.x; j < d.nout; j += (uint64_t)gridDim.x * kThreads) {
        const uint64_t num = j * d.from;
        const int64_t i0 = (int64_t)(num / d.to);
        const uint64_t ph = (num % d.to) * (uint64_t)kSincOver;
        const uint64_t p = ph / d.to;
        const float a = (float)(ph % d.to) / (float)d.to;
        const float* __restrict__ t0 = d.table + p * kSincLen;
        const float* __restrict__ t1 = t0 + kSincLen;
        const float one_minus_a = 1.0f - a;
        float al0 = 0.0f, ar0 = 0.0f, al1 = 0.0f, ar1 = 0.0f;   // the two neighbouring phases' convolutions
        for (int k = 0; k < kSincLen; ++k) {
            const int64_t idx = i0 - (127 + kSincLen / 2) + k;   // (delayed by sinc_len / 2 input frames, see kernels.h)
            if (idx < 0 || idx >= (int64_t)d.len) continue;
            const float2 x = d.in[idx];
            al0 += x.x * t0[k];
            ar0 += x.y * t0[k];
            al1 += x.x * t1[k];
            ar1 += x.y * t1[k];
        }
        d.out[j] = make_float2(one_minus_a * al0 + a * al1, one_minus_a * ar0 + a * ar1);   // interp_lin over the RESULTS
    }
}

// ------------------------------------------------------------------------------------------------
// sample load pipeline (SampleBank::add, sample.rs:262-303)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_pcm_decode(const uint8_t* __restrict__ raw, float* __restrict__ out, uint32_t n,
                                                         uint32_t format) {
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        float v;
        switch (format) {
            case PCM_U8: v = (float)((int)raw[i] - 128); break;
            case PCM_S16: v = (float)(int16_t)((uint32_t)raw[2 * i] | ((uint32_t)raw[2 * i + 1] << 8)); break;
            case PCM_S24: {
                int32_t w = (int32_t)((uint32_t)raw[3 * i] | ((uint32_t)raw[3 * i + 1] << 8) | ((uint32_t)raw[3 * i + 2] << 16));
                if (w & 0x800000) w |= ~0xFFFFFF;
                v = (float)w;
            } break;
            case PCM_S32: v = (float)(int32_t)((uint32_t)raw[4 * i] | ((uint32_t)raw[4 * i + 1] << 8) | ((uint32_t)raw[4 * i + 2] << 16) |
                                                ((uint32_t)raw[4 * i + 3] << 24)); break;
            default: v = __uint_as_float((uint32_t)raw[4 * i] | ((uint32_t)raw[4 * i + 1] << 8) | ((uint32_t)raw[4 * i + 2] << 16) |
                                         ((uint32_t)raw[4 * i + 3] << 24));
        }
        out[i] = v;
    }
}
__global__ __launch_bounds__(kThreads) void k_sample_split(const float* __restrict__ lin, uint32_t channels, uint32_t src_l,
                                                           uint32_t src_r, float* __restrict__ l, float* __restrict__ r,
                                                           uint32_t nl, uint32_t nr) {
    const uint32_t n = max(nl, nr);
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        if (i < nl) l[i] = lin[(size_t)i * channels + src_l];
        if (i < nr) r[i] = lin[(size_t)i * channels + src_r];
    }
}
__global__ __launch_bounds__(kThreads) void k_absmax_atomic(const float* __restrict__ v, uint32_t n, float* out) {
    float m = 0.0f;
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        const float a = fabsf(v[i]);
        if (a > m) m = a;   // absmaxlen's fold: NaN never wins
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0 && m > 0.0f) atomicMax(reinterpret_cast<unsigned int*>(out), __float_as_uint(m));
}
// mean_energy (sample.rs:16-22) sums |s| left to right in f32: the order is part of the result.  One
// workgroup stages 4096 values at a time in LDS, lane 0 adds them in order.
__global__ __launch_bounds__(kThreads) void k_abs_sum_serial(const float* __restrict__ v, uint32_t n, float* out) {
    __shared__ float tile[4096];
    float sum = 0.0f;
    for (uint32_t base = 0; base < n; base += 4096) {
        const uint32_t cnt = min(4096u, n - base);
        for (uint32_t i = threadIdx.x; i < cnt; i += kThreads) tile[i] = fabsf(v[base + i]);
        __syncthreads();
        if (threadIdx.x == 0)
            for (uint32_t i = 0; i < cnt; ++i) sum += tile[i];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = sum;
}
__global__ __launch_bounds__(kThreads) void k_add_planar(const float* __restrict__ a, const float* __restrict__ b,
                                                         float* __restrict__ o, uint32_t n) {
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) o[i] = a[i] + b[i];
}
__global__ __launch_bounds__(kThreads) void k_sample_pack(const float* __restrict__ l, const float* __restrict__ r,
                                                          const float* max_l, const float* max_r, float2* __restrict__ frames,
                                                          uint32_t n) {
    const float sl = 1.0f / *max_l, sr = 1.0f / *max_r;   // `1.0 / max`, then multiply (sample.rs:127-129)
    // (frames n .. n + 14 = the loop's first frames again: a looping reader takes up to 16 consecutive frames
    // behind one modulo)
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n + 15u; i += gridDim.x * kThreads) {
        const uint32_t f = i < n ? i : (i - n) % n;
        frames[i] = make_float2(l[f] * sl, r[f] * sr);
    }
}

// ------------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------------
static inline uint32_t tiles(uint32_t frames) { return (frames + kTileFrames - 1) / kTileFrames; }
constexpr int kMaxGridY = 65535;

// Batched launches put the vertex index in grid.y (limit 65535): larger batches go out in slices.
#define TD_BATCHED(KERNEL, GRID_X, BLOCK, D, N, ...)                                                        \
    for (int o_ = 0; o_ < (N); o_ += kMaxGridY)                                                             \
        hipLaunchKernelGGL(KERNEL, dim3((GRID_X), std::min((N) - o_, kMaxGridY)), dim3(BLOCK), 0, s, (D) + o_, __VA_ARGS__)

static const auto k_sum32w_2 = &k_sum16w<2, false>;   // (names without a comma for the launch macro)
// Workgroups of a k_sum16w form that the device holds at once (0: unknown): a speed figure -- launch slices are sized so that
// a slice's workgroups start together -- never a correctness assumption (the single-pass Normalize's waits are bounded).
constexpr int kMaxDev = 16;   // the occupancy x CU caches below are kept per device (a process may drive several)
static inline int cur_dev() { int d = 0; return (hipGetDevice(&d) == hipSuccess && d >= 0 && d < kMaxDev) ? d : 0; }
int sum16w_resident_capacity(int nq, bool packed) {
    static int cap_all[kMaxDev][3];   // <4, true>, <2, true>, <2, false>; 0 = not asked yet (stored + 1)
    int* const cap = cap_all[cur_dev()];
    const int i = packed ? (nq == 4 ? 0 : 1) : 2;
    if (cap[i] > 0) return cap[i] - 1;
    {
        int per_cu = 0, dev = 0;
        hipDeviceProp_t prop;
        hipError_t e = i == 0 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_sum16w<4, true>, kThreads, 0)
                     : i == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_sum16w<2, true>, kThreads, 0)
                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_sum16w<2, false>, kThreads, 0);
        cap[i] = 1 + ((e == hipSuccess && hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
                          ? per_cu * prop.multiProcessorCount : 0);
    }
    return cap[i] - 1;
}
// k_norm1: the instantiation for (term mode, tiles per workgroup), its resident capacity, its launch
template <int TMODE, int TPW>
static int norm1_capacity_of() {
    static int cap_all[kMaxDev];   // per device; 0 = not asked yet (stored + 1)
    int& cap = cap_all[cur_dev()];
    if (cap == 0) {
        int per_cu = 0, dev = 0;
        hipDeviceProp_t prop;
        cap = 1 + ((hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_norm1<TMODE, TPW>, kThreads, 0) == hipSuccess && hipGetDevice(&dev) == hipSuccess &&
                    hipGetDeviceProperties(&prop, dev) == hipSuccess) ? per_cu * prop.multiProcessorCount : 0);
    }
    return cap - 1;
}
template <int TMODE>
static int norm1_capacity_mode(int tpw) {
    return tpw == 1 ? norm1_capacity_of<TMODE, 1>() : tpw == 2 ? norm1_capacity_of<TMODE, 2>() : norm1_capacity_of<TMODE, 4>();
}
static int norm1_capacity(uint32_t term_mode, int tpw) {
    switch (term_mode) {
        case TERMS_ALL_EDGE: return norm1_capacity_mode<TERMS_ALL_EDGE>(tpw);
        case TERMS_ALL_LOOP32: return norm1_capacity_mode<TERMS_ALL_LOOP32>(tpw);
        case TERMS_ALL_LOOP16: return norm1_capacity_mode<TERMS_ALL_LOOP16>(tpw);
        case TERMS_EDGE_FEW: return norm1_capacity_mode<TERMS_EDGE_FEW>(tpw);
        case TERMS_ADSR1: return norm1_capacity_mode<TERMS_ADSR1>(tpw);
        case TERMS_WITH_ADSR: return norm1_capacity_mode<TERMS_WITH_ADSR>(tpw);
        default: return norm1_capacity_mode<TERMS_MIXED>(tpw);
    }
}
int norm1_tiles_per_workgroup(uint32_t term_mode, uint32_t frames) {
    const uint32_t nt = (frames + kTileFrames - 1) / kTileFrames;
    for (int tpw : {1, 2, 4})
        if ((uint32_t)norm1_capacity(term_mode, tpw) >= (nt + (uint32_t)tpw - 1u) / (uint32_t)tpw) return tpw;
    return 0;
}
template <int TMODE, int TPW>
static void launch_norm1_of(const SumDesc* d, int n, uint32_t frames, uint32_t tag, hipStream_t s) {
    static const auto kern = &k_norm1<TMODE, TPW>;   // (a name without a comma for the launch macro)
    const uint32_t nt = (frames + kTileFrames - 1) / kTileFrames, gx = (nt + TPW - 1) / TPW;
    const int per = std::max(1, norm1_capacity_of<TMODE, TPW>() / (int)gx);   // (a slice must be resident at once)
    for (int o = 0; o < n; o += per)
        hipLaunchKernelGGL(kern, dim3(gx, std::min(per, n - o)), dim3(kThreads), 0, s, d + o, frames, nt, tag);
}
template <int TMODE>
static void launch_norm1_mode(const SumDesc* d, int n, uint32_t frames, int tpw, uint32_t tag, hipStream_t s) {
    if (tpw == 1) launch_norm1_of<TMODE, 1>(d, n, frames, tag, s);
    else if (tpw == 2) launch_norm1_of<TMODE, 2>(d, n, frames, tag, s);
    else launch_norm1_of<TMODE, 4>(d, n, frames, tag, s);
}
void launch_norm1(const SumDesc* d, int n, uint32_t frames, uint32_t term_mode, int tpw, uint32_t tag, hipStream_t s) {
    if (!n || !frames) return;
    switch (term_mode) {
        case TERMS_ALL_EDGE: launch_norm1_mode<TERMS_ALL_EDGE>(d, n, frames, tpw, tag, s); break;
        case TERMS_ALL_LOOP32: launch_norm1_mode<TERMS_ALL_LOOP32>(d, n, frames, tpw, tag, s); break;
        case TERMS_ALL_LOOP16: launch_norm1_mode<TERMS_ALL_LOOP16>(d, n, frames, tpw, tag, s); break;
        case TERMS_EDGE_FEW: launch_norm1_mode<TERMS_EDGE_FEW>(d, n, frames, tpw, tag, s); break;
        case TERMS_ADSR1: launch_norm1_mode<TERMS_ADSR1>(d, n, frames, tpw, tag, s); break;
        case TERMS_WITH_ADSR: launch_norm1_mode<TERMS_WITH_ADSR>(d, n, frames, tpw, tag, s); break;
        default: launch_norm1_mode<TERMS_MIXED>(d, n, frames, tpw, tag, s); break;
    }
}
void launch_sum(const SumDesc* d, int n, uint32_t frames, uint32_t bl, uint32_t term_mode, bool wide_ok, bool must_wide, uint32_t tag, hipStream_t s) {
    if (!n || !frames) return;
    const uint32_t tpb = (bl % kTileFrames == 0) ? bl / kTileFrames : 0;
    static const int env_nq = getenv("TD_FORCE_NQ") ? atoi(getenv("TD_FORCE_NQ")) : 0;   // tuning aid: 1 | 2 | 4
    const int forced_nq = (must_wide && env_nq == 1) ? 0 : env_nq;
    if (must_wide) wide_ok = true;
    static const int slice_env = getenv("TD_SUM_SLICE") ? atoi(getenv("TD_SUM_SLICE")) : 0;  // tuning aid: projects per launch slice
    switch (term_mode) {
        case TERMS_ALL_EDGE: TD_BATCHED(HIP_KERNEL_NAME(k_sum<TERMS_ALL_EDGE>), tiles(frames), kThreads, d, n, frames, bl, tpb); break;
        case TERMS_ALL_LOOP32:
            // (8 frames per lane: 0.183 -> 0.152 ms on config 2 with f32 samples; 16 per lane: 0.211 ms -- twice the
            // bytes per frame of the packed form, the whole grid's working set no longer sits in L2)
            if (wide_ok && frames >= 1800u * kTileFrames) {
                const uint32_t gx = (frames + kTileFrames * 2 - 1) / (kTileFrames * 2);
                const int per = must_wide ? std::max(1, sum16w_resident_capacity(2, false) / (int)gx) : n;
                for (int o = 0; o < n; o += per)
                    hipLaunchKernelGGL(k_sum32w_2, dim3(gx, std::min(per, n - o)), dim3(kThreads), 0, s, d + o, frames, tag);
            }
            else
                TD_BATCHED(HIP_KERNEL_NAME(k_sum<TERMS_ALL_LOOP32>), tiles(frames), kThreads, d, n, frames, bl, tpb);
            break;
        case TERMS_ALL_LOOP16:
            // wide_ok: every descriptor is a plain Sum or a Normalize whose reference block is the 1024-frame tile.
            // Long timelines take more frames per lane (more of the timeline resident per XCD -> more L2 hits among
            // concurrently running tiles); short ones keep the workgroup count up.
            // (measured on config-2-like renders of 3 .. 300 s, tools/nq_sweep.py: below ~1 800 tiles the narrow form wins --
            // a wide workgroup's own serial walk over the sources, ~55 us for 64 of them at 16 frames per lane, is then
            // the whole launch; 8 per lane pays from ~1 800 tiles, 16 per lane from ~2 600)
            // Batched launches (td_batch: n descriptors = n projects) go out in slices of about four workgroups per CU -- one
            // project per launch at config 2's size: the workgroups of a slice start together and walk the sources in step,
            // so every source table is gathered by all of them while it sits in L2 (hit rate ~75 %), exactly like the
            // single-project launch.  One grid over 64 projects instead staggers the workgroups over all 64 source indices:
            // 0.080 ms per project against 0.0685 sliced (2 / 3 / 4 projects per slice: 0.074 / 0.077 / 0.074; tools/ubench's
            // bare gather shows the same loss, so it is the access pattern, not the arithmetic).
            if (wide_ok && (forced_nq ? forced_nq == 4 : frames >= 2600u * kTileFrames)) {
                const uint32_t gx = (frames + kTileFrames * 4 - 1) / (kTileFrames * 4);
                int per = slice_env > 0 ? slice_env : (int)std::max(1u, (256u * 4u) / gx);
                if (must_wide) per = std::max(1, std::min(per, sum16w_resident_capacity(4, true) / (int)gx));   // (mode 5: a slice must be resident at once)
                for (int o = 0; o < n; o += per)
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sum16w<4>), dim3(gx, std::min(per, n - o)), dim3(kThreads), 0, s, d + o, frames, tag);
            } else if (wide_ok && (forced_nq ? forced_nq == 2 : frames >= 1800u * kTileFrames)) {
                const uint32_t gx = (frames + kTileFrames * 2 - 1) / (kTileFrames * 2);
                int per = slice_env > 0 ? slice_env : (int)std::max(1u, (256u * 4u) / gx);
                if (must_wide) per = std::max(1, std::min(per, sum16w_resident_capacity(2, true) / (int)gx));
                for (int o = 0; o < n; o += per)
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sum16w<2>), dim3(gx, std::min(per, n - o)), dim3(kThreads), 0, s, d + o, frames, tag);
            }
            else
                TD_BATCHED(HIP_KERNEL_NAME(k_sum<TERMS_ALL_LOOP16>), tiles(frames), kThreads, d, n, frames, bl, tpb);
            break;
        case TERMS_EDGE_FEW: TD_BATCHED(HIP_KERNEL_NAME(k_sum<TERMS_EDGE_FEW>), tiles(frames), kThreads, d, n, frames, bl, tpb); break;
        case TERMS_ADSR1: TD_BATCHED(HIP_KERNEL_NAME(k_sum<TERMS_ADSR1>), tiles(frames), kThreads, d, n, frames, bl, tpb); break;
        case TERMS_WITH_ADSR: TD_BATCHED(HIP_KERNEL_NAME(k_sum<TERMS_WITH_ADSR>), tiles(frames), kThreads, d, n, frames, bl, tpb); break;
        default: TD_BATCHED(HIP_KERNEL_NAME(k_sum<TERMS_MIXED>), tiles(frames), kThreads, d, n, frames, bl, tpb); break;
    }
}
void launch_scale(const ScaleDesc* d, int n, uint32_t frames, uint32_t bl, int is_scan, hipStream_t s) {
    if (!n || !frames) return;
    TD_BATCHED(k_scale, tiles(frames), kThreads, d, n, frames, bl, frames / bl, is_scan);
}
void launch_norm_fix(const SumDesc* d, int n, uint32_t frames, uint32_t bl, hipStream_t s) {
    if (!n || !frames) return;
    TD_BATCHED(k_norm_fix, std::min(tiles(frames), 512u), kThreads, d, n, frames, bl, frames / bl);
}
// (debugging aid, TD_DEBUG_SYNC & 16: does every workgroup of a wide grid -- every XCD, every L2 -- see the bytes an upload
// just put there?  Each workgroup sums the whole region with position weights and compares with the host's sum; a workgroup
// that disagrees leaves {1 + count, XCC_ID, first differing 256-byte segment by a second array of per-segment sums}.)
__global__ __launch_bounds__(kThreads) void k_debug_verify(const uint32_t* __restrict__ p, uint32_t n_words, const uint32_t* __restrict__ seg_sums,
                                                           uint32_t* __restrict__ report) {
    __shared__ uint32_t bad_seg;
    if (threadIdx.x == 0) bad_seg = 0xFFFFFFFFu;
    __syncthreads();
    const uint32_t n_seg = (n_words + 63u) / 64u;
    const uint32_t wave = threadIdx.x / 64u, lane = threadIdx.x % 64u;
    for (uint32_t sg = wave; sg < n_seg; sg += kThreads / 64u) {
        const uint32_t i = sg * 64u + lane;
        uint32_t v = i < n_words ? p[i] * (2u * lane + 1u) : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0u && v != seg_sums[sg]) atomicMin(&bad_seg, sg);
    }
    __syncthreads();
    if (threadIdx.x == 0 && bad_seg != 0xFFFFFFFFu) {
        uint32_t xcc = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const uint32_t k = atomicAdd(report, 1u);
        if (k < 15u) {
            report[4 + 4 * k] = blockIdx.x;
            report[5 + 4 * k] = xcc;
            report[6 + 4 * k] = bad_seg;
            report[7 + 4 * k] = p[bad_seg * 64u];
        }
    }
}
// the engine's two sines over an array (td_device_sinf: what tests/test_gpu_sine_exact.py compares with the host's sinf)
__global__ __launch_bounds__(kThreads) void k_sinf(const float* __restrict__ in, float* __restrict__ out, uint32_t n, int exact) {
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) out[i] = exact ? sin_glibc(in[i]) : sin_any(in[i]);
}
void launch_sinf(const float* in, float* out, uint32_t n, int exact, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_sinf, dim3(std::min<uint32_t>((n + kThreads - 1) / kThreads, 8192u)), dim3(kThreads), 0, s, in, out, n, exact);
}
void launch_debug_verify(const uint32_t* p, uint32_t n_words, const uint32_t* seg_sums, uint32_t* report, hipStream_t s) {
    if (!n_words) return;
    hipLaunchKernelGGL(k_debug_verify, dim3(512), dim3(kThreads), 0, s, p, n_words, seg_sums, report);
}
void launch_quantise(const QuantDesc* d, int n, uint32_t frames, hipStream_t s) {
    if (!n || !frames) return;
    TD_BATCHED(k_quantise, tiles(frames), kThreads, d, n, frames);
}
void launch_sample_loop(const LoopDesc* d, int n, uint32_t frames, hipStream_t s) {
    if (!n || !frames) return;
    TD_BATCHED(k_sample_loop, tiles(frames), kThreads, d, n, frames);
}
void launch_sample_multi(const MultiDesc* d, int n, uint32_t frames, hipStream_t s) {
    if (!n || !frames) return;
    TD_BATCHED(k_sample_multi, tiles(frames), kThreads, d, n, frames);
}
void launch_sample_lerp(const LerpDesc* d, int n, uint32_t frames, hipStream_t s) {
    if (!n || !frames) return;
    TD_BATCHED(k_sample_lerp, tiles(frames), kThreads, d, n, frames);
}
void launch_debug_sine(const SineDesc* d, int n, uint32_t frames, uint32_t bl, hipStream_t s) {
    if (!n || !frames) return;
    (void)bl;
    TD_BATCHED(k_debug_sine, tiles(frames), kThreads, d, n, frames);
}
void launch_synth(const SynthDesc* d, int n, uint32_t frames, bool affine, hipStream_t s) {
    if (!n || !frames) return;
    if (affine) TD_BATCHED(k_synth_affine, tiles(frames), kThreads, d, n, frames);
    else TD_BATCHED(k_synth, tiles(frames), kThreads, d, n, frames);
}
void launch_sampsyn(const SampsynDesc* d, int n, uint32_t frames, hipStream_t s) {
    if (!n || !frames) return;
    TD_BATCHED(k_sampsyn, tiles(frames), kThreads, d, n, frames);
}
void adsr_fill_run_consts(AdsrVDesc* d) {
    const AdsrConfD& c = d->conf;
    d->rcp[0] = 1.0 / (double)c.attack_sec;
    d->rcp[1] = 1.0 / (double)c.decay_sec;
    d->rcp[2] = 1.0 / (double)c.sustain_sec;
    d->rcp[3] = 1.0 / (double)c.release_sec;
    d->rcp[4] = 1.0 / (double)(float)d->sr;
    const float lo = fminf(fminf(c.std_vel, c.attack_vel), fminf(c.decay_vel, c.sustain_vel));
    d->tame = lo > -0.999f ? 1u : 0u;   // (NaN levels: not tame)
}
void launch_adsr_env(const AdsrVDesc* d, int n, uint32_t frames, hipStream_t s) {
    if (!n || !frames) return;
    const uint32_t per = (uint32_t)kThreads * (uint32_t)kEnvRun;
    TD_BATCHED(k_adsr_env, (frames + per - 1u) / per, kThreads, d, n, frames);
}
uint32_t source_part_grid(uint32_t kind, uint32_t frames) {
    if (kind == SRC_ENV) {
        const uint32_t per = (uint32_t)kThreads * (uint32_t)kEnvRun;
        return (frames + per - 1u) / per;
    }
    return tiles(frames);
}
int launch_sources(const SourceParts& P, uint32_t frames, void* zero, size_t zero_bytes, hipStream_t s) {
    if (!frames) return 1;
    SourceGrid G{};
    G.zero = (uint4*)zero;
    G.zero_n16 = (uint32_t)(zero_bytes / 16);
    uint32_t kinds = 0u, total = 0u;
    const int n[4] = {P.n_synth, P.n_sampsyn, P.n_lerp, P.n_env};
    for (uint32_t k = 0; k < 4u; ++k) {
        G.gx[k] = source_part_grid(k, frames);
        total += G.gx[k] * (uint32_t)n[k];
        G.end[k] = total;
        if (n[k]) kinds |= 1u << k;
    }
    if (!total) return 1;
    constexpr uint32_t S = 1u << SRC_SYNTH_AFFINE, Y = 1u << SRC_SAMPSYN, L = 1u << SRC_LERP, E = 1u << SRC_ENV;
    // (the combinations the engine forms: an affine Synth launch keeps its six workgroups per CU; everything else is compiled
    // with all the block functions it may need -- 108 registers, k_sample_lerp's: capped to five workgroups per CU it spills 11
    // and loses 2 us on config 4, to six 26 us)
    if (kinds == (S | E)) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sources<S | E, 6>), dim3(total), dim3(kThreads), 0, s, P.synth, P.sampsyn, P.lerp, P.env, G, frames);
    else if (!(kinds & S)) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sources<Y | L | E, 4>), dim3(total), dim3(kThreads), 0, s, P.synth, P.sampsyn, P.lerp, P.env, G, frames);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sources<S | Y | L | E, 4>), dim3(total), dim3(kThreads), 0, s, P.synth, P.sampsyn, P.lerp, P.env, G, frames);
    return 1;
}
void launch_adsr(const AdsrVDesc* d, int n, uint32_t frames, uint32_t term_mode, hipStream_t s) {
    if (!n || !frames) return;
    switch (term_mode) {
        case TERMS_ALL_EDGE: TD_BATCHED(HIP_KERNEL_NAME(k_adsr<TERMS_ALL_EDGE>), tiles(frames), kThreads, d, n, frames); break;
        case TERMS_ALL_LOOP32: TD_BATCHED(HIP_KERNEL_NAME(k_adsr<TERMS_ALL_LOOP32>), tiles(frames), kThreads, d, n, frames); break;
        case TERMS_EDGE_FEW: TD_BATCHED(HIP_KERNEL_NAME(k_adsr<TERMS_EDGE_FEW>), tiles(frames), kThreads, d, n, frames); break;
        default: TD_BATCHED(HIP_KERNEL_NAME(k_adsr<TERMS_MIXED>), tiles(frames), kThreads, d, n, frames); break;
    }
}
void launch_band_pass(const BandDesc* d, int n, uint32_t frames, hipStream_t s) {   // vertex index in grid.x
    if (!n || !frames) return;
    hipLaunchKernelGGL(k_band_pass, dim3(n), dim3(kThreads), 0, s, d, frames);
}
void launch_band_spec(const BandSpecDesc* d, int n, uint32_t frames, uint32_t max_nseg, hipStream_t s) {
    if (!n || !frames) return;
    TD_BATCHED(k_band_spec, (max_nseg + kThreads / 4 - 1) / (kThreads / 4), kThreads, d, n, frames);
}
void launch_band_fix(const BandSpecDesc* d, int n, uint32_t frames, uint32_t max_nseg, hipStream_t s) {    // (slice + helpers, vertex)
    if (n <= 0) return;
    const uint32_t G = std::max(1u, std::min(16u, max_nseg / 512u));
    // helpers for the parked stretches' output (k_band_fix): enough workgroups to stream a long timeline, none for a block pull
    static const int h_env = getenv("TD_FILL_HELPERS") ? atoi(getenv("TD_FILL_HELPERS")) : -1;   // (experiments)
    const uint32_t H = h_env >= 0 ? (uint32_t)h_env : std::min(240u, tiles(frames) / 8u);
    for (int o = 0; o < n; o += kMaxGridY)
        hipLaunchKernelGGL(k_band_fix, dim3(G + H, std::min(n - o, kMaxGridY)), dim3(kFixThreads), 0, s, d + o, frames, G);
}
int band_scan_resident_capacity(int nf);
template <int MODE>
static void launch_band_scan_mode(const BandScanDesc* d, int n, uint32_t frames, uint32_t gx, int nf, hipStream_t s) {
    static const auto k16 = &k_band_scan<MODE, 16>;
    static const auto k8 = &k_band_scan<MODE, 8>;
    // (the vertices of one launch must be resident TOGETHER: the gather at the kernel's end waits for every earlier tile of
    // its vertex without bound -- a batch's vertices go out in slices that fit)
    const int per = std::max(1, std::min(band_scan_resident_capacity(nf) / (int)std::max(gx, 1u), kMaxGridY));
    for (int o = 0; o < n; o += per) {
        if (nf == 16) hipLaunchKernelGGL(k16, dim3(gx, (uint32_t)std::min(per, n - o)), dim3(kThreads), 0, s, d + o, frames);
        else hipLaunchKernelGGL(k8, dim3(gx, (uint32_t)std::min(per, n - o)), dim3(kThreads), 0, s, d + o, frames);
    }
}
// Workgroups of k_band_scan the device holds at once (0: unknown) -- the all-earlier-tiles gather at its end waits without
// bound, so the engine only lets a vertex take this kernel when its grid fits (the smallest occupancy over the instantiations)
template <int NF>
static int band_scan_min_blocks_per_cu() {   // over the term-mode instantiations launch_band_scan can pick
    int per_cu = 1 << 30, v = 0;
    const void* ks[] = {(const void*)k_band_scan<TERMS_EDGE_FEW, NF>, (const void*)k_band_scan<TERMS_ALL_EDGE, NF>, (const void*)k_band_scan<TERMS_ADSR1, NF>,
                        (const void*)k_band_scan<TERMS_WITH_ADSR, NF>, (const void*)k_band_scan<TERMS_MIXED, NF>};
    for (const void* k : ks) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, k, kThreads, 0) != hipSuccess) return 0;
        per_cu = std::min(per_cu, v);
    }
    return per_cu;
}
int band_scan_resident_capacity(int nf) {
    static int cap_all[kMaxDev][2];   // per device; 0 = not asked yet (stored + 1)
    int* const cap = cap_all[cur_dev()];
    const int i = nf == 16 ? 0 : 1;
    if (cap[i] == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        const int per_cu = nf == 16 ? band_scan_min_blocks_per_cu<16>() : band_scan_min_blocks_per_cu<8>();
        cap[i] = 1 + ((hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? per_cu * prop.multiProcessorCount : 0);
    }
    return cap[i] - 1;
}
void launch_band_scan(const BandScanDesc* d, int n, uint32_t frames, uint32_t term_mode, int nf, hipStream_t s) {
    if (!n || !frames) return;
    const uint32_t tile = band_scan_tile_frames(nf), gx = (frames + tile - 1u) / tile;
    switch (term_mode) {
        case TERMS_EDGE_FEW: launch_band_scan_mode<TERMS_EDGE_FEW>(d, n, frames, gx, nf, s); break;
        case TERMS_ALL_EDGE: launch_band_scan_mode<TERMS_ALL_EDGE>(d, n, frames, gx, nf, s); break;
        case TERMS_ADSR1: launch_band_scan_mode<TERMS_ADSR1>(d, n, frames, gx, nf, s); break;
        case TERMS_WITH_ADSR: launch_band_scan_mode<TERMS_WITH_ADSR>(d, n, frames, gx, nf, s); break;
        default: launch_band_scan_mode<TERMS_MIXED>(d, n, frames, gx, nf, s); break;
    }
}
void launch_band_chain(const BandScanDesc* d, int n, uint32_t frames, uint32_t term_mode, bool guarded, hipStream_t s) {
    if (!n || !frames) return;
    const uint32_t tile = band_scan_tile_frames(16), gx = (frames + tile - 1u) / tile;
    void (*k)(const BandScanDesc*, uint32_t, uint32_t) = nullptr;
    switch (term_mode) {
        case TERMS_EDGE_FEW: k = guarded ? k_band_chain<TERMS_EDGE_FEW, true> : k_band_chain<TERMS_EDGE_FEW, false>; break;
        case TERMS_ALL_EDGE: k = guarded ? k_band_chain<TERMS_ALL_EDGE, true> : k_band_chain<TERMS_ALL_EDGE, false>; break;
        case TERMS_ADSR1: k = guarded ? k_band_chain<TERMS_ADSR1, true> : k_band_chain<TERMS_ADSR1, false>; break;
        case TERMS_WITH_ADSR: k = guarded ? k_band_chain<TERMS_WITH_ADSR, true> : k_band_chain<TERMS_WITH_ADSR, false>; break;
        default: k = guarded ? k_band_chain<TERMS_MIXED, true> : k_band_chain<TERMS_MIXED, false>; break;
    }
    if (gx <= (uint32_t)kMaxGridY) {   // chains along x (dispatched round-robin), tiles along y
        hipLaunchKernelGGL(k, dim3((uint32_t)n, gx), dim3(kThreads), 0, s, d, frames, 1u);
    } else {
        for (int o = 0; o < n; o += kMaxGridY)
            hipLaunchKernelGGL(k, dim3(gx, (uint32_t)std::min(n - o, kMaxGridY)), dim3(kThreads), 0, s, d + o, frames, 0u);
    }
}
#undef TD_BATCHED
static inline uint32_t grid_for(uint32_t n) { return max(1u, min((n + kThreads - 1) / kThreads, 2048u)); }
void launch_absmax(const float* v, uint32_t n, float* out, hipStream_t s) {
    (void)hipMemsetAsync(out, 0, sizeof(float), s);
    if (n) hipLaunchKernelGGL(k_absmax_atomic, dim3(grid_for(n)), dim3(kThreads), 0, s, v, n, out);
}
__global__ __launch_bounds__(kThreads) void k_peak_table(const float* const* __restrict__ src, float* __restrict__ table,
                                                         uint32_t n_total, uint32_t n_own, uint32_t first, uint32_t stride) {
    const uint32_t j = blockIdx.x * kThreads + threadIdx.x;
    if (j >= n_total) return;
    float v = 0.0f;
    if (j >= first && (j - first) % stride == 0u && (j - first) / stride < n_own) v = *src[(j - first) / stride];
    table[j] = v;
}
void launch_peak_table(const float* const* src, float* table, uint32_t n_total, uint32_t n_own, uint32_t first, uint32_t stride,
                       hipStream_t s) {
    if (n_total) hipLaunchKernelGGL(k_peak_table, dim3((n_total + kThreads - 1) / kThreads), dim3(kThreads), 0, s, src, table, n_total, n_own, first, stride);
}
void launch_resample(const ResampleDesc& d, hipStream_t s) {
    if (!d.nout) return;
    const uint64_t blocks = (d.nout + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(k_resample, dim3((uint32_t)(blocks < 65535u * 16u ? blocks : 65535u * 16u)), dim3(kThreads), 0, s, d);
}
void launch_pcm_decode(const uint8_t* raw, float* linear, uint32_t n, uint32_t format, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_pcm_decode, dim3(grid_for(n)), dim3(kThreads), 0, s, raw, linear, n, format);
}
void launch_sample_split(const float* linear, uint32_t channels, uint32_t src_l, uint32_t src_r, float* l, float* r, uint32_t nl,
                         uint32_t nr, hipStream_t s) {
    const uint32_t n = nl > nr ? nl : nr;
    if (n) hipLaunchKernelGGL(k_sample_split, dim3(grid_for(n)), dim3(kThreads), 0, s, linear, channels, src_l, src_r, l, r, nl, nr);
}
void launch_abs_sum_serial(const float* v, uint32_t n, float* out, hipStream_t s) {
    hipLaunchKernelGGL(k_abs_sum_serial, dim3(1), dim3(kThreads), 0, s, v, n, out);
}
void launch_add_planar(const float* a, const float* b, float* out, uint32_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_add_planar, dim3(grid_for(n)), dim3(kThreads), 0, s, a, b, out, n);
}
__global__ __launch_bounds__(kThreads) void k_sample_pack16(const float* __restrict__ l, const float* __restrict__ r,
                                                            uint32_t* __restrict__ packed, uint32_t n, uint32_t* not_int16) {
    // word i = frame i % n for i < roundup(n + 3, 4): the loop plus its first frames again, so that any four
    // consecutive loop frames (wrap included) are four consecutive words
    const uint32_t total = (n + 15u + 3u) & ~3u;
    bool bad = false;
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < total; i += gridDim.x * kThreads) {
        const uint32_t f = i % n;
        const float a = l[f], b = r[f];
        const int ia = (int)fminf(fmaxf(a, -32768.0f), 32767.0f), ib = (int)fminf(fmaxf(b, -32768.0f), 32767.0f);
        bad = bad || (float)ia != a || (float)ib != b;   // NaN, fractions and out-of-range values all fail here
        packed[i] = ((uint32_t)ia & 0xFFFFu) | ((uint32_t)ib << 16);
    }
    if (__any(bad ? 1 : 0) && (threadIdx.x & 63) == 0) atomicOr(not_int16, 1u);
}
void launch_sample_pack16(const float* l, const float* r, uint32_t* packed, uint32_t n, uint32_t* not_int16, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_sample_pack16, dim3(grid_for((n + 18u) & ~3u)), dim3(kThreads), 0, s, l, r, packed, n, not_int16);
}
void launch_sample_pack(const float* l, const float* r, const float* max_l, const float* max_r, float2* frames, uint32_t n,
                        hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_sample_pack, dim3(grid_for(n + 15u)), dim3(kThreads), 0, s, l, r, max_l, max_r, frames, n);
}

}  // namespace tdk
