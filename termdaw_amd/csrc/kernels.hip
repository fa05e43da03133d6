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

// Two consecutive frames starting at frame m (m even).  Buffers are padded to an even frame count, so a
// pair whose first frame is valid may always be accessed as one 16-byte word.
TD_DEV float4 load_pair(const float2* p, uint32_t m, uint32_t M) {
    if (m + 1 < M) return *reinterpret_cast<const float4*>(p + m);
    if (m < M) { const float2 a = p[m]; return make_float4(a.x, a.y, 0.f, 0.f); }   // odd tail: pad reads as 0
    return make_float4(0.f, 0.f, 0.f, 0.f);
}
TD_DEV void store_pair(float2* p, uint32_t m, uint32_t M, float4 v) {
    if (m < M) *reinterpret_cast<float4*>(p + m) = v;
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
        if (m + 1 < M) *reinterpret_cast<uint2*>(o) = make_uint2(w0, w1);
        else o[0] = w0;
    } else {
        int32_t* o = reinterpret_cast<int32_t*>(pcm) + 2 * (size_t)m;
        if (m + 1 < M) *reinterpret_cast<int4*>(o) = make_int4(quant32(v.x, amp), quant32(v.y, amp),
                                                              quant32(v.z, amp), quant32(v.w, amp));
        else *reinterpret_cast<int2*>(o) = make_int2(quant32(v.x, amp), quant32(v.y, amp));
    }
}

TD_DEV float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

template <typename IDX>
TD_DEV float4 loop_pair(const float2* s, IDX len, IDX idx) {
    float2 a = s[idx];
    IDX i1 = idx + 1;
    if (i1 == len) i1 = 0;
    float2 b = s[i1];
    return make_float4(a.x, a.y, b.x, b.y);
}
// x mod len for x < 2^32 with magic = floor(2^32 / len): q underestimates x / len by at most 1
TD_DEV uint32_t barrett_mod(uint32_t x, uint32_t len, uint32_t magic) {
    const uint32_t r = x - __umulhi(x, magic) * len;
    return r >= len ? r - len : r;
}
// sample_loop_gen for one frame pair (extensions.rs:337-338), 32-bit cursor.  The pair is one aligned
// 16-byte load when the loop position is even and does not wrap inside the pair (the parity is the same
// for a whole tile of a given source), else two 8-byte loads.
TD_DEV float4 loop_pair32(const float2* s, uint32_t len, uint32_t magic, uint32_t x) {
    const uint32_t idx = barrett_mod(x, len, magic);
    if (((idx & 1u) == 0u) && idx + 1u < len) return *reinterpret_cast<const float4*>(s + idx);
    return loop_pair<uint32_t>(s, len, idx);
}
// stand-alone sample_loop vertex (k_sample_loop): wave-uniform choice of the 32-bit form
TD_DEV float4 gather_loop_pair(const float2* s, uint64_t len64, uint64_t t0, uint32_t m, uint32_t M) {
    if (len64 <= 0xFFFFFFFFull && t0 + M + kTileFrames <= 0xFFFFFFFFull) {
        const uint32_t len = (uint32_t)len64;
        return loop_pair<uint32_t>(s, len, ((uint32_t)t0 + m) % len);
    }
    return loop_pair<uint64_t>(s, len64, (t0 + m) % len64);
}

TD_DEV float4 zero_tail(float4 v, uint32_t m, uint32_t M) {   // frames at or beyond M contribute nothing
    if (m + 1 >= M) { v.z = 0.f; v.w = 0.f; }
    if (m >= M) { v.x = 0.f; v.y = 0.f; }
    return v;
}

// value of one input term for the frame pair starting at m (generic form)
TD_DEV float4 term_pair(const InTerm& t, uint32_t m, uint32_t M) {
    if (t.kind == 0) return load_pair(t.p, m, M);
    float4 v = t.kind == 1 ? loop_pair32(t.p, (uint32_t)t.len, t.magic, (uint32_t)t.t0 + m)
                           : loop_pair<uint64_t>(t.p, t.len, (t.t0 + m) % t.len);
    return zero_tail(epilogue4(v, t.pg), m, M);
}
TD_DEV float4 loop_term_pair(const InTerm& t, uint32_t m, uint32_t M) {
    return zero_tail(epilogue4(loop_pair32(t.p, (uint32_t)t.len, t.magic, (uint32_t)t.t0 + m), t.pg), m, M);
}

// sum_inputs (extensions.rs:310-319): zero, then += each input in edge order.  Terms are fetched four
// (edge buffers: eight) at a time so that 8-16 x 16 B loads are in flight per lane before the first add;
// the adds themselves stay strictly sequential per element.
template <int MODE>
TD_DEV void sum_terms(const InTerm* __restrict__ ins, uint32_t k, uint32_t m0, uint32_t m1, uint32_t M,
                      float4& a0, float4& a1) {
    uint32_t j = 0;
    if (MODE == TERMS_ALL_EDGE) {
        for (; j + 8 <= k; j += 8) {
            float4 x0[8], x1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const float2* p = ins[j + u].p; x0[u] = load_pair(p, m0, M); x1[u] = load_pair(p, m1, M); }
#pragma unroll
            for (int u = 0; u < 8; ++u) { a0 = add4(a0, x0[u]); a1 = add4(a1, x1[u]); }
        }
    }
    for (; j + 4 <= k; j += 4) {
        float4 x0[4], x1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE == TERMS_ALL_EDGE) { const float2* p = ins[j + u].p; x0[u] = load_pair(p, m0, M); x1[u] = load_pair(p, m1, M); }
            else if (MODE == TERMS_ALL_LOOP32) { x0[u] = loop_term_pair(ins[j + u], m0, M); x1[u] = loop_term_pair(ins[j + u], m1, M); }
            else { x0[u] = term_pair(ins[j + u], m0, M); x1[u] = term_pair(ins[j + u], m1, M); }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { a0 = add4(a0, x0[u]); a1 = add4(a1, x1[u]); }
    }
    for (; j < k; ++j) {
        a0 = add4(a0, term_pair(ins[j], m0, M));
        a1 = add4(a1, term_pair(ins[j], m1, M));
    }
}
TD_DEV void sum_inputs_pairs(const InTerm* ins, uint32_t k, uint32_t term_mode, uint32_t m0, uint32_t m1, uint32_t M,
                             float4& a0, float4& a1) {
    a0 = make_float4(0.f, 0.f, 0.f, 0.f);
    a1 = a0;
    if (term_mode == TERMS_ALL_EDGE) sum_terms<TERMS_ALL_EDGE>(ins, k, m0, m1, M, a0, a1);
    else if (term_mode == TERMS_ALL_LOOP32) sum_terms<TERMS_ALL_LOOP32>(ins, k, m0, m1, M, a0, a1);
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
__global__ __launch_bounds__(kThreads) void k_sum(const SumDesc* __restrict__ descs, uint32_t M, uint32_t bl,
                                                  uint32_t tiles_per_block) {
    const SumDesc& d = descs[blockIdx.y];
    const uint32_t m0 = blockIdx.x * kTileFrames + 2 * threadIdx.x;
    const uint32_t m1 = m0 + kTileFrames / 2;
    float4 a0, a1;
    sum_inputs_pairs(d.ins, d.k, d.term_mode, m0, m1, M, a0, a1);
    if (d.mode == 0) {
        store_pair(d.out, m0, M, epilogue4(a0, d.pg));
        store_pair(d.out, m1, M, epilogue4(a1, d.pg));
        return;
    }
    store_pair(d.out, m0, M, a0);
    store_pair(d.out, m1, M, a1);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        d.init_copy[0] = d.use_init ? d.init_max : d.state->max;
        d.init_copy[1] = d.state->scan_max;
    }
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
        }
    } else {
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
    // max of the peaks of all blocks before this tile's first block (identity 0: peaks are >= 0, never NaN)
    __shared__ float wmax[kThreads / 64];
    float p = 0.0f;
    for (uint32_t b = threadIdx.x; b < b_lo; b += kThreads) p = fmaxf(d.peaks[b], p);
    p = wave_max(p);
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = p;
    __syncthreads();
    const float before = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    const float r_stale = 1.0f / init;
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
            float4 v = load_pair(d.buf, m, M);
            const float r0 = rscale_of(m);
            const float r1 = (m + 1 < M) ? rscale_of(m + 1) : r0;
            v = epilogue4(make_float4(v.x * r0, v.y * r0, v.z * r1, v.w * r1), d.pg);
            store_pair(d.buf, m, M, v);
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
    const float4 v0 = gather_loop_pair(d.sample, d.len, d.t0, m0, M);
    const float4 v1 = gather_loop_pair(d.sample, d.len, d.t0, m1, M);
    store_pair(d.out, m0, M, epilogue4(v0, d.pg));
    store_pair(d.out, m1, M, epilogue4(v1, d.pg));
}

// ------------------------------------------------------------------------------------------------
// k_sample_multi (extensions.rs:344-381)
// ------------------------------------------------------------------------------------------------
TD_DEV float2 multi_frame(const MultiDesc& d, int64_t m) {
    // live voices: origin in (m - len, m]; hits are sorted by origin (onset order = deque order)
    const int64_t lo_key = m - (int64_t)d.len;
    uint32_t lo = 0, hi = d.n_hits;
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (d.hits[mid].origin > lo_key) hi = mid; else lo = mid + 1; }
    const uint32_t first = lo;
    hi = d.n_hits;
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (d.hits[mid].origin > m) hi = mid; else lo = mid + 1; }
    const uint32_t last = lo;
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
    uint32_t lo = 0, hi = d.n_hits;   // number of entries with key <= m (>= 2: the carried pair)
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (d.hits[mid].key > m) hi = mid; else lo = mid + 1; }
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
__global__ __launch_bounds__(kThreads) void k_sample_lerp(const LerpDesc* __restrict__ descs, uint32_t M) {
    const LerpDesc& d = descs[blockIdx.y];
    const uint32_t m0 = blockIdx.x * kTileFrames + 2 * threadIdx.x;
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

// envelope math: adsr_math.h (shared with the host event compiler)

TD_DEV uint32_t find_interval(const uint32_t* __restrict__ s, uint32_t n, uint32_t m) {
    uint32_t lo = 0, hi = n;   // last i with s[i] <= m ; s[0] == 0
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (s[mid] <= m) lo = mid; else hi = mid; }
    return lo;
}

constexpr float kPi = 3.14159274101257324f;   // core::f32::consts::PI

// ------------------------------------------------------------------------------------------------
// k_debug_sine (extensions.rs:423-457)
// ------------------------------------------------------------------------------------------------
TD_DEV float sine_frame(const SineDesc& d, uint32_t m) {
    const uint32_t it = find_interval(d.tab.istart, d.tab.n_int, m);
    const uint32_t v0 = d.tab.ivoff[it], v1 = d.tab.ivoff[it + 1];
    const float time = (float)(d.t0 + m) / (float)d.sr;
    float acc = 0.0f;
    for (uint32_t v = v0; v < v1; ++v) {
        const float4 nv = d.tab.voices[v];   // (hz, vel)
        acc += sinf(time * nv.x * 2.0f * kPi) * nv.y;
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
TD_DEV float synth_frame(const SynthDesc& d, uint32_t m) {
    const uint32_t it = find_interval(d.tab.istart, d.tab.n_int, m);
    const uint32_t v0 = d.tab.ivoff[it], v1 = d.tab.ivoff[it + 1];
    const float time = (float)(d.t0 + m) / (float)d.sr;
    const float off = (float)(m % d.bl) / (float)d.sr;
    float acc = 0.0f;
    for (uint32_t v = v0; v < v1; ++v) {
        const float4 n = d.tab.voices[v];   // (hz, vel, env_t, rel_t)
        const float hz = n.x, vel = n.y, rel_t = n.w;
        const float env_time = n.z + off;
        float s = 0.0f;
        float sn = 0.0f;
        if (d.square.volume > 0.0f || d.topflat.volume > 0.0f) sn = sinf(time * hz * 2.0f * kPi);
        if (d.square.volume > 0.0f) {
            const float z = d.square.param;
            const float osc = fminf(fmaxf(sn, -z), z) * (1.0f / z);
            const float env = rel_t == 0.0f ? apply_ads(d.square.adsr, env_time) : apply_r_rt(d.square.adsr, env_time, rel_t);
            s += osc * vel * env * d.square.volume;
        }
        if (d.topflat.volume > 0.0f) {
            const float z = d.topflat.param;
            const float osc = (fminf(sn, z) + ((1.0f - z) / 2.0f)) * (2.0f / (1.0f + z));
            const float env = rel_t == 0.0f ? apply_ads(d.topflat.adsr, env_time) : apply_r_rt(d.topflat.adsr, env_time, rel_t);
            s += osc * vel * env * d.topflat.volume;
        }
        if (d.triangle.volume > 0.0f) {
            const float th = time * hz;
            const float osc = 4.0f * fabsf(th - floorf(th + 0.5f)) - 1.0f;
            const float env = rel_t == 0.0f ? apply_ads(d.triangle.adsr, env_time) : apply_r_rt(d.triangle.adsr, env_time, rel_t);
            s += osc * vel * env * d.triangle.volume;
        }
        s *= d.osc_amp_multiplier;
        acc += s;
    }
    return acc;
}
__global__ __launch_bounds__(kThreads) void k_synth(const SynthDesc* __restrict__ descs, uint32_t M) {
    const SynthDesc& d = descs[blockIdx.y];
    const uint32_t m0 = blockIdx.x * kTileFrames + 2 * threadIdx.x;
    const uint32_t m1 = m0 + kTileFrames / 2;
    if (m0 < M) {
        float a = synth_frame(d, m0), b = (m0 + 1 < M) ? synth_frame(d, m0 + 1) : 0.0f;
        store_pair(d.out, m0, M, epilogue4(make_float4(a, a, b, b), d.pg));
    }
    if (m1 < M) {
        float a = synth_frame(d, m1), b = (m1 + 1 < M) ? synth_frame(d, m1 + 1) : 0.0f;
        store_pair(d.out, m1, M, epilogue4(make_float4(a, a, b, b), d.pg));
    }
}

// ------------------------------------------------------------------------------------------------
// k_adsr: envelope-follower vertex (extensions.rs:593-651)
// ------------------------------------------------------------------------------------------------
TD_DEV float2 adsr_frame(const AdsrVDesc& d, uint32_t m, float2 x) {
    const uint32_t it = find_interval(d.tab.istart, d.tab.n_int, m);
    const float4 p = d.tab.voices[2 * it], g = d.tab.voices[2 * it + 1];   // (t_off, vel, release_val, skip)
    if (p.w != 0.0f) return x;   // extensions.rs:632-635: `continue` leaves this frame untouched
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
    const float adsr_vel = fmaxf(pvel, gvel) * maxmul + fminf(pvel, gvel) * minmul;
    const float vel = lerpf(1.0f, adsr_vel, d.wet);
    return make_float2(x.x * vel, x.y * vel);
}
__global__ __launch_bounds__(kThreads) void k_adsr(const AdsrVDesc* __restrict__ descs, uint32_t M) {
    const AdsrVDesc& d = descs[blockIdx.y];
    const uint32_t m0 = blockIdx.x * kTileFrames + 2 * threadIdx.x;
    const uint32_t m1 = m0 + kTileFrames / 2;
    float4 a0, a1;
    sum_inputs_pairs(d.ins, d.k, d.term_mode, m0, m1, M, a0, a1);
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
    bool first = d.state->first != 0;
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

// single-float absolute max of a small table (per-project peak)
__global__ __launch_bounds__(kThreads) void k_absmax(const float* __restrict__ v, uint32_t n, float* out) {
    float m = 0.0f;
    for (uint32_t i = threadIdx.x; i < n; i += kThreads) m = fmaxf(m, fabsf(v[i]));
    m = wave_max(m);
    __shared__ float w[kThreads / 64];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) *out = fmaxf(fmaxf(w[0], w[1]), fmaxf(w[2], w[3]));
}

// ------------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------------
static inline uint32_t tiles(uint32_t frames) { return (frames + kTileFrames - 1) / kTileFrames; }

void launch_sum(const SumDesc* d, int n, uint32_t frames, uint32_t bl, hipStream_t s) {
    if (!n || !frames) return;
    uint32_t tpb = (bl % kTileFrames == 0) ? bl / kTileFrames : 0;
    hipLaunchKernelGGL(k_sum, dim3(tiles(frames), n), dim3(kThreads), 0, s, d, frames, bl, tpb);
}
void launch_scale(const ScaleDesc* d, int n, uint32_t frames, uint32_t bl, int is_scan, hipStream_t s) {
    if (!n || !frames) return;
    hipLaunchKernelGGL(k_scale, dim3(tiles(frames), n), dim3(kThreads), 0, s, d, frames, bl, frames / bl, is_scan);
}
void launch_quantise(const QuantDesc* d, int n, uint32_t frames, hipStream_t s) {
    if (!n || !frames) return;
    hipLaunchKernelGGL(k_quantise, dim3(tiles(frames), n), dim3(kThreads), 0, s, d, frames);
}
void launch_sample_loop(const LoopDesc* d, int n, uint32_t frames, hipStream_t s) {
    if (!n || !frames) return;
    hipLaunchKernelGGL(k_sample_loop, dim3(tiles(frames), n), dim3(kThreads), 0, s, d, frames);
}
void launch_sample_multi(const MultiDesc* d, int n, uint32_t frames, hipStream_t s) {
    if (!n || !frames) return;
    hipLaunchKernelGGL(k_sample_multi, dim3(tiles(frames), n), dim3(kThreads), 0, s, d, frames);
}
void launch_sample_lerp(const LerpDesc* d, int n, uint32_t frames, hipStream_t s) {
    if (!n || !frames) return;
    hipLaunchKernelGGL(k_sample_lerp, dim3(tiles(frames), n), dim3(kThreads), 0, s, d, frames);
}
void launch_debug_sine(const SineDesc* d, int n, uint32_t frames, uint32_t bl, hipStream_t s) {
    if (!n || !frames) return;
    (void)bl;
    hipLaunchKernelGGL(k_debug_sine, dim3(tiles(frames), n), dim3(kThreads), 0, s, d, frames);
}
void launch_synth(const SynthDesc* d, int n, uint32_t frames, hipStream_t s) {
    if (!n || !frames) return;
    hipLaunchKernelGGL(k_synth, dim3(tiles(frames), n), dim3(kThreads), 0, s, d, frames);
}
void launch_adsr(const AdsrVDesc* d, int n, uint32_t frames, hipStream_t s) {
    if (!n || !frames) return;
    hipLaunchKernelGGL(k_adsr, dim3(tiles(frames), n), dim3(kThreads), 0, s, d, frames);
}
void launch_band_pass(const BandDesc* d, int n, uint32_t frames, hipStream_t s) {
    if (!n || !frames) return;
    hipLaunchKernelGGL(k_band_pass, dim3(n), dim3(kThreads), 0, s, d, frames);
}
void launch_absmax(const float* peaks, uint32_t n, float* out, hipStream_t s) {
    hipLaunchKernelGGL(k_absmax, dim3(1), dim3(kThreads), 0, s, peaks, n, out);
}

}  // namespace tdk
