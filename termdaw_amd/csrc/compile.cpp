// compile.cpp -- the HOST COMPILER of the render engine: one chunk of whole reference blocks -> event tables, kernel
// descriptors and the launch list (compile_chunk), from the graph, the banks and the FlowwBank cursors alone.  Nothing in
// this file calls the HIP runtime: device memory reaches it as addresses handed out by three functions of engine.cpp
// (compile.h: take_buffer / ensure_buffers for edge buffers, upload_tables for a vertex' event tables), everything it
// writes goes into the plain-memory staging arena of the ChunkBuild (offsets + fix-up lists that submit_chunk patches once
// the device addresses are known).  tests/test_compile_asan.py builds it with g++ -fsanitize=address,undefined.
#include "compile.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <limits>

using namespace tdk;

namespace tde {

const char* const kFamilyName[F_COUNT] = {"k_sample_loop", "k_sample_multi", "k_sample_lerp", "k_debug_sine",
                                           "k_synth",       "k_sampsyn", "k_adsr_env", "k_sine_probe", "k_sum",          "k_scale",       "k_norm_fix",
                                           "k_adsr",        "k_band_pass",    "k_band_spec", "k_band_fix", "k_band_fill", "k_band_scan", "k_quantise", "k_band_audit", "k_sources"};


// ------------------------------------------------------------------------------------------------
// event pulls, one reference block at a time
// ------------------------------------------------------------------------------------------------
// get_block_drum (floww.rs:99-121) called for offsets 0..bl-1: events before the wanted frame are
// skipped, only the FIRST on-event (vel > 0.001) of a frame is delivered, note-offs are dropped.
template <class Hit>
static void drum_block(const std::vector<td_event>& ev, const td_flowwbank& fb, size_t frame, size_t start,
                       size_t bl, Hit on_hit) {
    size_t p = start, i = 0;
    while (p < ev.size() && i < bl) {
        const size_t f = fb.frame_of(ev[p]);
        const size_t target = frame + i;
        if (f < target) { ++p; continue; }
        if (f == target) {
            const td_event e = ev[p++];
            if (e.vel > 0.001f) {
                on_hit(i, e.note, e.vel);
                ++i;   // the next pull is for the next offset
            }
            continue;
        }
        if (f - frame >= bl) break;
        i = f - frame;
    }
}
// get_block_simple (floww.rs:124-141): ALL events of the wanted frame, in order; never skips a stale
// event (an event behind the cursor blocks the rest of the block).
template <class Ev, class Done>
static void simple_block(const std::vector<td_event>& ev, const td_flowwbank& fb, size_t frame, size_t start,
                         size_t bl, Ev on_event, Done offset_done) {
    size_t p = start, i = 0;
    bool any = false;
    while (p < ev.size()) {
        const size_t f = fb.frame_of(ev[p]);
        const size_t target = frame + i;
        if (f == target) {
            const td_event e = ev[p++];
            on_event(i, e.vel > 0.001f, e.note, e.vel);
            any = true;
            continue;
        }
        if (any) { offset_done(i); any = false; }
        if (f < target || f - frame >= bl) break;
        i = f - frame;
    }
    if (any) offset_done(i);
}


struct IntervalBuilder {
    std::vector<uint32_t> istart, ivoff;
    std::vector<float4> voices;
    uint32_t limit = 0;
    bool open = false;
    void reserve(size_t blocks, size_t voices_per) {
        istart.reserve(blocks + blocks / 4 + 16);
        ivoff.reserve(blocks + blocks / 4 + 17);
        voices.reserve((blocks + blocks / 4 + 16) * voices_per);
    }
    bool begin(size_t m) {
        if (m >= limit) { open = false; return false; }
        if (!istart.empty() && istart.back() == (uint32_t)m) {
            voices.resize(ivoff.back());
        } else {
            istart.push_back((uint32_t)m);
            ivoff.push_back((uint32_t)voices.size());
        }
        open = true;
        return true;
    }
    void push(float a, float b, float c, float d) { if (open) voices.push_back(make_float4(a, b, c, d)); }
    void finish() { ivoff.push_back((uint32_t)voices.size()); }
};

static inline float note_hz(float note) { return 440.0f * powf(2.0f, (note - 69.0f) / 12.0f); }   // extensions.rs:451,503

struct VTables {   // per-vertex compile result: offsets into the vertex' table buffer (TableCache)
    const uint8_t* dev = nullptr;   // device address the offsets refer to
    size_t hits_off = 0;
    uint32_t n_hits = 0;
    size_t istart_off = 0, ivoff_off = 0, voices_off = 0, tile_first_off = 0, tile_order_off = 0;
    uint32_t n_int = 0;
    uint64_t t0 = 0;
    size_t raw_istart_off = 0, raw_ivoff_off = 0, raw_voices_off = 0, raw_tile_first_off = 0;   // (TableCache: a probed affine Synth vertex)
    uint32_t raw_n_int = 0;
    size_t probe_v_off = 0;   // (a probed sine / Synth vertex) per sample of k_sine_probe: its frame's voice range in the raw table
    float hz_max = 0.0f;      // (Synth) the largest |hz| of any voice record of the tables (NaN: +inf): SynthDesc::small_args
};

static PanGain make_pg(float gain, float angle) {
    PanGain pg{1.0f, 1.0f, 1.0f, 0u};
    if (!(fabsf(angle) < 0.001f)) {   // sample.rs:98
        const float angle_rad = angle * 0.5f * 0.01745329f;
        const float k = 0.707106781186547524400844362104849039f;   // FRAC_1_SQRT_2
        pg.l_amp = k * (cosf(angle_rad) + sinf(angle_rad));
        pg.r_amp = k * (cosf(angle_rad) - sinf(angle_rad));
        pg.flags |= 1u;
    }
    if (!(fabsf(gain - 1.0f) < 0.001f)) {   // sample.rs:109
        pg.gain = gain;
        pg.flags |= 2u;
    }
    return pg;
}

static const std::vector<td_event>& floww_of(const td_flowwbank* fb, size_t idx) {
    static const std::vector<td_event> empty;
    return idx < fb->flowws.size() ? fb->flowws[idx] : empty;
}
static size_t start_of(const BlockCursor& c, size_t idx) { return idx < c.n ? c.start[idx] : 0; }

// ---- SampleMulti (extensions.rs:344-381) ----
static void compile_multi(Vertex& v, size_t L, const td_flowwbank* fb, const std::vector<BlockCursor>& cur,
                          size_t bl, Staging& st, VTables& vt) {
    const size_t M = cur.size() * bl;
    std::vector<MultiHit> hits;
    for (auto& tv : v.ts) hits.push_back({-tv.first, tv.second, 0.f});
    const auto& ev = floww_of(fb, v.floww_index);
    for (size_t b = 0; b < cur.size(); ++b) {
        drum_block(ev, *fb, cur[b].frame, start_of(cur[b], v.floww_index), bl, [&](size_t i, float note, float vel) {
            const bool ok = v.has_note ? fabsf(note - (float)v.note) < 0.01f : true;
            if (ok) hits.push_back({(int64_t)(b * bl + i), vel, 0.f});
        });
    }
    v.ts.clear();
    for (auto& h : hits)
        if (h.origin + (int64_t)L > (int64_t)M) v.ts.push_back({(int64_t)M - h.origin, h.vel});
    vt.n_hits = (uint32_t)hits.size();
    vt.hits_off = st.put(hits);
    // per tile: first hit whose voice can still sound at the tile's first frame
    std::vector<uint32_t> tf((M + kTileFrames - 1) / kTileFrames + 1);
    uint32_t j = 0;
    for (size_t t = 0; t < tf.size(); ++t) {
        const int64_t lo_key = (int64_t)(t * kTileFrames) - (int64_t)L;
        while (j < hits.size() && hits[j].origin <= lo_key) ++j;
        tf[t] = j;
    }
    vt.tile_first_off = st.put(tf);
}

// ---- SampleLerp (extensions.rs:384-421) ----
static void compile_lerp(Vertex& v, const td_flowwbank* fb, const std::vector<BlockCursor>& cur, size_t bl,
                         Staging& st, VTables& vt) {
    const int64_t M = (int64_t)(cur.size() * bl);
    const int64_t never = std::numeric_limits<int64_t>::min();
    std::vector<LerpHit> hits;
    hits.push_back({never, -v.g_off, never / 2, v.g_vel, 0.f});
    hits.push_back({never, -v.p_off, (int64_t)v.countdown - (int64_t)v.lerp_len, v.p_vel, 0.f});
    const auto& ev = floww_of(fb, v.floww_index);
    for (size_t b = 0; b < cur.size(); ++b) {
        drum_block(ev, *fb, cur[b].frame, start_of(cur[b], v.floww_index), bl, [&](size_t i, float note, float vel) {
            const bool ok = v.has_note ? fabsf(note - (float)v.note) < 0.01f : true;
            if (ok) {
                const int64_t m = (int64_t)(b * bl + i);
                hits.push_back({m, m, m, vel, 0.f});
            }
        });
    }
    const LerpHit& p = hits.back();
    const LerpHit& g = hits[hits.size() - 2];
    const int64_t since = M - p.fade;
    v.countdown = since < (int64_t)v.lerp_len ? (uint64_t)((int64_t)v.lerp_len - since) : 0;
    v.p_off = M - p.origin;
    v.p_vel = p.vel;
    v.g_off = M - g.origin;
    v.g_vel = g.vel;
    vt.n_hits = (uint32_t)hits.size();
    vt.hits_off = st.put(hits);
    // per tile: number of entries whose key is <= the tile's first frame
    std::vector<uint32_t> tf((size_t)(M + kTileFrames - 1) / kTileFrames + 1);
    uint32_t j = 0;
    for (size_t t = 0; t < tf.size(); ++t) {
        while (j < hits.size() && hits[j].key <= (int64_t)(t * kTileFrames)) ++j;
        tf[t] = j;
    }
    vt.tile_first_off = st.put(tf);
}

struct IntervalView {   // what put_intervals reads of an interval table
    std::vector<uint32_t>& istart;
    std::vector<uint32_t>& ivoff;
    std::vector<float4>& voices;
    uint32_t limit;
};
static void put_intervals(IntervalView ib, Staging& st, VTables& vt);
static void put_intervals(IntervalBuilder& b, Staging& st, VTables& vt) {
    b.finish();
    put_intervals(IntervalView{b.istart, b.ivoff, b.voices, b.limit}, st, vt);
}
static void put_intervals(IntervalView ib, Staging& st, VTables& vt) {
    vt.n_int = (uint32_t)ib.istart.size();
    // per 1024-frame tile: the interval that holds the tile's first frame (istart[0] == 0, ascending)
    std::vector<uint32_t> tile_first((ib.limit + kTileFrames - 1) / kTileFrames + 1);
    uint32_t it = 0;
    for (size_t t = 0; t < tile_first.size(); ++t) {
        const uint32_t m = (uint32_t)(t * kTileFrames);
        while (it + 1 < ib.istart.size() && ib.istart[it + 1] <= m) ++it;
        tile_first[t] = it;
    }
    vt.tile_first_off = st.put(tile_first);
    {   // IntervalTab::tile_order: the costliest tiles first -- a tile's cost is a pass of the voice loop per interval it holds
        // frames of (the tiles with an interval start strictly inside them: the waves that take the per-interval passes), each as
        // long as the interval has voices.  Longest first, the grid's last workgroups are the cheap ones -- the tiles between two
        // chords -- and the CUs run dry together (round 6: the tests' 60-voice project 72 -> 65 us, config 3's k_sources -1 us; before:
        // multi-interval tiles first, the rest in timeline order).
        const size_t nt = tile_first.size() - 1;
        std::vector<uint64_t> cost(nt);
        for (size_t t = 0; t < nt; ++t) {
            uint32_t last = tile_first[t + 1];
            if (last > tile_first[t] && ib.istart[last] == (t + 1) * kTileFrames) --last;   // (the next tile's first interval starts exactly there)
            uint64_t c = 0;
            for (uint32_t i = tile_first[t]; i <= last; ++i) c += (uint64_t)(ib.ivoff[i + 1] - ib.ivoff[i]) + 2u;
            cost[t] = c;
        }
        std::vector<uint32_t> order(nt);
        for (size_t t = 0; t < nt; ++t) order[t] = (uint32_t)t;
        std::stable_sort(order.begin(), order.end(), [&cost](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
        vt.tile_order_off = st.put(order);
    }
    vt.istart_off = st.put(ib.istart);
    vt.ivoff_off = st.put(ib.ivoff);
    vt.voices_off = st.put(ib.voices);
}

// k_sine_probe's sampling of a chunk of M frames (kernels.h ProbeDesc): one frame in every `1 << lg` -- every 256th of a long
// chunk, every 16th of a single block
static uint32_t probe_stride_log2(size_t M) {
    uint32_t lg = 4;
    while (lg < 8 && ((size_t)128 << lg) <= M) ++lg;
    return lg;
}
// ... and, per sample, the voice records of its frame's interval {first, one past the last} -- what the kernel would find by
// walking the interval table (two or three dependent loads in a launch that is nothing but latency)
static size_t put_probe_ranges(IntervalView ib, Staging& st) {
    const uint32_t lg = probe_stride_log2(ib.limit);
    const uint32_t n = (uint32_t)(((size_t)ib.limit + ((size_t)1 << lg) - 1) >> lg);
    std::vector<uint32_t> r((size_t)n * 2 + 2, 0u);
    size_t it = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t m = probe_frame(i, lg);   // (ascending in i)
        if (m >= ib.limit || ib.istart.empty()) continue;
        while (it + 1 < ib.istart.size() && ib.istart[it + 1] <= m) ++it;
        r[2 * (size_t)i] = ib.ivoff[it];
        r[2 * (size_t)i + 1] = ib.ivoff[it + 1];
    }
    return st.put(r);
}
// ---- DebugSine (extensions.rs:423-457) ----
static void compile_sine(Vertex& v, const td_flowwbank* fb, const std::vector<BlockCursor>& cur, size_t bl,
                         Staging& st, VTables& vt) {
    IntervalBuilder ib;
    ib.limit = (uint32_t)(cur.size() * bl);
    auto emit = [&](size_t m) {
        if (!ib.begin(m)) return;
        for (auto& n : v.sine_notes) ib.push(note_hz(n.note), n.vel, 0.f, 0.f);
    };
    const auto& ev = floww_of(fb, v.floww_index);
    for (size_t b = 0; b < cur.size(); ++b) {
        emit(b * bl);
        simple_block(ev, *fb, cur[b].frame, start_of(cur[b], v.floww_index), bl,
            [&](size_t, bool on, float note, float vel) {
                if (on) {
                    bool has = false;
                    for (auto& n : v.sine_notes)
                        if (fabsf(n.note - note) < 0.001f) { n.vel = vel; has = true; break; }
                    if (!has) v.sine_notes.push_back({note, vel});
                } else {
                    v.sine_notes.erase(std::remove_if(v.sine_notes.begin(), v.sine_notes.end(),
                                                      [&](const SineNote& x) { return !(fabsf(x.note - note) > 0.001f); }),
                                       v.sine_notes.end());
                }
            },
            [&](size_t i) { emit(b * bl + i); });
    }
    put_intervals(ib, st, vt);
    if (v.probe) vt.probe_v_off = put_probe_ranges(IntervalView{ib.istart, ib.ivoff, ib.voices, ib.limit}, st);
}

// ---- Synth (extensions.rs:460-529) ----
static float synth_release_sec(const Vertex& v) {   // extensions.rs:469-478
    float release_sec = 0.0f;
    if (v.square.volume > 0.0f) release_sec = v.square.adsr.release_sec;
    if (v.topflat.volume > 0.0f) release_sec = fmaxf(release_sec, v.topflat.adsr.release_sec);
    if (v.triangle.volume > 0.0f) release_sec = fmaxf(release_sec, v.triangle.adsr.release_sec);
    return release_sec;
}
// ---- k_synth's affine envelope form -------------------------------------------------------------------------------------
// Every envelope of synth_gen (extensions.rs:498-524; adsr.rs:46-92) is piecewise linear in the voice's envelope time: the
// host cuts a Synth vertex' intervals at every frame where a voice changes piece (attack / decay / sustain ramp / hold; for a
// released voice: release ramp / its clamp), so that inside an interval each oscillator's  envelope x velocity x volume x
// amplitude multiplier x shape scale  is ONE affine function  A + B ((t - s1) - s2)  of t = env_t + off -- (A, B, s1, s2) ride
// in the voice record, the subtraction order is the reference's (`t - attack_sec - decay_sec`), the kernel does two
// subtractions and one FMA per oscillator and frame pair and no piece selection at all.  Voice record: four float4 --
// (hz, env_t, 0, 0), then (s1, s2, A, B) for square, top-flat, triangle (A = B = 0: oscillator off).
// Only for confs whose pieces cannot reach the `res <= -1.0` escape of adsr.rs:62-69 and whose times are finite, the attack
// longer than zero (quirk Q6's NaN frame) -- anything else keeps the generic per-frame evaluation.
static bool synth_affine_ok(const Vertex& v) {
    if (v.exact_sin) return false;   // (sine_mode 1: the generic per-frame form carries sin_glibc)
    float amp = 0.0f;
    for (const tdk::OscConfD* o : {&v.square, &v.topflat, &v.triangle}) {
        if (!(o->volume > 0.0f)) {
            if (o->volume != o->volume) return false;
            continue;
        }
        const AdsrConfD& c = o->adsr;
        for (float x : {c.std_vel, c.attack_vel, c.decay_vel, c.sustain_vel, c.release_vel, o->volume, o->param})
            if (!std::isfinite(x)) return false;
        if (!(fminf(fminf(c.std_vel, c.attack_vel), fminf(c.decay_vel, c.sustain_vel)) > -0.999f)) return false;
        // A zero-length decay / sustain / release piece is never the selected one (`t <= a + d + 0` is piece 1's own test;
        // `min(t / 0, 1)` is 1 for every t >= 0, NaN included: f32::min); a zero-length attack IS selected at t == 0 and
        // yields 0 / 0 (quirk Q6): generic form.
        if (!(c.attack_sec > 0.0f)) return false;
        for (float x : {c.attack_sec, c.decay_sec, c.sustain_sec, c.release_sec})
            if (!(x >= 0.0f) || !std::isfinite(x)) return false;
        amp += o->volume * adsr_max_vel(c);
    }
    if (v.square.volume > 0.0f && !(v.square.param > 0.0f)) return false;
    if (v.topflat.volume > 0.0f && !(1.0f + v.topflat.param > 0.0f)) return false;
    return amp > 0.0f && std::isfinite(1.0f / amp);
}
// piece of conf c at in-block frame i: 0 attack, 1 decay, 2 sustain ramp, 3 hold; released voices: 4 release ramp, 5 clamped
static inline int synth_piece(const AdsrConfD& c, float env_t, float rel_t, size_t i, float srf) {
    const float t = env_t + (float)i / srf;
    if (rel_t != 0.0f) return (t / c.release_sec < 1.0f) ? 4 : 5;   // fminf(t / release_sec, 1.0), adsr.rs:72
    return t <= c.attack_sec ? 0 : t <= c.attack_sec + c.decay_sec ? 1 : t <= c.attack_sec + c.decay_sec + c.sustain_sec ? 2 : 3;
}
static inline float4 synth_osc_piece(const AdsrConfD& c, int piece, float rel_t, double K) {   // (s1, s2, A, B)
    double v0 = 0.0, dv = 0.0, len = 1.0;
    float s1 = 0.0f, s2 = 0.0f;
    switch (piece) {
        case 0: v0 = c.std_vel; dv = (double)(c.attack_vel - c.std_vel); len = c.attack_sec; break;
        case 1: v0 = c.attack_vel; dv = (double)(c.decay_vel - c.attack_vel); len = c.decay_sec; s1 = c.attack_sec; break;
        case 2: v0 = c.decay_vel; dv = (double)(c.sustain_vel - c.decay_vel); len = c.sustain_sec; s1 = c.attack_sec; s2 = c.decay_sec; break;
        case 3: v0 = c.sustain_vel; break;
        case 4: {
            const float held = apply_ads(c, rel_t);   // adsr.rs:89-92
            v0 = held;
            dv = (double)(c.release_vel - held);
            len = c.release_sec;
        } break;
        default: v0 = c.release_vel; break;
    }
    return make_float4(s1, s2, (float)(v0 * K), (float)(dv / len * K));
}
// raw intervals (block starts + event frames, one float4 (hz, vel, env_t, rel_t) per voice) -> refined ones + affine records
static void synth_refine_affine(const Vertex& v, IntervalBuilder& raw, size_t bl, size_t sr, std::vector<uint32_t>& istart,
                                std::vector<uint32_t>& ivoff, std::vector<float4>& rec) {
    const float srf = (float)sr;
    const tdk::OscConfD* osc[3] = {&v.square, &v.topflat, &v.triangle};
    const double amp = 1.0 / (double)(v.square.volume * adsr_max_vel(v.square.adsr) + v.topflat.volume * adsr_max_vel(v.topflat.adsr) +
                                      v.triangle.volume * adsr_max_vel(v.triangle.adsr));
    // shape scales folded into the records: square clamp(sn, -z, z) * (1 / z); top-flat (min(sn, z) + (1 - z) / 2) * (2 / (1 + z))
    const double shape[3] = {1.0 / (double)v.square.param, 2.0 / (double)(1.0f + v.topflat.param), 1.0};
    // distinct enabled confs (cuts are needed once per distinct conf)
    int conf_of[3] = {-1, -1, -1}, n_conf = 0;
    const AdsrConfD* confs[3];
    for (int o = 0; o < 3; ++o) {
        if (!(osc[o]->volume > 0.0f)) continue;
        int k = -1;
        for (int q = 0; q < n_conf; ++q)
            if (memcmp(confs[q], &osc[o]->adsr, sizeof(AdsrConfD)) == 0) k = q;
        if (k < 0) { confs[n_conf] = &osc[o]->adsr; k = n_conf++; }
        conf_of[o] = k;
    }
    const size_t n_raw = raw.istart.size();
    istart.clear(); ivoff.clear(); rec.clear();
    istart.reserve(n_raw + n_raw / 8 + 16);
    ivoff.reserve(n_raw + n_raw / 8 + 17);
    rec.reserve(raw.voices.size() * 4 + 64);
    std::vector<uint32_t> cuts;
    for (size_t r = 0; r < n_raw; ++r) {
        const uint32_t s = raw.istart[r], e = r + 1 < n_raw ? raw.istart[r + 1] : raw.limit;
        const size_t blk0 = (size_t)s / bl * bl;            // (an interval never crosses a block start)
        const size_t i_s = s - blk0, i_e = e - blk0;
        const float4* vo = raw.voices.data() + raw.ivoff[r];
        const size_t nv = raw.ivoff[r + 1] - raw.ivoff[r];
        cuts.clear();
        for (size_t q = 0; q < nv; ++q)
            for (int k = 0; k < n_conf; ++k) {
                const AdsrConfD& c = *confs[k];
                const float env_t = vo[q].z, rel_t = vo[q].w;
                size_t i = i_s;
                int p = synth_piece(c, env_t, rel_t, i, srf);
                while (i + 1 < i_e && synth_piece(c, env_t, rel_t, i_e - 1, srf) != p) {
                    size_t lo = i + 1, hi = i_e - 1;        // first frame in (i, i_e) whose piece differs from p (it exists: i_e - 1 differs)
                    while (lo < hi) {
                        const size_t mid = (lo + hi) / 2;
                        if (synth_piece(c, env_t, rel_t, mid, srf) != p) hi = mid; else lo = mid + 1;
                    }
                    cuts.push_back((uint32_t)(blk0 + lo));
                    i = lo;
                    p = synth_piece(c, env_t, rel_t, i, srf);
                }
            }
        std::sort(cuts.begin(), cuts.end());
        cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
        size_t ci = 0;
        uint32_t a = s;
        for (;;) {
            istart.push_back(a);
            ivoff.push_back((uint32_t)(rec.size() / 4));
            const size_t ia = a - blk0;
            for (size_t q = 0; q < nv; ++q) {
                const float hz = vo[q].x, vel = vo[q].y, env_t = vo[q].z, rel_t = vo[q].w;
                const size_t head = rec.size();
                rec.push_back(make_float4(hz, env_t, 0.0f, 0.0f));
                uint32_t live = 0;   // bit o: oscillator o's record is not A = B = 0 (a piece that is identically 0 adds nothing: the kernel skips it)
                for (int o = 0; o < 3; ++o) {
                    if (conf_of[o] < 0) { rec.push_back(make_float4(0.f, 0.f, 0.f, 0.f)); continue; }
                    const AdsrConfD& c = osc[o]->adsr;
                    const double K = (double)vel * (double)osc[o]->volume * amp * shape[o];
                    const float4 piece = synth_osc_piece(c, synth_piece(c, env_t, rel_t, ia, srf), rel_t, K);
                    rec.push_back(piece);
                    if (!(piece.z == 0.0f && piece.w == 0.0f)) live |= 1u << o;
                }
                memcpy(&rec[head].z, &live, 4);   // (a bit pattern in a float's place: read by the scalar unit, never by an FP instruction)
            }
            if (ci == cuts.size()) break;
            a = cuts[ci++];
        }
    }
    ivoff.push_back((uint32_t)(rec.size() / 4));
    for (int z = 0; z < 8; ++z) rec.push_back(make_float4(0.f, 0.f, 0.f, 0.f));   // (the voice loop reads one record ahead)
}

// Shared by Synth (extensions.rs:460-529) and SampSyn (extensions.rs:532-578): identical voice bookkeeping,
// only the retain threshold differs (max release over enabled oscillators vs the single ADSR's release).
static int compile_synth(Vertex& v, const td_flowwbank* fb, const std::vector<BlockCursor>& cur, size_t bl,
                         size_t sr, Staging& st, VTables& vt) {
    IntervalBuilder ib;
    ib.limit = (uint32_t)(cur.size() * bl);
    ib.reserve(cur.size(), std::max<size_t>(v.notes.size(), 8));
    const float release_sec = v.kind == K_SAMPSYN ? v.conf.release_sec : synth_release_sec(v);
    auto emit = [&](size_t m) {
        if (!ib.begin(m)) return;
        for (auto& n : v.notes) ib.push(n.hz, n.vel, n.env_t, n.rel_t);
    };
    const auto& ev = floww_of(fb, v.floww_index);
    bool impossible = false;
    for (size_t b = 0; b < cur.size(); ++b) {
        emit(b * bl);
        simple_block(ev, *fb, cur[b].frame, start_of(cur[b], v.floww_index), bl,
            [&](size_t i, bool on, float note, float vel) {
                const float off = (float)i / (float)sr;
                if (on) {
                    v.notes.push_back({note, vel, -off, 0.0f, note_hz(note)});
                } else {
                    v.notes.erase(std::remove_if(v.notes.begin(), v.notes.end(),
                                                 [&](const SynthNote& x) {
                                                     return !(fabsf(x.note - note) > 0.001f || x.rel_t == 0.0f);
                                                 }),
                                  v.notes.end());
                    for (auto& x : v.notes) {
                        if (fabsf(x.note - note) > 0.001f) continue;
                        if (x.rel_t == 0.0f) {
                            x.rel_t = x.env_t + off;
                            x.env_t = -off;
                        } else {
                            impossible = true;   // the reference panics here (extensions.rs:492)
                        }
                    }
                }
            },
            [&](size_t i) { emit(b * bl + i); });
        for (auto& x : v.notes) x.env_t += (float)bl / (float)sr;
        v.notes.erase(std::remove_if(v.notes.begin(), v.notes.end(),
                                     [&](const SynthNote& x) { return !(x.rel_t == 0.0f || x.env_t <= release_sec); }),
                      v.notes.end());
    }
    if (impossible) return fail("Synth: impossible release stage note");
    vt.hz_max = 0.0f;
    for (const float4& n : ib.voices) vt.hz_max = fabsf(n.x) <= vt.hz_max ? vt.hz_max : (n.x == n.x ? fabsf(n.x) : INFINITY);
    if (v.kind == K_SYNTH && synth_affine_ok(v)) {
        ib.finish();
        std::vector<uint32_t> istart, ivoff;
        std::vector<float4> rec;
        synth_refine_affine(v, ib, bl, sr, istart, ivoff, rec);
        put_intervals(IntervalView{istart, ivoff, rec, ib.limit}, st, vt);
        if (v.probe) {   // (sine_mode 2) the raw table too: k_sine_probe evaluates the reference's own per-frame form
            VTables raw;
            put_intervals(IntervalView{ib.istart, ib.ivoff, ib.voices, ib.limit}, st, raw);
            vt.raw_istart_off = raw.istart_off; vt.raw_ivoff_off = raw.ivoff_off; vt.raw_voices_off = raw.voices_off;
            vt.raw_tile_first_off = raw.tile_first_off; vt.raw_n_int = raw.n_int;
            vt.probe_v_off = put_probe_ranges(IntervalView{ib.istart, ib.ivoff, ib.voices, ib.limit}, st);
        }
        return 1;
    }
    put_intervals(ib, st, vt);
    vt.raw_istart_off = vt.istart_off; vt.raw_ivoff_off = vt.ivoff_off; vt.raw_voices_off = vt.voices_off;
    vt.raw_tile_first_off = vt.tile_first_off; vt.raw_n_int = vt.n_int;
    if (v.probe) vt.probe_v_off = put_probe_ranges(IntervalView{ib.istart, ib.ivoff, ib.voices, ib.limit}, st);
    return 1;
}

// ---- Adsr vertex (extensions.rs:593-651) ----
static void compile_adsr(Vertex& v, const td_flowwbank* fb, const std::vector<BlockCursor>& cur, size_t bl,
                         size_t sr, Staging& st, VTables& vt) {
    IntervalBuilder ib;
    ib.limit = (uint32_t)(cur.size() * bl);
    ib.reserve(cur.size(), 2);
    auto emit = [&](size_t m, float skip) {
        if (!ib.begin(m)) return;
        ib.push(v.ap.t, v.ap.vel, v.ap.rel, skip);
        ib.push(v.ag.t, v.ag.vel, v.ag.rel, 0.f);
    };
    const auto& ev = floww_of(fb, v.floww_index);
    for (size_t b = 0; b < cur.size(); ++b) {
        emit(b * bl, 0.f);
        if (v.use_off) {
            simple_block(ev, *fb, cur[b].frame, start_of(cur[b], v.floww_index), bl,
                [&](size_t i, bool on, float n, float vel) {
                    if (v.has_note && fabsf((float)v.note - n) > 0.01f) return;   // :606-608
                    const float off = (float)i / (float)sr;
                    if (on) {
                        v.ag = v.ap;
                        v.ap = {-off, vel, 0.0f};
                    } else if (v.ag.rel == 0.0f) {
                        v.ag.t = -off;
                        v.ag.rel = apply_ads(v.conf, v.ag.t + off) * v.ag.vel;
                    } else {
                        v.ap.t = -off;
                        v.ap.rel = apply_ads(v.conf, v.ap.t + off) * v.ap.vel;
                    }
                },
                [&](size_t i) { emit(b * bl + i, 0.f); });
        } else {
            drum_block(ev, *fb, cur[b].frame, start_of(cur[b], v.floww_index), bl, [&](size_t i, float n, float vel) {
                if (v.has_note && fabsf((float)v.note - n) > 0.01f) {
                    // :632-635 `continue`: this frame is left un-enveloped, state unchanged
                    emit(b * bl + i, 1.f);
                    if (i + 1 < bl) emit(b * bl + i + 1, 0.f);
                    return;
                }
                v.ag = v.ap;
                v.ap = {-((float)i / (float)sr), vel, 0.0f};
                emit(b * bl + i, 0.f);
            });
        }
        v.ap.t += (float)bl / (float)sr;
        v.ag.t += (float)bl / (float)sr;
    }
    put_intervals(ib, st, vt);
}

// ------------------------------------------------------------------------------------------------
// event-table cache (TableCache, engine.h)
// ------------------------------------------------------------------------------------------------
template <class T>
static void put_pod(std::string& s, const T& v) { s.append(reinterpret_cast<const char*>(&v), sizeof(T)); }
template <class T>
static void get_pod(const std::string& s, size_t& at, T& v) { memcpy(&v, s.data() + at, sizeof(T)); at += sizeof(T); }

// the carried host state of an event-driven vertex (what the reference keeps inside VertexExt), as bytes
void save_state(const Vertex& v, std::string& out) {
    switch (v.kind) {
        case K_SAMPLE_MULTI:
            put_pod(out, (uint64_t)v.ts.size());
            for (auto& e : v.ts) { put_pod(out, e.first); put_pod(out, e.second); }
            break;
        case K_SAMPLE_LERP:
            put_pod(out, v.p_off); put_pod(out, v.g_off); put_pod(out, v.p_vel); put_pod(out, v.g_vel); put_pod(out, v.countdown);
            break;
        case K_DEBUG_SINE:
            put_pod(out, (uint64_t)v.sine_notes.size());
            for (auto& n : v.sine_notes) put_pod(out, n);
            break;
        case K_SYNTH:
        case K_SAMPSYN:
            put_pod(out, (uint64_t)v.notes.size());
            for (auto& n : v.notes) put_pod(out, n);
            break;
        case K_ADSR: put_pod(out, v.ap); put_pod(out, v.ag); break;
        default: break;
    }
}
void load_state(Vertex& v, const std::string& in) {
    size_t at = 0;
    uint64_t n = 0;
    switch (v.kind) {
        case K_SAMPLE_MULTI:
            get_pod(in, at, n);
            v.ts.clear();
            for (uint64_t i = 0; i < n; ++i) { std::pair<int64_t, float> e; get_pod(in, at, e.first); get_pod(in, at, e.second); v.ts.push_back(e); }
            break;
        case K_SAMPLE_LERP:
            get_pod(in, at, v.p_off); get_pod(in, at, v.g_off); get_pod(in, at, v.p_vel); get_pod(in, at, v.g_vel); get_pod(in, at, v.countdown);
            break;
        case K_DEBUG_SINE:
            get_pod(in, at, n);
            v.sine_notes.resize((size_t)n);
            for (auto& x : v.sine_notes) get_pod(in, at, x);
            break;
        case K_SYNTH:
        case K_SAMPSYN:
            get_pod(in, at, n);
            v.notes.resize((size_t)n);
            for (auto& x : v.notes) get_pod(in, at, x);
            break;
        case K_ADSR: get_pod(in, at, v.ap); get_pod(in, at, v.ag); break;
        default: break;
    }
}
// Everything the compiled tables of vertex v for this chunk depend on.  (The per-block FlowwBank cursor follows from
// its first block: set_time_to_next_block is a pure function of the events and the previous cursor, floww.rs:70-91.)
static void table_key(const Vertex& v, const td_flowwbank* fb, const std::vector<BlockCursor>& cur, size_t bl, size_t sr,
                      size_t sample_len, std::string& key) {
    key.clear();
    put_pod(key, (uint32_t)v.kind);
    put_pod(key, (uint64_t)(uintptr_t)fb);
    put_pod(key, (uint64_t)v.floww_index);
    put_pod(key, v.floww_index < fb->versions.size() ? fb->versions[v.floww_index] : (uint64_t)0);
    put_pod(key, (uint64_t)cur.size()); put_pod(key, (uint64_t)bl); put_pod(key, (uint64_t)sr);
    put_pod(key, (uint64_t)cur[0].frame);
    put_pod(key, (uint64_t)(v.floww_index < cur[0].n ? cur[0].start[v.floww_index] : 0));
    put_pod(key, (uint8_t)v.has_note); put_pod(key, (uint64_t)v.note);
    switch (v.kind) {
        case K_SAMPLE_MULTI: put_pod(key, (uint64_t)sample_len); break;
        case K_SAMPLE_LERP: put_pod(key, (uint64_t)v.lerp_len); break;
        case K_DEBUG_SINE: put_pod(key, (uint8_t)v.probe); break;   // (the table's form: with or without k_sine_probe's voice ranges)
        case K_SYNTH: put_pod(key, v.square); put_pod(key, v.topflat); put_pod(key, v.triangle); put_pod(key, (uint8_t)v.exact_sin); put_pod(key, (uint8_t)v.probe); break;   // (retain rule: release times; the table's form)
        case K_SAMPSYN: put_pod(key, v.conf); break;
        case K_ADSR: put_pod(key, (uint8_t)v.use_off); put_pod(key, v.conf); break;
        default: break;
    }
    save_state(v, key);
}
// ------------------------------------------------------------------------------------------------
// plan: reachable set, topological levels (graph.rs:98-108 visits exactly the vertices that reach the
// output; others never run and never advance -- quirk Q12)
// ------------------------------------------------------------------------------------------------
void build_plan(td_graph* g) {
    const size_t n = g->vertices.size();
    g->level.assign(n, -1);
    g->order.clear();
    g->n_levels = 0;
    if (g->output_vertex >= 0) {
        // iterative post-order DFS over reverse edges
        std::vector<char> seen(n, 0);
        std::vector<std::pair<size_t, size_t>> stack;
        stack.push_back({(size_t)g->output_vertex, 0});
        seen[(size_t)g->output_vertex] = 1;
        while (!stack.empty()) {
            auto& top = stack.back();
            const size_t v = top.first;
            if (top.second < g->edges[v].size()) {
                const size_t u = g->edges[v][top.second++];
                if (!seen[u]) {
                    seen[u] = 1;
                    stack.push_back({u, 0});
                }
            } else {
                int lv = 0;
                for (size_t u : g->edges[v]) lv = std::max(lv, g->level[u] + 1);
                g->level[v] = lv;
                g->n_levels = std::max(g->n_levels, lv + 1);
                g->order.push_back(v);
                stack.pop_back();
            }
        }
    }
    g->plan_dirty = false;
}

// Band-pass execution plan (DESIGN.md "exact parallel band-pass"): segment length S, warm-up W.  The
// warm-up must outlast the contraction (1 - gamma)^W of the slower chain.  The choice affects speed only --
// k_band_fix verifies every segment bit for bit and repairs what failed.
constexpr size_t kBandBatchQuads = 12288;   // segments of one launch over all its projects before they are made longer (192 workgroups; measured: P = 2 .. 32 config-4 projects)
constexpr uint32_t kBandBatchMaxS = 4096;
constexpr uint32_t kBandGuessMin = 4096;    // the block-response guess is used where the short warm-up is at least this long (frames)
constexpr unsigned kBandScanDepth = 64;     // band_mode 1 / 2: look-back until (1 - gamma)^(tile K) <= e^-64
struct BandPlan {
    bool parallel = false;
    uint32_t S = 0, W = 0, Ws = 0, nseg = 0;
    float2* tmp = nullptr;    // materialised input sum
    float2* tmpq = nullptr;   // ... and its planar-in-4 copy (warm-up input)
    size_t blk_peaks_off = 0;
    // block-response guess (kernels.h BandRespParam): quick warm-up, Horner depth per smoother
    uint32_t Wq = 0, Wq2 = 0, Kl = 0, Kh = 0;
    double Al = 0.0, Ah = 0.0;
    size_t resp_off = 0, rp_off = 0;
};
static BandPlan plan_band(const td_graph* g, const Vertex& v, size_t M) {
    BandPlan p;
    if (g->band_serial) return p;   // (tests: the serial kernel)
    float gmin = 1.0f;
    if (v.lgamma != 0.0f) gmin = fminf(gmin, fabsf(v.lgamma));
    if (v.hgamma != 0.0f) gmin = fminf(gmin, fabsf(v.hgamma));
    // 150 / gamma: ~103 / gamma frames take a full-scale tail down to the denormal floor (so a warm-up that
    // starts in the sound before a silence reproduces the decay into it), the rest is coalescence margin
    const double w = (double)g->band_warmup / (double)gmin + 64.0;
    if (!(w <= 262144.0)) return p;   // cut-offs below ~5 Hz: the serial kernel is the better plan
    p.W = ((uint32_t)w + 31u) & ~31u;
    p.Ws = std::min(p.W, ((uint32_t)((double)g->band_short / (double)gmin + 64.0) + 31u) & ~31u);   // coalescence only
    p.Ws = (p.Ws + 255u) & ~255u;                                                   // whole 256-frame liveness blocks
    p.W = (std::max(p.W, p.Ws) + 255u) & ~255u;                                     // (every window starts on a block boundary)
    // (the guess pays where the short warm-up is long -- cut-offs below ~75 Hz; elsewhere the walk is a few
    // hundred steps anyway and the block responses would only cost their reduction in the input-sum kernel)
    if (g->band_quick && p.Ws >= kBandGuessMin) {   // (segments are whole 256-frame blocks: a window starts where a block does)
        // Horner depth: the chained block responses must carry the memory of everything that can still matter.  A deep
        // effect chain swings over tens of decades (84 envelope stages: 25), so "matter" is priced against the whole
        // f32 exponent range a past burst can tower over the present: (1 - gamma)^(256 K) <= e^-band_depth, 100 by
        // default.  A smoother slower than 256 blocks' worth keeps the plain warm-up.
        const double depth_nats = (double)g->band_depth;
        auto depth = [depth_nats](float gamma, double* A) -> uint32_t {
            *A = 0.0;
            if (gamma == 0.0f || gamma >= 0.05f) return 0u;      // constant chain / fast smoother: no responses, no guess (K = 0)
            const double q = 1.0 - (double)gamma;
            if (!(q > 0.0)) return 0u;
            *A = pow(q, 256.0);
            const double k = ceil(depth_nats / (-256.0 * log(q)));
            return k < 1.0 ? 1u : (k > 1e6 ? 1000000u : (uint32_t)k);
        };
        p.Kl = depth(v.lgamma, &p.Al);
        p.Kh = depth(v.hgamma, &p.Ah);
        const uint32_t wq = (((uint32_t)((double)g->band_quick / (double)gmin + 64.0) + 31u) & ~31u);
        const uint32_t wq2 = (((uint32_t)((double)g->band_medium / (double)gmin + 64.0) + 31u) & ~31u);
        if (p.Kl <= 200u && p.Kh <= 200u) {
            p.Wq = std::min(p.Ws, (wq + 255u) & ~255u);
            p.Wq2 = std::min(p.Ws, (std::max(wq, wq2) + 255u) & ~255u);
        }
    }
    p.S = 256;
    while ((M + p.S - 1) / p.S > kBandMaxSegs && p.S < kBandMaxS) p.S *= 2;
    // In a batch the segments grow with the number of projects rendering beside this one (round 6): every segment pays its
    // warm-up walk -- ~12 000 frames of input re-read for a 20 Hz smoother -- and a lone project needs all the 256-frame segments
    // it can get to fill the chip (11 252 quads for 60 s), where the neighbouring windows' re-reads hit L2.  P projects offer P
    // times the quads, their buffers no longer fit any cache, and the walks become HBM traffic: 11.3 ms per config-4 project
    // at P = 32, the same as alone.  Longer segments, fewer walks: the smallest power of two that still leaves ~1.5 workgroups
    // per CU (4.5 ms per project at P = 32 with 4 096 frames).  Speed only: k_band_fix checks every hand-over bit for bit.
    {
        const size_t P = std::max<size_t>(1, g->batch_projects);
        static const long seg_env = getenv("TD_BAND_SEG") ? atol(getenv("TD_BAND_SEG")) : 0;   // (experiments: the length by hand)
        // (... and never longer than the walk it saves: a 200 Hz smoother's 1 600 frames are not worth a 4 096-frame segment's
        // serial output phase -- BASELINE config 3 in a batch of 32: 0.35 ms per project with 4 096 frames, 0.20 with 1 024)
        const uint32_t walk = p.Wq2 ? p.Wq2 : p.Ws;
        if (seg_env >= 256 && seg_env % 256 == 0) p.S = (uint32_t)seg_env;
        else while (P * ((M + p.S - 1) / p.S) > kBandBatchQuads && p.S < kBandBatchMaxS && 2u * p.S <= walk) p.S *= 2;
    }
    p.nseg = (uint32_t)((M + p.S - 1) / p.S);
    p.parallel = p.nseg <= kBandMaxSegs && p.nseg >= 8;   // tiny chunks (block pulls) stay on the serial kernel
    if (!p.parallel) p.Wq = p.Wq2 = p.Kl = p.Kh = 0;     // (the serial kernel takes no guess)
    return p;
}

// Tolerance-class band-pass (engine option "band_mode" 1, kernels.h BandScanDesc): tile = nf * 256 frames, look-back depth
// K = tiles after which (1 - gamma)^(tile K) <= e^-band_scan_depth for the slower smoother (64 nats by default: what is cut
// off is below 2e-28 of the largest state the chunk has seen, so a decaying tail keeps its RELATIVE accuracy down to 1e-22 of
// the peak -- this mode answers to an RMS bound on the output, not to the
// exact kernels' bit-for-bit guess, which prices a past burst against the whole f32 exponent range: band_depth).  Not usable (-> the exact kernels)
// when that takes more than kScanMaxK tiles (cut-offs below ~1.5 Hz) or the chunk is too long for 32-bit tile frames.
struct ScanPlan {
    int nf = 16;
    uint32_t n_tiles = 0, K = 1;
    uint32_t Kw = 0;          // k_band_chain's look-back depth (= K); 0: not chainable
    size_t pw_off = 0, pk_off = 0;
};
static bool plan_band_scan(const td_graph* g, const Vertex& v, size_t M, ScanPlan* sp) {
    if (M >= 0x7FFF0000ull) return false;
    sp->nf = g->band_scan_nf == 8 ? 8 : 16;
    const double tile = (double)band_scan_tile_frames(sp->nf);
    sp->n_tiles = (uint32_t)((M + (size_t)tile - 1) / (size_t)tile);
    double kmax = 1.0;
    for (float gamma : {v.lgamma, v.hgamma}) {
        if (gamma == 0.0f) continue;            // constant chain: no look-back (kernels.hip k_band_scan)
        const double q = 1.0 - (double)gamma;
        if (!(q > 0.0)) continue;               // gamma = 1: the state is the last input frame
        const double per_tile = -tile * log(q); // nats of decay per tile
        kmax = std::max(kmax, ceil((double)kBandScanDepth / per_tile));
    }
    if (!(kmax <= (double)kScanMaxK)) return false;
    sp->K = (uint32_t)kmax;
    sp->Kw = sp->nf == 16 ? sp->K : 0u;   // (k_band_chain is built for 16 frames per lane)
    return true;
}

// ------------------------------------------------------------------------------------------------
// one chunk: compile tables + descriptors (compile_chunk), then upload and launch level by level (submit_chunk)
// ------------------------------------------------------------------------------------------------
double ms_between(std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::milli>(b - a).count();
}

// Steps 1 and 2 for ONE graph, appended to `cb` (which several graphs of a batch may share).
int compile_chunk(td_graph* g, const td_samplebank* sb, const td_flowwbank* fb,
                         const std::vector<BlockCursor>& cur, uint64_t t0, bool is_scan, void* pcm_dst, int qmode,
                         float amplitude, ChunkBuild& cb) {
    const size_t bl = g->bl, sr = g->sr;
    const size_t nb = cur.size();
    const size_t M = nb * bl;
    if (M == 0) return 1;
    if (M > 0xFFFFFFF0ull) return fail("termdaw_amd: chunk too long");
    const size_t nv = g->vertices.size();

    g->band_stats_off.clear();
    g->band_stats_base = nullptr;
    g->guard.chunk_audited = false;
    std::vector<std::vector<size_t>> cons(nv);
    for (size_t vi : g->order)
        for (size_t u : g->edges[vi]) cons[u].push_back(vi);
    const bool scan_on = g->band_mode >= 1;
    // ---- the guard (band_mode 2, engine.h tde::Guard): a band-pass vertex takes the scan only where the launch's own estimate
    // of its deviation can be carried to the graph's output -- `down[u]`: the static gain from vertex u's output to the
    // graph's (pan / gain of everything downstream, largest channel; several paths add up; an Adsr vertex on the way at the
    // largest gain its conf and its events' velocities allow) and the ONE Normalize vertex every path runs through, if any
    // (its 1 / max is read from its peak table by k_band_audit); anything else -- two Normalize vertices in a row, paths that
    // differ in it -- keeps the exact kernels.  So does a vertex downstream of a sample loop shorter than 2 048 frames:
    // a period shorter than the smoother's memory repeats its rounding pattern, the offsets add up coherently and no
    // level-based estimate bounds them (DESIGN.md 3e "The guard").
    const bool guard_on = g->band_mode == 2 && !g->guard.in_redo && g->band_chain;
    // ---- the sine kinds under the same guard (engine option "sine_mode" 2, kernels.h ProbeDesc): a debug_sine / synth vertex
    // takes its fast form only where what k_sine_probe measures at its output can be carried to the graph's output the same way;
    // anything else -- and the second render of a graph whose verdict was over the bound -- takes glibc's sinf (sine_mode 1's form)
    const bool sguard_on = g->sine_mode == 2 && !g->guard.in_redo;
    struct PathGain { double g; long norm; };   // norm: -1 none, >= 0 that Normalize vertex, -2 not analysable
    std::vector<PathGain> down;
    std::vector<char> short_up;
    auto own_gain = [](const Vertex& v) {
        const PanGain pg = make_pg(v.gain, v.angle);
        double a = 1.0;
        if (pg.flags & 1u) a *= std::max(fabs((double)pg.l_amp), fabs((double)pg.r_amp));
        if (pg.flags & 2u) a *= fabs((double)pg.gain);
        return a;
    };
    if (guard_on || sguard_on) {
        down.assign(nv, PathGain{0.0, -1});
        short_up.assign(nv, 0);
        for (size_t vi : g->order) {   // inputs first
            const Vertex& v = g->vertices[vi];
            char su = (v.kind == K_SAMPLE_LOOP && v.sample_index < sb->samples.size() && sb->samples[v.sample_index].len < 2048) ? 1 : 0;
            for (size_t u : g->edges[vi]) su = su || short_up[u];
            short_up[vi] = su;
        }
        for (size_t k = g->order.size(); k-- > 0;) {   // consumers first
            const size_t u = g->order[k];
            if ((long)u == g->output_vertex) { down[u] = PathGain{1.0, -1}; continue; }
            double sum = 0.0;
            long nz = -1;
            bool first = true;
            for (size_t w : cons[u]) {   // (a duplicate edge is listed twice: the term is summed twice)
                const Vertex& wv = g->vertices[w];
                long through = down[w].norm;
                if (through == -2) { nz = -2; break; }
                double L = own_gain(wv);
                if (wv.kind == K_ADSR && !(wv.wet < 0.0001f)) {   // |lerp(1, level x vel, wet)| <= max(1, |level| |vel|)
                    const AdsrConfD& c = wv.conf;
                    double lv = std::max(std::max(fabs((double)c.std_vel), fabs((double)c.attack_vel)),
                                         std::max(std::max(fabs((double)c.decay_vel), fabs((double)c.sustain_vel)), fabs((double)c.release_vel)));
                    double mv = 0.0;
                    for (const td_event& e : floww_of(fb, wv.floww_index)) mv = std::max(mv, fabs((double)e.vel));
                    L *= std::max(1.0, lv * mv);
                }
                if (wv.kind == K_NORMALIZE) {
                    if (through != -1) { nz = -2; break; }
                    through = (long)w;
                }
                if (first) { nz = through; first = false; }
                else if (nz != through) { nz = -2; break; }
                sum += L * down[w].g;
            }
            if (!(sum == sum) || std::isinf(sum)) nz = -2;
            down[u] = PathGain{sum, nz};
        }
    }
    auto guard_ok = [&](size_t vi, const ScanPlan& sp) {   // (`pass` vertices: the chain launch; `cut` vertices: k_band_scan, one vertex per launch)
        (void)sp;
        return down[vi].norm != -2 && !short_up[vi];
    };
    for (size_t vi : g->order) {
        Vertex& v = g->vertices[vi];
        if (v.kind != K_DEBUG_SINE && v.kind != K_SYNTH) continue;
        v.probe = sguard_on && down[vi].norm != -2;
        v.exact_sin = g->sine_mode == 1 || (g->sine_mode == 2 && !v.probe);
    }
    // ---- 1. host compile: sequential bookkeeping -> tables
    const auto tp0 = std::chrono::steady_clock::now();
    Staging& st = *cb.st;   // capacity kept from render to render
    cb.n_graphs += 1;
    cb.one_grid_sources = cb.one_grid_sources && g->one_grid_sources;
    std::vector<VTables> vt(nv);
    std::map<std::string, size_t> chunk_keys;   // table key -> first vertex of this chunk compiled from it
    std::string key;
    Staging tmp;
    for (size_t vi : g->order) {
        Vertex& v = g->vertices[vi];
        vt[vi].t0 = t0;
        size_t sample_len = 0;
        switch (v.kind) {
            case K_SAMPLE_LOOP:
                if (v.sample_index >= sb->samples.size()) return fail("sampleloop: sample index out of range");
                vt[vi].t0 = v.loop_t;
                v.loop_t += M;   // *t += len per block (extensions.rs:340)
                continue;
            case K_SAMPLE_MULTI:
                if (v.sample_index >= sb->samples.size()) return fail("sample_multi: sample index out of range");
                sample_len = sb->samples[v.sample_index].len;
                break;
            case K_SAMPLE_LERP:
                if (v.sample_index >= sb->samples.size()) return fail("sample_lerp: sample index out of range");
                break;
            case K_DEBUG_SINE:
            case K_SYNTH:
            case K_SAMPSYN: break;
            case K_ADSR:
                if (v.wet < 0.0001f) continue;   // :598 early return keeps clocks
                break;
            default: continue;
        }
        // an event-driven vertex: tables from (1) a vertex of this chunk with the same key, (2) this vertex' own cache
        // when the key has not changed since it was filled, (3) a replay of the events
        table_key(v, fb, cur, bl, sr, sample_len, key);
        TableCache* tc = nullptr;
        auto shared = g->table_cache ? chunk_keys.find(key) : chunk_keys.end();
        if (shared != chunk_keys.end()) {
            tc = g->vertices[shared->second].tables.get();
            load_state(v, tc->end_state);
        } else {
            if (!v.tables) v.tables = std::make_shared<TableCache>();
            tc = v.tables.get();
            if (g->table_cache && !tc->key.empty() && tc->key == key) {
                load_state(v, tc->end_state);
            } else {
                tmp.b.clear();
                VTables t;
                switch (v.kind) {
                    case K_SAMPLE_MULTI: compile_multi(v, sample_len, fb, cur, bl, tmp, t); break;
                    case K_SAMPLE_LERP: compile_lerp(v, fb, cur, bl, tmp, t); break;
                    case K_DEBUG_SINE: compile_sine(v, fb, cur, bl, tmp, t); break;
                    case K_SYNTH:
                    case K_SAMPSYN:
                        if (!compile_synth(v, fb, cur, bl, sr, tmp, t)) return 0;
                        break;
                    default: compile_adsr(v, fb, cur, bl, sr, tmp, t); break;
                }
                tc->key.clear();   // (not valid until the upload below has been queued)
                if (!upload_tables(g, *tc, tmp)) return 0;
                tc->hits_off = t.hits_off; tc->n_hits = t.n_hits;
                tc->istart_off = t.istart_off; tc->ivoff_off = t.ivoff_off; tc->voices_off = t.voices_off;
                tc->tile_first_off = t.tile_first_off; tc->n_int = t.n_int;
                tc->tile_order_off = t.tile_order_off;
                tc->raw_istart_off = t.raw_istart_off; tc->raw_ivoff_off = t.raw_ivoff_off; tc->raw_voices_off = t.raw_voices_off;
                tc->raw_tile_first_off = t.raw_tile_first_off; tc->raw_n_int = t.raw_n_int; tc->probe_v_off = t.probe_v_off;
                tc->hz_max = t.hz_max;
                tc->end_state.clear();
                save_state(v, tc->end_state);
                tc->key = key;
            }
            chunk_keys[key] = vi;
        }
        VTables& o = vt[vi];
        o.dev = tc->d;
        o.hits_off = tc->hits_off; o.n_hits = tc->n_hits;
        o.istart_off = tc->istart_off; o.ivoff_off = tc->ivoff_off; o.voices_off = tc->voices_off;
        o.tile_first_off = tc->tile_first_off; o.n_int = tc->n_int;
        o.tile_order_off = tc->tile_order_off;
        o.raw_istart_off = tc->raw_istart_off; o.raw_ivoff_off = tc->raw_ivoff_off; o.raw_voices_off = tc->raw_voices_off;
        o.raw_tile_first_off = tc->raw_tile_first_off; o.raw_n_int = tc->raw_n_int; o.probe_v_off = tc->probe_v_off;
        o.hz_max = tc->hz_max;
    }

    // ---- 2. descriptors: walk levels, assign edge buffers
    const auto tp1 = std::chrono::steady_clock::now();
    if (!ensure_buffers(g, M)) return 0;
    g->vbuf.assign(nv, nullptr);
    std::vector<int> last_use(nv, -1);
    for (size_t vi : g->order)
        for (size_t u : g->edges[vi]) last_use[u] = std::max(last_use[u], g->level[vi]);

    // source inlining: a sample_loop vertex that is not the output is gathered directly by its consumers
    // (all of them sum their inputs through the same term loop), so its edge buffer is never materialised
    std::vector<char> inlined(nv, 0);
    if (g->fuse_sources)
        for (size_t vi : g->order)
            inlined[vi] = g->vertices[vi].kind == K_SAMPLE_LOOP && (long)vi != g->output_vertex;
    // ... and so is a Sum vertex with exactly one (materialised) input -- a gain / pan stage: its consumers read
    // the input's buffer and apply `0.0 + x`, pan, gain themselves (term kind 4); one launch and one buffer less
    // ... and an Adsr vertex with one materialised input and ONE consumer whose kernel is of the summing family (a Sum, a
    // Normalize, a band-pass -- directly or through one gain / pan stage): that consumer evaluates the envelope itself,
    // as its only term or among others (term kind 5); inlined 3 = such an Adsr vertex, 4 = the stage behind one
    auto is_stage = [&](size_t vi) {   // a single-input Sum that is not the output
        return g->vertices[vi].kind == K_SUM && (long)vi != g->output_vertex && g->edges[vi].size() == 1;
    };
    auto takes_adsr_terms = [&](size_t c) {   // consumer kernels that take a kind-5 term (the k_sum family, k_band_pass)
        const Vertex& w = g->vertices[c];
        if (w.kind == K_NORMALIZE || w.kind == K_BAND_PASS) return true;
        return w.kind == K_SUM && !is_stage(c);
    };
    // the buffer behind input u (read directly, or through a gain / pan stage) must live until level lv
    auto outlive = [&](size_t u, int lv) {
        if (inlined[u] == 2) u = g->edges[u][0];
        else if (inlined[u]) return;   // (a loop source has no buffer)
        last_use[u] = std::max(last_use[u], lv);
    };
    if (g->fuse_sources)
        for (size_t vi : g->order) {   // topological order: the input's own flag is final here
            const Vertex& v = g->vertices[vi];
            if (v.kind == K_ADSR && g->inline_adsr && !(v.wet < 0.0001f) && (long)vi != g->output_vertex &&
                g->edges[vi].size() == 1 && inlined[g->edges[vi][0]] < 3 && cons[vi].size() == 1) {
                // (its input: an edge buffer, an inlined loop source or a gain / pan stage -- anything but another envelope)
                const size_t c = cons[vi][0];
                const bool direct = takes_adsr_terms(c);
                const bool staged = is_stage(c) && cons[c].size() == 1 && takes_adsr_terms(cons[c][0]);
                if (direct || staged) {
                    inlined[vi] = 3;
                    outlive(g->edges[vi][0], last_use[vi]);
                }
                continue;
            }
            if (!is_stage(vi)) continue;
            const size_t u = g->edges[vi][0];
            if (inlined[u] == 3) {
                inlined[vi] = 4;
                outlive(g->edges[u][0], last_use[vi]);   // the envelope's input outlives the stage's consumer
                continue;
            }
            if (inlined[u]) continue;
            inlined[vi] = 2;
            last_use[u] = std::max(last_use[u], last_use[vi]);   // the input must outlive the stage's consumers
        }

    std::vector<std::vector<size_t>> by_level(g->n_levels);
    for (size_t vi : g->order) by_level[g->level[vi]].push_back(vi);
    std::map<std::string, size_t> scan_pw_off;   // k_band_scan power tables of this chunk, by (gammas, nf)
    // gain buffers of the Adsr vertices that are read through by their consumers (InTerm kind 5): one per distinct
    // (event tables, conf, wet, flags) of the chunk -- the 84 envelope stages of a deep chain share one -- filled by
    // k_adsr_env at the vertex' own level, held until the chunk has been compiled
    std::map<std::string, float*> env_of_key;
    std::map<const float*, size_t> env_tile_at;   // envelope buffer -> scratch offset of its mean squares per 512 frames (AdsrVDesc::env_tile)
    std::vector<float*> env_of(nv, nullptr);
    std::vector<float2*> env_bufs;
    auto add_launch = [&](int fam, size_t off, int n, uint32_t aux, int level) {
        cb.launches.push_back({fam, off, n, aux, level, (uint32_t)M, (uint32_t)bl, is_scan ? 1 : 0});
    };
    // the buffer vertex u's output is read from: behind a gain / pan stage, an inlined Adsr vertex or both; npos: a loop source
    auto buffer_behind = [&](size_t u) -> size_t {
        if (inlined[u] == 4) u = g->edges[u][0];
        if (inlined[u] == 3) u = g->edges[u][0];
        if (inlined[u] == 2) u = g->edges[u][0];
        return inlined[u] ? (size_t)-1 : u;
    };
    // ---- chains of band-pass vertices in scan mode (kernels.h BandScanDesc): vertex b follows vertex a when b's only input
    // is a, directly or through up to three single-input, single-consumer links (gain / pan stage, Adsr vertex read
    // through, stage) -- the shape of a chain of effect stages.  All vertices of a chain but the last are never
    // materialised (inlined 5); the launch sits at the last vertex' level and evaluates the FIRST vertex' input terms.
    struct ChainLink { size_t vertex; bool adsr; };
    std::map<size_t, ScanPlan> scan_plan;                        // band-pass vertices that take k_band_scan
    std::map<size_t, std::vector<size_t>> chain_of;              // last vertex of a launch -> its vertices, first to last
    std::map<size_t, std::vector<ChainLink>> links_before;       // band-pass vertex -> the links between its predecessor and it
    if (scan_on) {
        for (size_t vi : g->order) {
            const Vertex& v = g->vertices[vi];
            if (v.kind != K_BAND_PASS || v.wet < 0.0001f || (v.lgamma == 0.0f && v.hgamma == 0.0f)) continue;
            if (g->band_mode == 2 && !guard_on) continue;   // (the redo of a guarded render, or chains switched off: exact kernels)
            ScanPlan sp;
            if (plan_band_scan(g, v, M, &sp) && (!guard_on || guard_ok(vi, sp))) scan_plan[vi] = sp;
        }
        std::map<size_t, size_t> prev_of, next_of;
        if (g->fuse_sources && g->band_chain)
            for (auto& kv : scan_plan) {
                const size_t b = kv.first;
                if (g->edges[b].size() != 1) continue;
                std::vector<ChainLink> links;
                size_t u = g->edges[b][0];
                bool ok = true;
                while (ok && links.size() < 3 && (inlined[u] == 2 || inlined[u] == 3 || inlined[u] == 4)) {
                    ok = cons[u].size() == 1;
                    links.insert(links.begin(), ChainLink{u, inlined[u] == 3});
                    u = g->edges[u][0];
                }
                if (!ok || inlined[u] || !scan_plan.count(u) || cons[u].size() != 1 || (long)u == g->output_vertex) continue;
                // (k_band_chain runs `pass` vertices -- whose right-channel smoothers reach no output -- at 16 frames per lane)
                if (!g->vertices[b].pass || !g->vertices[u].pass || !kv.second.Kw || !scan_plan[u].Kw) continue;
                prev_of[b] = u;
                next_of[u] = b;
                links_before[b] = links;
            }
        for (auto& kv : scan_plan) {
            const size_t head = kv.first;
            if (prev_of.count(head)) continue;   // not the first vertex of its chain
            std::vector<size_t> piece{head};
            size_t b = head;
            for (;;) {
                auto nx = next_of.find(b);
                const bool more = nx != next_of.end();
                if (!more || piece.size() == kScanMaxStages) {   // (longer chains are cut: the cut vertex is materialised)
                    if (piece.size() > 1) {
                        const size_t last = piece.back();
                        for (size_t i = 0; i + 1 < piece.size(); ++i) inlined[piece[i]] = 5;
                        for (size_t u : g->edges[piece[0]]) {   // the first vertex' inputs are read at the LAST vertex' level
                            const size_t bu = buffer_behind(u);
                            if (bu != (size_t)-1) last_use[bu] = std::max(last_use[bu], g->level[last]);
                        }
                        chain_of[last] = piece;
                    }
                    if (!more) break;
                    piece.clear();
                }
                b = nx->second;
                piece.push_back(b);
            }
        }
    }
    // ---- a `pass` band-pass vertex that is in no chain takes the chain kernel too, as a chain of one: that kernel keeps a
    // non-finite state non-finite for the rest of the chunk (BandScanDesc::poison), and can take the vertices on either side in
    if (scan_on && g->band_chain)
        for (auto& kv : scan_plan)
            if (inlined[kv.first] != 5 && !chain_of.count(kv.first) && g->vertices[kv.first].pass && kv.second.Kw)
                chain_of[kv.first] = std::vector<size_t>{kv.first};
    // ---- a Normalize vertex whose one input is the last vertex of a scan launch (directly or through such links) is
    // evaluated by that launch's epilogue in its fresh-render form (k_band_chain, BandScanDesc::norm): the conditions of
    // SumDesc mode 5 (k_norm1), with the wave-tile as the reference block.  A single band-pass vertex takes the chain kernel
    // as a chain of one for this.
    std::map<size_t, size_t> norm_of;                           // scan launch vertex -> the Normalize vertex it evaluates
    std::map<size_t, size_t> fused_norm;                        // ... and back
    std::map<size_t, std::vector<ChainLink>> links_after;       // scan launch vertex -> the links between it and that Normalize vertex
    if (scan_on && g->band_chain && g->fuse_sources && g->fuse_normalize && g->spec_normalize && g->single_pass_normalize && !is_scan &&
        bl == (size_t)kTileFrames && M < ((size_t)1 << 31)) {
        for (auto& kv : scan_plan) {
            const size_t L = kv.first;
            if (inlined[L] == 5 || (long)L == g->output_vertex || cons[L].size() != 1) continue;   // (only a launch's last vertex)
            const std::vector<size_t> piece = chain_of.count(L) ? chain_of[L] : std::vector<size_t>{L};
            bool ok = true;
            for (size_t b : piece) ok = ok && g->vertices[b].pass && scan_plan[b].Kw != 0u;
            if (!ok) continue;
            std::vector<ChainLink> links;
            size_t u = cons[L][0];
            while (links.size() < 3 && (inlined[u] == 2 || inlined[u] == 3 || inlined[u] == 4) && cons[u].size() == 1 &&
                   (long)u != g->output_vertex) {
                links.push_back(ChainLink{u, inlined[u] == 3});
                u = cons[u][0];
            }
            const Vertex& nv2 = g->vertices[u];
            if (nv2.kind != K_NORMALIZE || inlined[u] || g->edges[u].size() != 1) continue;
            if (nv2.peak_known && !nv2.has_init_override) continue;   // (after a scan: the speculative single pass of its own)
            // the walk must have come up the Normalize vertex' own input chain
            if (buffer_behind(g->edges[u][0]) != L) continue;
            norm_of[L] = u;
            fused_norm[u] = L;
            links_after[L] = links;
            if (!chain_of.count(L)) chain_of[L] = piece;   // a chain of one
        }
    }
    // ---- ... and a Sum vertex with several inputs whose one consumer is the first vertex of a chain launch is evaluated by
    // that launch's input phase (its terms summed, its pan / gain applied: BandScanDesc::pre) instead of a launch of its own
    std::map<size_t, size_t> presum_of;   // first vertex of a chain launch -> that Sum vertex
    std::map<size_t, size_t> presum_stage;   // ... -> the gain / pan stage between the two, if any (only meaningful with presum_of)
    if (scan_on && g->fuse_sources && g->band_chain && g->fuse_normalize)
        for (auto& kv : chain_of) {
            const size_t first = kv.second[0], last = kv.first;
            if (g->edges[first].size() != 1) continue;
            size_t u = g->edges[first][0];
            if (inlined[u] == 2 && cons[u].size() == 1) {   // a gain / pan stage in between
                presum_stage[first] = u;
                u = g->edges[u][0];
            }
            const Vertex& uv = g->vertices[u];
            if (inlined[u] || uv.kind != K_SUM || g->edges[u].size() < 2 || cons[u].size() != 1 || (long)u == g->output_vertex) continue;
            bool ok = true;
            for (size_t w : g->edges[u]) ok = ok && inlined[w] < 3;   // (edge buffers, loop sources, gain / pan stages: kinds 0 .. 4)
            if (!ok) continue;
            presum_of[first] = u;
            inlined[u] = 7;
            for (size_t w : g->edges[u]) {   // its inputs are read at the LAST vertex' level
                size_t bw = w;
                if (inlined[bw] == 2) bw = g->edges[bw][0];
                if (!inlined[bw]) last_use[bw] = std::max(last_use[bw], g->level[last]);
            }
        }
    // ---- gain buffers of the Adsr vertices that are read through (k_adsr_env), before everything else: they depend on
    // the event tables only, and a chain launch needs those of its links however deep they sit in the graph
    {
        std::vector<size_t> envs;
        for (size_t vi : g->order) {
            if (inlined[vi] != 3) continue;
            const Vertex& v = g->vertices[vi];
            std::string key;
            put_pod(key, (uint64_t)(uintptr_t)vt[vi].dev);
            put_pod(key, (uint64_t)vt[vi].istart_off); put_pod(key, (uint64_t)vt[vi].ivoff_off);
            put_pod(key, (uint64_t)vt[vi].voices_off); put_pod(key, (uint64_t)vt[vi].tile_first_off);
            put_pod(key, vt[vi].n_int);
            put_pod(key, v.conf); put_pod(key, v.wet);
            put_pod(key, (uint8_t)v.use_off); put_pod(key, (uint8_t)v.use_max);
            auto it = env_of_key.find(key);
            if (it == env_of_key.end()) {
                float2* b = take_buffer(g);
                if (!b) return fail("termdaw_amd: out of device memory for edge buffers");
                env_bufs.push_back(b);
                it = env_of_key.emplace(key, reinterpret_cast<float*>(b)).first;
                envs.push_back(vi);
            }
            env_of[vi] = it->second;
        }
        if (!envs.empty()) {   // the vertex' tables and envelope, as k_adsr would get them; output: its gain buffer
            std::vector<AdsrVDesc> d;
            for (size_t vi : envs) {
                const Vertex& v = g->vertices[vi];
                AdsrVDesc x{};
                x.tab.n_int = vt[vi].n_int;
                x.sr = (uint32_t)sr;
                x.bl = (uint32_t)bl;
                x.use_off = v.use_off;
                x.use_max = v.use_max;
                x.wet = v.wet;
                x.conf = v.conf;
                adsr_fill_run_consts(&x);
                x.env = env_of[vi];
                d.push_back(x);
            }
            const size_t off = st.put(d);
            for (size_t i = 0; i < envs.size(); ++i) {
                if (guard_on) {   // (the guarded chain launches multiply their estimate by the link's gain: kernels.h AdsrVDesc::env_tile)
                    const size_t so = cb.scratch_bytes;
                    cb.scratch_bytes += (((M + 511) / 512 + 1) * sizeof(float) + 255) & ~(size_t)255;
                    env_tile_at[env_of[envs[i]]] = so;
                    cb.scratch_fix.push_back({off + i * sizeof(AdsrVDesc) + offsetof(AdsrVDesc, env_tile), so});
                }
                const size_t t = off + i * sizeof(AdsrVDesc) + offsetof(AdsrVDesc, tab);
                const auto tf = [&](size_t field_off, size_t o2) {
                    const uint64_t p = (uint64_t)(uintptr_t)(vt[envs[i]].dev + o2);
                    memcpy(&st.b[t + field_off], &p, 8);
                };
                tf(offsetof(IntervalTab, istart), vt[envs[i]].istart_off);
                tf(offsetof(IntervalTab, tile_first), vt[envs[i]].tile_first_off);
                tf(offsetof(IntervalTab, ivoff), vt[envs[i]].ivoff_off);
                tf(offsetof(IntervalTab, voices), vt[envs[i]].voices_off);
            }
            add_launch(F_ENV, off, (int)envs.size(), 0u, 0);   // (level 0: beside the source vertices, which it may share a grid with -- k_sources)
        }
    }

    // scratch (device-only) region is laid out after the uploaded region
    auto scratch = [&](size_t n) { size_t o = cb.scratch_bytes; cb.scratch_bytes += (n + 255) & ~(size_t)255; return o; };
    auto ptr_field = [&](size_t desc_off, size_t field_off, size_t staging_off) {
        cb.table_fix.push_back({desc_off + field_off, staging_off});
    };
    // pointer to a table of an event-driven vertex: those live in the vertex' own device buffer (TableCache)
    auto tab_field = [&](size_t desc_off, size_t field_off, const VTables& t, size_t off) {
        const uint64_t p = (uint64_t)(uintptr_t)(t.dev + off);
        memcpy(&st.b[desc_off + field_off], &p, 8);
    };
    auto scratch_field = [&](size_t desc_off, size_t field_off, size_t s_off) { cb.scratch_fix.push_back({desc_off + field_off, s_off}); };

    const bool peaks_need_zero = !(bl == (size_t)kTileFrames);
    // the guard's bookkeeping (band_mode 2): where every Normalize vertex of the chunk keeps its peak table and carried max,
    // and what every guarded scan launch leaves for k_band_audit
    std::map<size_t, std::pair<size_t, size_t>> audit_norm;   // Normalize vertex -> scratch offsets (peaks, init snapshot)
    struct AuditSrc { size_t noise_off; uint32_t n_wt; size_t from; bool fused; size_t desc_off; uint32_t tile_frames; bool sampled = false; };   // from: the vertex whose output the estimate stands at
    std::vector<AuditSrc> audit_src;

    auto synth_desc_of = [&](size_t vi) {   // a Synth vertex' descriptor, tables aside
        const Vertex& v = g->vertices[vi];
        SynthDesc x{};
        x.tab.n_int = vt[vi].n_int;
        x.out = g->vbuf[vi];
        x.t0 = t0;
        x.sr = (uint32_t)sr;
        x.bl = (uint32_t)bl;
        x.square = v.square;
        x.topflat = v.topflat;
        x.triangle = v.triangle;
        x.osc_amp_multiplier =   // extensions.rs:465-468
            1.0f / (v.square.volume * adsr_max_vel(v.square.adsr) +
                    v.topflat.volume * adsr_max_vel(v.topflat.adsr) +
                    v.triangle.volume * adsr_max_vel(v.triangle.adsr));
        x.pg = make_pg(v.gain, v.angle);
        x.affine = synth_affine_ok(v) ? 1u : 0u;   // (the tables then hold affine records: compile_synth)
        x.exact_sin = v.exact_sin ? 1u : 0u;
        x.small_args = ((double)(t0 + M + 2) / (double)sr) * (double)vt[vi].hz_max * 6.2831854 < 2.0e6 ? 1u : 0u;
        {
            auto same = [](const AdsrConfD& a, const AdsrConfD& b) { return memcmp(&a, &b, sizeof(AdsrConfD)) == 0; };
            const bool sq = v.square.volume > 0.0f, tf = v.topflat.volume > 0.0f;
            x.tf_env_src = (sq && same(v.topflat.adsr, v.square.adsr)) ? 1u : 0u;
            x.tr_env_src = (sq && same(v.triangle.adsr, v.square.adsr)) ? 1u
                         : (tf && same(v.triangle.adsr, v.topflat.adsr)) ? 2u : 0u;
        }
        return x;
    };
    // k_sine_probe's sampling of this chunk (kernels.h ProbeDesc): one frame in every `1 << probe_lg` -- every 256th of a long
    // chunk, every 16th of a single block -- 64 samples per workgroup
    const uint32_t probe_lg = probe_stride_log2(M);
    const uint32_t probe_stride = 1u << probe_lg, probe_n = (uint32_t)((M + probe_stride - 1) / probe_stride), probe_groups = (probe_n + 15u) / 16u;
    std::vector<std::pair<size_t, size_t>> probe_src;   // (vertex, scratch offset of its ProbeDesc::noise)
    std::vector<size_t> probe_desc_off;                 // ... and where its ProbeDesc stands in the arena
    for (int lv = 0; lv < g->n_levels; ++lv) {
        std::vector<size_t> fam_v[F_COUNT];
        std::vector<float2*> level_tmp;            // scratch edge buffers that live for this level only
        std::map<size_t, BandPlan> band_plan;
        std::vector<size_t> norm_pending;
        std::map<size_t, uint32_t> norm_mode;   // Normalize vertices: SumDesc::mode (1 two passes, 3 / 4 one pass + k_norm_fix, 5 one pass)
        std::map<size_t, int> norm_tpw;         // ... of those, the ones that take k_norm1: tiles per workgroup
        for (size_t vi : by_level[lv]) {
            Vertex& v = g->vertices[vi];
            if (inlined[vi]) continue;
            if (fused_norm.count(vi)) continue;   // (evaluated, and its buffer taken, at its scan launch's level)
            g->vbuf[vi] = take_buffer(g);
            if (!g->vbuf[vi]) return fail("termdaw_amd: out of device memory for edge buffers");
            if (norm_of.count(vi)) {
                g->vbuf[norm_of[vi]] = take_buffer(g);
                if (!g->vbuf[norm_of[vi]]) return fail("termdaw_amd: out of device memory for edge buffers");
            }
            switch (v.kind) {
                case K_SAMPLE_LOOP: fam_v[F_LOOP].push_back(vi); break;
                case K_SAMPLE_MULTI: fam_v[F_MULTI].push_back(vi); break;
                case K_SAMPLE_LERP: fam_v[F_LERP].push_back(vi); break;
                case K_DEBUG_SINE: fam_v[F_SINE].push_back(vi); if (v.probe) fam_v[F_PROBE].push_back(vi); break;
                case K_SYNTH: fam_v[F_SYNTH].push_back(vi); if (v.probe) fam_v[F_PROBE].push_back(vi); break;
                case K_SAMPSYN: fam_v[F_SAMPSYN].push_back(vi); break;
                case K_SUM: fam_v[F_SUM].push_back(vi); break;
                case K_NORMALIZE:
                    fam_v[F_SUM].push_back(vi);
                    norm_pending.push_back(vi);   // (F_SCALE or F_NORMFIX: decided below, once the term modes are known)
                    break;
                case K_ADSR: fam_v[(v.wet < 0.0001f) ? F_SUM : F_ADSR].push_back(vi); break;
                case K_BAND_PASS:
                    if (v.wet < 0.0001f || (v.lgamma == 0.0f && v.hgamma == 0.0f)) {
                        fam_v[F_SUM].push_back(vi);   // extensions.rs:657-658: the summed input passes through
                    } else if (scan_plan.count(vi)) {
                        fam_v[F_BAND_SCAN].push_back(vi);
                    } else {
                        BandPlan bp = plan_band(g, v, M);
                        if (bp.parallel) {
                            bp.tmp = nullptr;   // (round 6: the band-pass kernels read the planar copy only -- 23 MB written and 23 MB read less per stage)
                            bp.tmpq = take_buffer(g);
                            if (!bp.tmpq) return fail("termdaw_amd: out of device memory for edge buffers");
                            level_tmp.push_back(bp.tmpq);
                            band_plan[vi] = bp;
                            fam_v[F_BAND_SPEC].push_back(vi);
                        } else {
                            fam_v[F_BAND].push_back(vi);
                        }
                    }
                    break;
                default: break;
            }
        }
        // input term tables: an edge buffer, or an inlined sample_loop source gathered by the consumer
        std::map<size_t, size_t> ins_off;
        std::map<size_t, uint32_t> term_mode;
        for (size_t vl : by_level[lv]) {
            if (!g->vertices[vl].has_input() || inlined[vl] || fused_norm.count(vl)) continue;   // (an inlined vertex' terms belong to its consumers)
            // (the launch of a band-pass chain sits at its last vertex and evaluates the first vertex' input terms)
            const size_t vi = chain_of.count(vl) ? chain_of[vl][0] : vl;
            std::vector<InTerm> ins;
            std::vector<std::pair<size_t, size_t>> adsr_through;   // (term index, the Adsr vertex a kind-5 term reads through)
            // a term of kind 0 .. 4: vertex u as an edge buffer, through a gain / pan stage, or as an inlined loop source
            auto plain_term = [&](size_t u) {
                InTerm t{};
                if (inlined[u] == 2) {   // single-input Sum stage, read through
                    t.p = g->vbuf[g->edges[u][0]];
                    t.kind = 4u;
                    t.pg = make_pg(g->vertices[u].gain, g->vertices[u].angle);
                } else if (inlined[u]) {
                    const Vertex& src = g->vertices[u];
                    const SampleEntry& s = sb->samples[src.sample_index];
                    t.p = s.d;
                    t.len = s.len;
                    t.t0 = vt[u].t0;
                    t.pg = make_pg(src.gain, src.angle);
                    const bool fits32 = s.len <= 0xFFFFFFFFull && t.t0 + M + kTileFrames <= 0xFFFFFFFFull;
                    t.kind = fits32 ? 1u : 2u;
                    t.magic = fits32 ? (s.len >= 2 ? (uint32_t)(0x100000000ull / s.len) : 0xFFFFFFFFu) : 0u;
                    if (fits32 && s.d16 && g->packed_samples) {   // half the gather bytes, same values
                        t.kind = 3u;
                        t.p = reinterpret_cast<const float2*>(s.d16);
                        t.scale_l = s.scale_l;
                        t.scale_r = s.scale_r;
                    }
                } else {
                    t.p = g->vbuf[u];
                }
                return t;
            };
            const std::vector<size_t>& in_edges = presum_of.count(vi) ? g->edges[presum_of[vi]] : g->edges[vi];
            for (size_t u : in_edges) {
                InTerm t{};
                if (inlined[u] == 3 || inlined[u] == 4) {   // Adsr vertex (and the stage behind it), evaluated here
                    const size_t a = inlined[u] == 4 ? g->edges[u][0] : u;
                    adsr_through.push_back({ins.size(), a});
                    t.kind = 5u;
                    if (inlined[u] == 4) {
                        t.magic = 1u;
                        t.pg = make_pg(g->vertices[u].gain, g->vertices[u].angle);
                    }
                } else {
                    t = plain_term(u);
                }
                ins.push_back(t);
            }
            bool all_edge = true, all_loop = !ins.empty(), all_loop16 = !ins.empty();
            for (auto& t : ins) {
                all_edge = all_edge && t.kind == 0;
                all_loop = all_loop && t.kind == 1;
                all_loop16 = all_loop16 && t.kind == 3;
            }
            // (the band-pass pre-sum keeps the pair mapping: its block-liveness reduction is written for it)
            if (all_loop16 && g->vertices[vi].kind == K_BAND_PASS) all_loop16 = false;
            term_mode[vi] = all_edge ? (ins.size() < 8 ? TERMS_EDGE_FEW : TERMS_ALL_EDGE)
                                     : (all_loop16 ? TERMS_ALL_LOOP16 : (all_loop ? TERMS_ALL_LOOP32 : TERMS_MIXED));
            ins_off[vi] = st.put(ins);
            if (!adsr_through.empty()) term_mode[vi] = ins.size() == 1 ? TERMS_ADSR1 : TERMS_WITH_ADSR;
            for (const auto& th : adsr_through) {   // the vertex' descriptor, as k_adsr would get it
                const size_t a = th.second;
                const Vertex& av = g->vertices[a];
                AdsrVDesc x{};
                x.tab.n_int = vt[a].n_int;
                x.k = 1u;   // its one input, as a term table of its own (kind 0 .. 4)
                x.sr = (uint32_t)sr;
                x.bl = (uint32_t)bl;
                x.use_off = av.use_off;
                x.use_max = av.use_max;
                x.wet = av.wet;
                x.conf = av.conf;
                x.pg = make_pg(av.gain, av.angle);
                adsr_fill_run_consts(&x);
                x.env = env_of[a];
                const size_t in_off = st.put(std::vector<InTerm>{plain_term(g->edges[a][0])});
                const size_t o = st.alloc(sizeof x);
                memcpy(&st.b[o], &x, sizeof x);
                ptr_field(o, offsetof(AdsrVDesc, ins), in_off);
                const size_t t = o + offsetof(AdsrVDesc, tab);
                tab_field(t, offsetof(IntervalTab, istart), vt[a], vt[a].istart_off);
                tab_field(t, offsetof(IntervalTab, tile_first), vt[a], vt[a].tile_first_off);
                tab_field(t, offsetof(IntervalTab, ivoff), vt[a], vt[a].ivoff_off);
                tab_field(t, offsetof(IntervalTab, voices), vt[a], vt[a].voices_off);
                ptr_field(ins_off[vi] + th.first * sizeof(InTerm), offsetof(InTerm, len), o);
            }
        }
        for (size_t vi : norm_pending) {
            const Vertex& v = g->vertices[vi];
            uint32_t mode = 1u;
            if (g->spec_normalize && !is_scan) {
                // after a normalize scan the peak is known: one pass + a (normally empty) fix launch instead of two (mode 3);
                // a FRESH render whose sum runs in the wide all-loop kernels (k_sum16w: every tile resident at once) finds the
                // running peak through granules inside that one launch (mode 4)
                const uint32_t tm = term_mode[vi];
                const bool wide = (tm == TERMS_ALL_LOOP16 || tm == TERMS_ALL_LOOP32) && bl == (size_t)kTileFrames &&
                                  M >= (size_t)1800 * kTileFrames && M < ((size_t)1 << 31);
                if (v.peak_known && !v.has_init_override) mode = 3u;
                else if (!wide && g->single_pass_normalize && bl == (size_t)kTileFrames && M < ((size_t)1 << 31)) {
                    // any other input terms, any timeline whose grid is resident at once: k_norm1
                    const int tpw = norm1_tiles_per_workgroup(tm, (uint32_t)M);
                    if (tpw) { mode = 5u; norm_tpw[vi] = tpw; }
                } else if (wide && g->single_pass_normalize) {
                    // (the kernel form launch_sum will pick: 16 frames per lane from 2 600 tiles on, packed sources only)
                    const bool packed = tm == TERMS_ALL_LOOP16;
                    const int nq = (packed && M >= (size_t)2600 * kTileFrames) ? 4 : 2;
                    const size_t gx = (M + (size_t)kTileFrames * nq - 1) / ((size_t)kTileFrames * nq);
                    mode = (size_t)sum16w_resident_capacity(nq, packed) >= gx ? 5u : 4u;
                }
            }
            norm_mode[vi] = mode;
            // (modes 3 / 4 / 5 all have k_norm_fix behind them; for 4 / 5 on the output vertex of a one-chunk render it is not
            // enqueued but kept for settle(): nothing in the submission reads the vertex' frames or its carried max)
            fam_v[mode == 1u ? F_SCALE : F_NORMFIX].push_back(vi);
        }
        std::map<size_t, std::pair<size_t, size_t>> norm_scratch;   // vi -> (peaks, init snapshot)
        std::map<size_t, SumDesc> sum_desc_of;                      // Normalize vertices: their k_sum descriptor (k_norm_fix reuses it)
        size_t band_desc_off = 0;
        uint32_t max_nseg = 0;
        for (int fam = 0; fam < F_COUNT; ++fam) {
            auto& vs = fam_v[fam];
            if (fam == F_BAND_FILL) continue;   // (the parked stretches' output is filled in by k_band_fix itself)
            if (fam == F_BAND_FIX) vs = fam_v[F_BAND_SPEC];   // same vertices, follow-up launch
            if (vs.empty() && !(fam == F_SUM && !fam_v[F_BAND_SPEC].empty())) continue;
            size_t off = 0;
            switch (fam) {
                case F_LOOP: {
                    std::vector<LoopDesc> d;
                    for (size_t vi : vs) {
                        const Vertex& v = g->vertices[vi];
                        const SampleEntry& s = sb->samples[v.sample_index];
                        const bool fits32 = s.len <= 0xFFFFFFFFull && vt[vi].t0 + M + kTileFrames <= 0xFFFFFFFFull;
                        const uint32_t magic = fits32 ? (s.len >= 2 ? (uint32_t)(0x100000000ull / s.len) : 0xFFFFFFFFu) : 0u;
                        d.push_back({s.d, g->vbuf[vi], s.len, vt[vi].t0, magic, {0, 0, 0}, make_pg(v.gain, v.angle)});
                    }
                    off = st.put(d);
                } break;
                case F_MULTI: {
                    std::vector<MultiDesc> d;
                    for (size_t vi : vs) {
                        const Vertex& v = g->vertices[vi];
                        const SampleEntry& s = sb->samples[v.sample_index];
                        d.push_back({s.d, g->vbuf[vi], nullptr, s.len, vt[vi].n_hits, 0, make_pg(v.gain, v.angle), nullptr});
                    }
                    off = st.put(d);
                    for (size_t i = 0; i < vs.size(); ++i) {
                        tab_field(off + i * sizeof(MultiDesc), offsetof(MultiDesc, hits), vt[vs[i]], vt[vs[i]].hits_off);
                        tab_field(off + i * sizeof(MultiDesc), offsetof(MultiDesc, tile_first), vt[vs[i]], vt[vs[i]].tile_first_off);
                    }
                } break;
                case F_LERP: {
                    std::vector<LerpDesc> d;
                    for (size_t vi : vs) {
                        const Vertex& v = g->vertices[vi];
                        const SampleEntry& s = sb->samples[v.sample_index];
                        d.push_back({s.d, g->vbuf[vi], nullptr, s.len, vt[vi].n_hits, (uint32_t)v.lerp_len,
                                     make_pg(v.gain, v.angle), nullptr});
                    }
                    off = st.put(d);
                    for (size_t i = 0; i < vs.size(); ++i) {
                        tab_field(off + i * sizeof(LerpDesc), offsetof(LerpDesc, hits), vt[vs[i]], vt[vs[i]].hits_off);
                        tab_field(off + i * sizeof(LerpDesc), offsetof(LerpDesc, tile_first), vt[vs[i]], vt[vs[i]].tile_first_off);
                    }
                } break;
                case F_SINE: {
                    std::vector<SineDesc> d;
                    for (size_t vi : vs) {
                        const Vertex& v = g->vertices[vi];
                        SineDesc x{};
                        x.tab.n_int = vt[vi].n_int;
                        x.out = g->vbuf[vi];
                        x.t0 = t0;
                        x.sr = (uint32_t)sr;
                        x.exact_sin = v.exact_sin ? 1u : 0u;
                        x.pg = make_pg(v.gain, v.angle);
                        d.push_back(x);
                    }
                    off = st.put(d);
                    for (size_t i = 0; i < vs.size(); ++i) {
                        const size_t o = off + i * sizeof(SineDesc) + offsetof(SineDesc, tab);
                        tab_field(o, offsetof(IntervalTab, istart), vt[vs[i]], vt[vs[i]].istart_off);
                        tab_field(o, offsetof(IntervalTab, tile_first), vt[vs[i]], vt[vs[i]].tile_first_off);
                        tab_field(o, offsetof(IntervalTab, ivoff), vt[vs[i]], vt[vs[i]].ivoff_off);
                        tab_field(o, offsetof(IntervalTab, voices), vt[vs[i]], vt[vs[i]].voices_off);
                    }
                } break;
                case F_SYNTH: {
                    std::vector<SynthDesc> d;
                    // (k_synth and k_synth_affine are two kernels: the generic vertices first, the affine ones behind them)
                    std::stable_sort(vs.begin(), vs.end(), [&](size_t a, size_t b) { return synth_affine_ok(g->vertices[a]) < synth_affine_ok(g->vertices[b]); });
                    for (size_t vi : vs) d.push_back(synth_desc_of(vi));
                    off = st.put(d);
                    for (size_t i = 0; i < vs.size(); ++i) {
                        const size_t o = off + i * sizeof(SynthDesc) + offsetof(SynthDesc, tab);
                        tab_field(o, offsetof(IntervalTab, tile_order), vt[vs[i]], vt[vs[i]].tile_order_off);
                        tab_field(o, offsetof(IntervalTab, istart), vt[vs[i]], vt[vs[i]].istart_off);
                        tab_field(o, offsetof(IntervalTab, tile_first), vt[vs[i]], vt[vs[i]].tile_first_off);
                        tab_field(o, offsetof(IntervalTab, ivoff), vt[vs[i]], vt[vs[i]].ivoff_off);
                        tab_field(o, offsetof(IntervalTab, voices), vt[vs[i]], vt[vs[i]].voices_off);
                    }
                    size_t n_gen = 0;
                    while (n_gen < vs.size() && !d[n_gen].affine) ++n_gen;
                    if (n_gen) add_launch(fam, off, (int)n_gen, 0u, lv);
                    if (n_gen < vs.size()) add_launch(fam, off + n_gen * sizeof(SynthDesc), (int)(vs.size() - n_gen), 1u, lv);
                    continue;
                }
                case F_SAMPSYN: {
                    std::vector<SampsynDesc> d;
                    for (size_t vi : vs) {
                        const Vertex& v = g->vertices[vi];
                        SampsynDesc x{};
                        x.tab.n_int = vt[vi].n_int;
                        x.out = g->vbuf[vi];
                        x.wt = v.wavetable;
                        x.sr = (uint32_t)sr;
                        x.bl = (uint32_t)bl;
                        x.adsr = v.conf;
                        x.amp_multiplier = 1.0f / adsr_max_vel(v.conf);   // extensions.rs:537
                        x.pg = make_pg(v.gain, v.angle);
                        d.push_back(x);
                    }
                    off = st.put(d);
                    for (size_t i = 0; i < vs.size(); ++i) {
                        const size_t o = off + i * sizeof(SampsynDesc) + offsetof(SampsynDesc, tab);
                        tab_field(o, offsetof(IntervalTab, tile_order), vt[vs[i]], vt[vs[i]].tile_order_off);
                        tab_field(o, offsetof(IntervalTab, istart), vt[vs[i]], vt[vs[i]].istart_off);
                        tab_field(o, offsetof(IntervalTab, tile_first), vt[vs[i]], vt[vs[i]].tile_first_off);
                        tab_field(o, offsetof(IntervalTab, ivoff), vt[vs[i]], vt[vs[i]].ivoff_off);
                        tab_field(o, offsetof(IntervalTab, voices), vt[vs[i]], vt[vs[i]].voices_off);
                    }
                } break;
                case F_PROBE: {   // (behind F_SINE / F_SYNTH of this level: their buffers are assigned)
                    std::vector<ProbeDesc> d;
                    for (size_t vi : vs) {
                        const Vertex& v = g->vertices[vi];
                        ProbeDesc x{};
                        x.kind = v.kind == K_SYNTH ? 1u : 0u;
                        x.stride_log2 = probe_lg;
                        x.n_groups = probe_groups;
                        if (x.kind) {
                            x.syn = synth_desc_of(vi);
                            x.syn.affine = 0u;
                            x.syn.exact_sin = 1u;
                            x.syn.tab.n_int = vt[vi].raw_n_int;
                        } else {
                            x.sine.tab.n_int = vt[vi].n_int;
                            x.sine.out = g->vbuf[vi];
                            x.sine.t0 = t0;
                            x.sine.sr = (uint32_t)sr;
                            x.sine.exact_sin = 1u;
                            x.sine.pg = make_pg(v.gain, v.angle);
                        }
                        d.push_back(x);
                    }
                    off = st.put(d);
                    for (size_t i = 0; i < vs.size(); ++i) {
                        const VTables& t = vt[vs[i]];
                        const size_t po = off + i * sizeof(ProbeDesc);
                        if (d[i].kind) {
                            const size_t o = po + offsetof(ProbeDesc, syn) + offsetof(SynthDesc, tab);
                            tab_field(o, offsetof(IntervalTab, istart), t, t.raw_istart_off);
                            tab_field(o, offsetof(IntervalTab, tile_first), t, t.raw_tile_first_off);
                            tab_field(o, offsetof(IntervalTab, ivoff), t, t.raw_ivoff_off);
                            tab_field(o, offsetof(IntervalTab, voices), t, t.raw_voices_off);
                        } else {
                            const size_t o = po + offsetof(ProbeDesc, sine) + offsetof(SineDesc, tab);
                            tab_field(o, offsetof(IntervalTab, istart), t, t.istart_off);
                            tab_field(o, offsetof(IntervalTab, tile_first), t, t.tile_first_off);
                            tab_field(o, offsetof(IntervalTab, ivoff), t, t.ivoff_off);
                            tab_field(o, offsetof(IntervalTab, voices), t, t.voices_off);
                        }
                        tab_field(po, offsetof(ProbeDesc, ranges), t, t.probe_v_off);
                        const size_t no = scratch((size_t)probe_groups * 16 * sizeof(float));
                        scratch_field(po, offsetof(ProbeDesc, noise), no);
                        probe_src.push_back({vs[i], no});
                        probe_desc_off.push_back(po);
                    }
                    add_launch(fam, off, (int)vs.size(), probe_groups, lv);
                    continue;
                }
                case F_SUM: {
                    std::vector<SumDesc> d;
                    // parallel band-pass vertices first get their summed input materialised (no epilogue)
                    for (size_t vi : fam_v[F_BAND_SPEC]) vs.push_back(vi);
                    // one launch per term mode (k_sum is instantiated per mode): group the vertices by it
                    // (a single-pass running-peak Normalize -- mode 4 -- only exists in the wide kernels: a group of its own)
                    auto sum_key = [&](size_t vi) {
                        return term_mode[vi] * 16u + ((norm_mode.count(vi) && norm_mode[vi] >= 4u) ? 1u : 0u) + (norm_tpw.count(vi) ? 2u * (uint32_t)norm_tpw[vi] : 0u);
                    };
                    std::stable_sort(vs.begin(), vs.end(), [&](size_t a, size_t b) { return sum_key(a) < sum_key(b); });
                    for (size_t vi : vs) {
                        const Vertex& v = g->vertices[vi];
                        SumDesc x{};
                        const bool presum = band_plan.count(vi) != 0;
                        x.out = presum ? band_plan[vi].tmp : g->vbuf[vi];
                        x.out_q4 = presum ? band_plan[vi].tmpq : nullptr;
                        x.k = (uint32_t)g->edges[vi].size();
                        const bool spec = v.kind == K_NORMALIZE && norm_mode[vi] != 1u;
                        x.mode = v.kind == K_NORMALIZE ? norm_mode[vi] : (presum ? 2u : 0u);
                        if (spec && (long)vi == g->output_vertex && pcm_dst && qmode) {
                            x.pcm = pcm_dst;
                            x.amplitude = amplitude;
                            x.qmode = (uint32_t)qmode;
                            if (!g->output_f32) x.out = nullptr;
                        }
                        x.term_mode = term_mode[vi];
                        x.debug = (uint32_t)g->norm_debug;
                        x.pg = presum ? PanGain{1.0f, 1.0f, 1.0f, 0u} : make_pg(v.gain, v.angle);
                        if (presum && band_plan[vi].Wq) {   // block responses for the warm-up guess
                            BandPlan& bp = band_plan[vi];
                            BandRespParam rp{};
                            double ql = 1.0 - (double)v.lgamma, qh = 1.0 - (double)v.hgamma;
                            for (int j = 0; j < 8; ++j) { rp.ql[j] = ql; rp.qh[j] = qh; ql *= ql; qh *= qh; }
                            rp.gl = bp.Kl ? (double)v.lgamma : 0.0;   // (0: no responses for that smoother)
                            rp.gh = bp.Kh ? (double)v.hgamma : 0.0;
                            bp.rp_off = st.alloc(sizeof rp);
                            memcpy(&st.b[bp.rp_off], &rp, sizeof rp);
                            bp.resp_off = scratch(((M + 255) / 256) * 4 * sizeof(double));
                            scratch_field(bp.rp_off, offsetof(BandRespParam, resp), bp.resp_off);
                        }
                        if (v.kind == K_NORMALIZE) {
                            x.state = &g->dstate[v.state_slot].norm;
                            x.use_init = v.has_init_override ? 1u : 0u;   // reset_normalization consumed here
                            x.init_max = v.init_override;
                        }
                        d.push_back(x);
                    }
                    off = st.put(d);
                    for (size_t i = 0; i < vs.size(); ++i) {
                        const size_t o = off + i * sizeof(SumDesc);
                        ptr_field(o, offsetof(SumDesc, ins), ins_off[vs[i]]);
                        if (g->vertices[vs[i]].kind == K_NORMALIZE) {
                            const size_t pk = scratch(nb * sizeof(float)), ic = scratch(2 * sizeof(float));
                            norm_scratch[vs[i]] = {pk, ic};
                            audit_norm[vs[i]] = {pk, ic};
                            sum_desc_of[vs[i]] = d[i];
                            if (d[i].mode >= 4u) {   // one granule per workgroup (at most one per block)
                                cb.esync_fix.push_back({o + offsetof(SumDesc, sync), cb.esync_bytes});
                                cb.esync_bytes += (nb * 8 + 63) & ~(size_t)63;
                                cb.flag_fix.push_back(o + offsetof(SumDesc, host_flag));
                            }
                            scratch_field(o, offsetof(SumDesc, peaks), pk);
                            scratch_field(o, offsetof(SumDesc, init_copy), ic);
                            if (peaks_need_zero) cb.zero.push_back({pk, nb * sizeof(float)});
                            g->vertices[vs[i]].has_init_override = false;
                        } else if (band_plan.count(vs[i])) {
                            const size_t bpk = scratch(((M + 255) / 256) * sizeof(float));
                            band_plan[vs[i]].blk_peaks_off = bpk;
                            scratch_field(o, offsetof(SumDesc, peaks), bpk);
                            if (band_plan[vs[i]].Wq) ptr_field(o, offsetof(SumDesc, rp), band_plan[vs[i]].rp_off);
                        }
                    }
                } break;
                case F_SCALE: {
                    std::vector<ScaleDesc> d;
                    for (size_t vi : vs) {
                        const Vertex& v = g->vertices[vi];
                        const bool is_out = (long)vi == g->output_vertex && pcm_dst && qmode;
                        ScaleDesc x{};
                        x.buf = g->vbuf[vi];
                        x.state = &g->dstate[v.state_slot].norm;
                        x.pcm = is_out ? pcm_dst : nullptr;
                        x.amplitude = amplitude;
                        x.qmode = is_out ? (uint32_t)qmode : 0u;
                        x.pg = make_pg(v.gain, v.angle);
                        x.pcm_only = (is_out && !g->output_f32) ? 1u : 0u;
                        d.push_back(x);
                    }
                    off = st.put(d);
                    for (size_t i = 0; i < vs.size(); ++i) {
                        scratch_field(off + i * sizeof(ScaleDesc), offsetof(ScaleDesc, peaks), norm_scratch[vs[i]].first);
                        scratch_field(off + i * sizeof(ScaleDesc), offsetof(ScaleDesc, init_copy), norm_scratch[vs[i]].second);
                    }
                } break;
                case F_NORMFIX: {   // the same descriptors the speculative k_sum launch got
                    // deferred (kept for settle(), not launched): a mode 4 / 5 output vertex of a one-chunk render -- last in `vs`
                    auto deferred = [&](size_t vi) { return g->defer_fix && norm_mode[vi] >= 4u && (long)vi == g->output_vertex; };
                    std::stable_sort(vs.begin(), vs.end(), [&](size_t a, size_t b) { return deferred(a) < deferred(b); });
                    std::vector<SumDesc> d;
                    for (size_t vi : vs) d.push_back(sum_desc_of[vi]);
                    off = st.put(d);
                    size_t n_now = 0;
                    for (size_t i = 0; i < vs.size(); ++i) {
                        const size_t o = off + i * sizeof(SumDesc);
                        ptr_field(o, offsetof(SumDesc, ins), ins_off[vs[i]]);
                        scratch_field(o, offsetof(SumDesc, peaks), norm_scratch[vs[i]].first);
                        scratch_field(o, offsetof(SumDesc, init_copy), norm_scratch[vs[i]].second);
                        if (!deferred(vs[i])) ++n_now;
                    }
                    if (n_now) add_launch(fam, off, (int)n_now, 0u, lv);
                    if (n_now < vs.size()) add_launch(fam, off + n_now * sizeof(SumDesc), (int)(vs.size() - n_now), 1u, lv);
                    continue;
                }
                case F_ADSR: {
                    // (k_adsr is instantiated per term mode like k_sum; its pair mapping has no packed-loop form)
                    for (size_t vi : vs)
                        if (term_mode[vi] == TERMS_ALL_LOOP16) term_mode[vi] = TERMS_MIXED;
                    std::stable_sort(vs.begin(), vs.end(), [&](size_t a, size_t b) { return term_mode[a] < term_mode[b]; });
                    std::vector<AdsrVDesc> d;
                    for (size_t vi : vs) {
                        const Vertex& v = g->vertices[vi];
                        AdsrVDesc x{};
                        x.tab.n_int = vt[vi].n_int;
                        x.out = g->vbuf[vi];
                        x.k = (uint32_t)g->edges[vi].size();
                        x.sr = (uint32_t)sr;
                        x.bl = (uint32_t)bl;
                        x.use_off = v.use_off;
                        x.use_max = v.use_max;
                        x.term_mode = term_mode[vi];
                        x.wet = v.wet;
                        x.conf = v.conf;
                        x.pg = make_pg(v.gain, v.angle);
                        adsr_fill_run_consts(&x);
                        d.push_back(x);
                    }
                    off = st.put(d);
                    for (size_t i = 0; i < vs.size(); ++i) {
                        const size_t o = off + i * sizeof(AdsrVDesc);
                        ptr_field(o, offsetof(AdsrVDesc, ins), ins_off[vs[i]]);
                        const size_t t = o + offsetof(AdsrVDesc, tab);
                        tab_field(t, offsetof(IntervalTab, istart), vt[vs[i]], vt[vs[i]].istart_off);
                        tab_field(t, offsetof(IntervalTab, tile_first), vt[vs[i]], vt[vs[i]].tile_first_off);
                        tab_field(t, offsetof(IntervalTab, ivoff), vt[vs[i]], vt[vs[i]].ivoff_off);
                        tab_field(t, offsetof(IntervalTab, voices), vt[vs[i]], vt[vs[i]].voices_off);
                    }
                } break;
                case F_BAND: {
                    std::vector<BandDesc> d;
                    for (size_t vi : vs) {
                        Vertex& v = g->vertices[vi];
                        BandDesc x{};
                        x.out = g->vbuf[vi];
                        x.state = &g->dstate[v.state_slot].band;
                        x.first_override = v.first_pending ? 1u : 0u;   // (consumed here, like a Normalize vertex' init override)
                        v.first_pending = false;
                        x.k = (uint32_t)g->edges[vi].size();
                        x.term_mode = term_mode[vi];
                        x.pass = v.pass;
                        x.lgamma = v.lgamma;
                        x.hgamma = v.hgamma;
                        x.pg = make_pg(v.gain, v.angle);
                        d.push_back(x);
                    }
                    off = st.put(d);
                    for (size_t i = 0; i < vs.size(); ++i)
                        ptr_field(off + i * sizeof(BandDesc), offsetof(BandDesc, ins), ins_off[vs[i]]);
                } break;
                case F_BAND_SPEC: {
                    std::vector<BandSpecDesc> d;
                    for (size_t vi : vs) {
                        const Vertex& v = g->vertices[vi];
                        const BandPlan& bp = band_plan[vi];
                        BandSpecDesc x{};
                        x.x = bp.tmp;
                        x.xq4 = bp.tmpq;
                        x.out = g->vbuf[vi];
                        x.state = &g->dstate[v.state_slot].band;
                        x.first_override = g->vertices[vi].first_pending ? 1u : 0u;
                        g->vertices[vi].first_pending = false;
                        x.nseg = bp.nseg;
                        x.S = bp.S;
                        x.W = bp.W;
                        x.Ws = bp.Ws;
                        x.live_thr = g->band_live_thr;
                        {
                            float gmin = 1.0f;
                            if (v.lgamma != 0.0f) gmin = fminf(gmin, fabsf(v.lgamma));
                            if (v.hgamma != 0.0f) gmin = fminf(gmin, fabsf(v.hgamma));
                            x.gmin = gmin;
                            x.decay1 = (float)exp(-(double)gmin * 256.0);
                            x.decay4 = (float)exp(-(double)gmin * 1024.0);
                            x.post_blocks = (uint32_t)(20.0 / (double)gmin / 256.0) + 1u;
                        }
                        x.pass = v.pass;
                        x.lgamma = v.lgamma;
                        x.hgamma = v.hgamma;
                        x.pg = make_pg(v.gain, v.angle);
                        x.Wq = bp.Wq;
                        x.Wq2 = bp.Wq2;
                        x.quick_thr = fminf(1.0f, g->band_live_thr * 1.0e5f);
                        x.Al = bp.Al; x.Ah = bp.Ah;
                        x.Kl = bp.Kl; x.Kh = bp.Kh;
                        d.push_back(x);
                        max_nseg = std::max(max_nseg, bp.nseg);
                    }
                    off = st.put(d);
                    band_desc_off = off;
                    for (size_t i = 0; i < vs.size(); ++i) {
                        const size_t o = off + i * sizeof(BandSpecDesc);
                        const uint32_t ns = band_plan[vs[i]].nseg;
                        scratch_field(o, offsetof(BandSpecDesc, seg_start), scratch((size_t)ns * 16));
                        scratch_field(o, offsetof(BandSpecDesc, seg_final), scratch((size_t)ns * 16));
                        scratch_field(o, offsetof(BandSpecDesc, seg_flags), scratch((size_t)ns * 4));
                        scratch_field(o, offsetof(BandSpecDesc, blk_peaks), band_plan[vs[i]].blk_peaks_off);
                        if (band_plan[vs[i]].Wq) scratch_field(o, offsetof(BandSpecDesc, resp), band_plan[vs[i]].resp_off);
                        scratch_field(o, offsetof(BandSpecDesc, seg_x0), scratch((size_t)ns * 8));
                        scratch_field(o, offsetof(BandSpecDesc, jobs), scratch((size_t)ns * sizeof(BandJob)));
                        scratch_field(o, offsetof(BandSpecDesc, seg_job), scratch((size_t)ns * 4));
                        const size_t so = scratch(256);   // counters [0..7], verdict + fill claims on a line of their own [32..33]
                        scratch_field(o, offsetof(BandSpecDesc, stats), so);
                        g->band_stats_off.push_back(so);
                    }
                } break;
                case F_BAND_FIX:
                case F_BAND_FILL: off = band_desc_off; break;   // reuse the k_band_spec descriptors
                case F_BAND_SCAN: {
                    // vs: the vertices whose launch sits here -- single band-pass vertices and the LAST vertices of chains
                    auto first_of = [&](size_t vi) { return chain_of.count(vi) ? chain_of[vi][0] : vi; };
                    for (size_t vi : vs) {
                        uint32_t& tm = term_mode[first_of(vi)];
                        if (tm == TERMS_ALL_LOOP16 || tm == TERMS_ALL_LOOP32) tm = TERMS_MIXED;
                    }
                    auto launch_key = [&](size_t vi) { return term_mode[first_of(vi)] | (chain_of.count(vi) ? 0x10000u : 0u); };
                    std::stable_sort(vs.begin(), vs.end(), [&](size_t a, size_t b) { return launch_key(a) < launch_key(b); });
                    std::vector<BandScanDesc> d;
                    std::vector<size_t> stages_off, norm_desc_off;
                    for (size_t vi : vs) {
                        const std::vector<size_t> piece = chain_of.count(vi) ? chain_of[vi] : std::vector<size_t>{vi};
                        const ScanPlan& sp0 = scan_plan[piece[0]];
                        std::vector<BandStageDesc> sd;
                        std::vector<double> stage_gain;   // (guard) per stage: its own pan / gain and the static part of the links behind it
                        std::vector<const float*> stage_env;   // ... and the envelope link behind it, if any
                        for (size_t i = 0; i < piece.size(); ++i) {
                            const Vertex& v = g->vertices[piece[i]];
                            ScanPlan& sp = scan_plan[piece[i]];
                            BandStageDesc x{};
                            x.state = &g->dstate[v.state_slot].band;
                            x.first_override = v.first_pending ? 1u : 0u;
                            g->vertices[piece[i]].first_pending = false;
                            x.lgamma = v.lgamma;
                            x.hgamma = v.hgamma;
                            x.pass = v.pass;
                            x.K = sp.K;
                            x.pg = make_pg(v.gain, v.angle);
                            const double nf = (double)sp.nf;
                            const double q[2] = {1.0 - (double)v.lgamma, 1.0 - (double)v.hgamma};
                            for (int c = 0; c < 2; ++c) {
                                for (int s2 = 0; s2 < 6; ++s2) x.ap[c][s2] = pow(q[c], nf * (double)(1 << s2));
                                x.aw[c] = pow(q[c], nf * 64.0);
                                x.at[c] = pow(q[c], nf * 256.0);
                            }
                            std::string key;
                            put_pod(key, v.lgamma); put_pod(key, v.hgamma); put_pod(key, sp.nf);
                            auto it = scan_pw_off.find(key);
                            if (it == scan_pw_off.end()) {
                                std::vector<double> pw(128);
                                for (int c = 0; c < 2; ++c)
                                    for (int l = 0; l < 64; ++l) pw[c * 64 + l] = pow(q[c], nf * (double)l);
                                it = scan_pw_off.emplace(key, st.put(pw)).first;
                            }
                            sp.pw_off = it->second;
                            if (chain_of.count(vi)) {   // k_band_chain: per-frame and per-wave-tile powers
                                for (int c = 0; c < 2; ++c)
                                    for (int n2 = 0; n2 < 16; ++n2) x.pn[n2][c] = (float)pow(q[c], (double)(n2 + 1));
                                x.Kw = sp.Kw;
                                key.push_back('k');
                                auto ik = scan_pw_off.find(key);
                                if (ik == scan_pw_off.end()) {
                                    std::vector<double> pk(2 * kScanMaxK);
                                    for (int c = 0; c < 2; ++c)
                                        for (uint32_t j = 0; j < kScanMaxK; ++j) pk[c * kScanMaxK + j] = pow(q[c], nf * 256.0 * (double)j);
                                    ik = scan_pw_off.emplace(key, st.put(pk)).first;
                                }
                                sp.pk_off = ik->second;
                            }
                            double link_gain = own_gain(v);
                            const float* stage_env_link = nullptr;
                            if (i + 1 < piece.size() || norm_of.count(vi)) {   // the links to the next vertex of the chain / to the Normalize vertex
                                const std::vector<ChainLink>& links = i + 1 < piece.size() ? links_before[piece[i + 1]] : links_after[vi];
                                x.n_post = (uint32_t)links.size();
                                for (size_t l = 0; l < links.size(); ++l) {
                                    const Vertex& lv2 = g->vertices[links[l].vertex];
                                    x.post[l].env = links[l].adsr ? env_of[links[l].vertex] : nullptr;
                                    x.post[l].pg = make_pg(lv2.gain, lv2.angle);
                                    link_gain *= own_gain(lv2);
                                    if (links[l].adsr && guard_on) stage_env_link = env_of[links[l].vertex];   // (at most one Adsr vertex per hop)
                                }
                            }
                            if (guard_on) {   // (kernels.h BandStageDesc::nzv ..: DESIGN.md 3e "The guard")
                                const double K0 = 2.53e-8 * 2.53e-8;   // variance of one rounding of a state of unit level: E[ulp^2] / 12 over a binade
                                const float gm[2] = {v.lgamma, v.hgamma};
                                for (int c = 0; c < 2; ++c) {
                                    const double gmc = (double)gm[c];
                                    x.nzv[c] = gm[c] == 0.0f ? 0.0f : (float)(0.25 * K0 / (gmc * (2.0 - gmc)));
                                    x.nzs[c] = gm[c] == 0.0f ? 0.0f : (float)(0.5 * ldexp(1.0, -24) / gmc);
                                    x.nzk[c] = gm[c] == 0.0f ? 0.0f : (float)(4.0 * ldexp(1.0, -23) / gmc);
                                }
                                // both smoothers see the same input: when the slower one is parked so is the faster, at the same level
                                // and an offset smaller by gamma_low / gamma_high -- below 5 % the faster one's test is not run
                                if (v.lgamma != 0.0f && v.hgamma != 0.0f && (double)v.lgamma < 0.05 * (double)v.hgamma) { x.nzk[1] = 0.0f; x.nzs[1] = 0.0f; }
                                stage_gain.push_back(link_gain);
                                stage_env.push_back(stage_env_link);
                            }
                            sd.push_back(x);
                        }
                        if (guard_on) {   // the static gain from every stage to the launch's last one, folded into its coefficients
                            std::vector<double> G(piece.size(), 1.0);
                            double acc = 1.0;
                            for (size_t i = piece.size(); i-- > 0;) { acc *= stage_gain[i]; G[i] = acc; }
                            // Runs of identical filters (the 84 stages of BASELINE config 4) are evaluated at their FIRST stage only,
                            // for the whole run: a later member's state level is the first one's times the gains in between, and
                            // those times its own gain to the end are the first one's gain to the end -- so every member adds what
                            // the first adds (the envelope gains in between reach the estimate in the kernel either way; they are
                            // at most 1 here, so a later member's true level is lower, never higher).  Up to 8 stages per run: every
                            // VALU instruction of the stage loop costs it ~0.2 % (kernels.hip).
                            size_t f = 0;
                            for (size_t i = 0; i < piece.size(); ++i) {
                                const Vertex& vf = g->vertices[piece[f]];
                                const Vertex& vi2 = g->vertices[piece[i]];
                                bool same = i > f && i - f < 8 && vi2.lgamma == vf.lgamma && vi2.hgamma == vf.hgamma;
                                if (same)   // (an envelope link in between that may exceed 1 ends the run)
                                    for (const ChainLink& L : links_before[piece[i]])
                                        if (L.adsr) {
                                            const Vertex& av = g->vertices[L.vertex];
                                            const AdsrConfD& c = av.conf;
                                            double lv = std::max(std::max(fabs((double)c.std_vel), fabs((double)c.attack_vel)),
                                                                 std::max(std::max(fabs((double)c.decay_vel), fabs((double)c.sustain_vel)), fabs((double)c.release_vel)));
                                            double mv = 0.0;
                                            for (const td_event& e : floww_of(fb, av.floww_index)) mv = std::max(mv, fabs((double)e.vel));
                                            if (!(lv * mv <= 1.0)) same = false;
                                        }
                                if (!same) f = i;
                                const double n_run = 1.0;   // (this member's own share; the run's first stage collects it)
                                const double wv = (double)sd[i].nzv[0], wv1 = (double)sd[i].nzv[1], ws = (double)sd[i].nzs[0], ws1 = (double)sd[i].nzs[1];
                                if (i == f) {
                                    sd[i].nzv[0] = (float)(wv * G[i] * G[i]); sd[i].nzv[1] = (float)(wv1 * G[i] * G[i]);
                                    sd[i].nzs[0] = (float)(ws * G[i]); sd[i].nzs[1] = (float)(ws1 * G[i]);
                                } else {
                                    sd[f].nzv[0] += (float)(n_run * wv * G[f] * G[f]); sd[f].nzv[1] += (float)(n_run * wv1 * G[f] * G[f]);
                                    sd[f].nzs[0] += (float)(n_run * ws * G[f]); sd[f].nzs[1] += (float)(n_run * ws1 * G[f]);
                                    sd[i].nzv[0] = sd[i].nzv[1] = sd[i].nzs[0] = sd[i].nzs[1] = 0.0f;
                                    sd[i].nzk[0] = sd[i].nzk[1] = 0.0f;
                                }
                            }
                        }
                        const size_t so = st.put(sd);
                        stages_off.push_back(so);
                        for (size_t i = 0; i < piece.size(); ++i) {
                            const size_t o = so + i * sizeof(BandStageDesc);
                            ptr_field(o, offsetof(BandStageDesc, pw), scan_plan[piece[i]].pw_off);
                            if (chain_of.count(vi)) ptr_field(o, offsetof(BandStageDesc, pk), scan_plan[piece[i]].pk_off);
                            cb.sync_fix.push_back({o + offsetof(BandStageDesc, sync), cb.sync_bytes});
                            cb.sync_bytes += (size_t)sp0.n_tiles * 128;   // 8 granules per tile, or 4 per wave-tile (chain)
                            if (guard_on && stage_env[i]) {
                                const auto et = env_tile_at.find(stage_env[i]);
                                if (et == env_tile_at.end()) return fail("termdaw_amd: internal: the guard lost an envelope buffer");
                                cb.scratch_fix.push_back({o + offsetof(BandStageDesc, envt), et->second});
                            }
                        }
                        BandScanDesc x{};
                        x.out = g->vbuf[vi];
                        x.n_stages = (uint32_t)piece.size();
                        x.k = (uint32_t)(presum_of.count(piece[0]) ? g->edges[presum_of[piece[0]]].size() : g->edges[piece[0]].size());
                        if (presum_of.count(piece[0])) {
                            const Vertex& pv = g->vertices[presum_of[piece[0]]];
                            x.pre = make_pg(pv.gain, pv.angle);
                            if (presum_stage.count(piece[0])) {
                                const Vertex& sv = g->vertices[presum_stage[piece[0]]];
                                x.pre2 = make_pg(sv.gain, sv.angle);
                            }
                        }
                        x.term_mode = term_mode[piece[0]];
                        x.n_tiles = sp0.n_tiles;
                        x.flags = (uint32_t)g->band_scan_debug;
                        x.nz_end = norm_of.count(vi) ? (float)own_gain(g->vertices[norm_of[vi]]) : 1.0f;
                        d.push_back(x);
                        if (guard_on) {   // one entry per wave-tile (the chain launch) or per workgroup tile (k_band_scan: a single vertex)
                            const bool is_chain = chain_of.count(vi) != 0;
                            const uint32_t tf = is_chain ? (uint32_t)kTileFrames : band_scan_tile_frames(sp0.nf);
                            const uint32_t n_wt = is_chain ? (uint32_t)((M + (size_t)kTileFrames - 1) / (size_t)kTileFrames) : sp0.n_tiles;
                            audit_src.push_back({scratch((size_t)n_wt * sizeof(float)), n_wt, norm_of.count(vi) ? norm_of[vi] : vi, norm_of.count(vi) != 0, 0, tf});
                        }
                        if (norm_of.count(vi)) {   // the Normalize vertex behind the launch: its descriptor as k_norm1 would get it (mode 5)
                            const size_t ni = norm_of[vi];
                            Vertex& nv = g->vertices[ni];
                            SumDesc y{};
                            y.out = g->vbuf[ni];
                            y.k = 1u;
                            y.mode = 5u;
                            if ((long)ni == g->output_vertex && pcm_dst && qmode) {
                                y.pcm = pcm_dst;
                                y.amplitude = amplitude;
                                y.qmode = (uint32_t)qmode;
                                if (!g->output_f32) y.out = nullptr;
                            }
                            y.pg = make_pg(nv.gain, nv.angle);
                            y.state = &g->dstate[nv.state_slot].norm;
                            y.use_init = nv.has_init_override ? 1u : 0u;   // reset_normalization consumed here
                            y.init_max = nv.init_override;
                            nv.has_init_override = false;
                            const size_t no = st.put(std::vector<SumDesc>{y});
                            norm_desc_off.push_back(no);
                            const size_t pk = scratch(nb * sizeof(float)), ic = scratch(2 * sizeof(float));
                            scratch_field(no, offsetof(SumDesc, peaks), pk);
                            scratch_field(no, offsetof(SumDesc, init_copy), ic);
                            audit_norm[ni] = {pk, ic};
                            if (peaks_need_zero) cb.zero.push_back({pk, nb * sizeof(float)});
                            cb.sync_fix.push_back({no + offsetof(SumDesc, sync), cb.sync_bytes});   // one granule per tile
                            cb.sync_bytes += ((size_t)sp0.n_tiles * 8 + 63) & ~(size_t)63;
                        } else {
                            norm_desc_off.push_back((size_t)-1);
                        }
                    }
                    off = st.put(d);
                    for (size_t i = 0; i < vs.size(); ++i) {
                        const size_t o = off + i * sizeof(BandScanDesc);
                        ptr_field(o, offsetof(BandScanDesc, ins), ins_off[first_of(vs[i])]);
                        ptr_field(o, offsetof(BandScanDesc, stages), stages_off[i]);
                        if (norm_desc_off[i] != (size_t)-1) ptr_field(o, offsetof(BandScanDesc, norm), norm_desc_off[i]);
                        if (guard_on) {
                            AuditSrc& as = audit_src[audit_src.size() - vs.size() + i];
                            as.desc_off = o;
                            scratch_field(o, offsetof(BandScanDesc, noise), as.noise_off);
                        }
                        cb.sync_fix.push_back({o + offsetof(BandScanDesc, ticket), cb.sync_bytes});   // {tile counter, "states read"}
                        cb.sync_bytes += 64;
                        {   // one granule per tile: the stage it went non-finite at; and the frame its right input did (k_band_chain)
                            cb.sync_fix.push_back({o + offsetof(BandScanDesc, poison), cb.sync_bytes});
                            cb.sync_bytes += ((size_t)scan_plan[vs[i]].n_tiles * 8 + 63) & ~(size_t)63;
                            cb.sync_fix.push_back({o + offsetof(BandScanDesc, rpoison), cb.sync_bytes});
                            cb.sync_bytes += ((size_t)scan_plan[vs[i]].n_tiles * 8 + 63) & ~(size_t)63;
                        }
                    }
                    size_t b = 0;   // one launch per term mode (vs is sorted by it)
                    while (b < vs.size()) {
                        size_t e2 = b;
                        while (e2 < vs.size() && launch_key(vs[e2]) == launch_key(vs[b])) ++e2;
                        add_launch(fam, off + b * sizeof(BandScanDesc), (int)(e2 - b),
                                   launch_key(vs[b]) | ((uint32_t)scan_plan[first_of(vs[b])].nf << 8) | (guard_on ? 0x20000u : 0u), lv);
                        b = e2;
                    }
                    continue;
                }
                default: continue;
            }
            if (fam == F_SUM || fam == F_ADSR) {   // split at term-mode boundaries (vs is sorted by it)
                const size_t dsz = fam == F_SUM ? sizeof(SumDesc) : sizeof(AdsrVDesc);
                size_t b = 0;
                while (b < vs.size()) {
                    size_t e2 = b;
                    bool wide_ok = true;   // (k_sum16w: plain sums, or normalize pass A with the tile as reference block)
                    auto m4 = [&](size_t vi) { return fam == F_SUM && norm_mode.count(vi) && norm_mode[vi] >= 4u; };
                    auto tpw_of = [&](size_t vi) { return (fam == F_SUM && norm_tpw.count(vi)) ? norm_tpw[vi] : 0; };
                    while (e2 < vs.size() && term_mode[vs[e2]] == term_mode[vs[b]] && m4(vs[e2]) == m4(vs[b]) && tpw_of(vs[e2]) == tpw_of(vs[b])) {
                        wide_ok = wide_ok && (g->vertices[vs[e2]].kind != K_NORMALIZE || bl == (size_t)kTileFrames);
                        // (a band-pass vertex' input sum -- mode 2: planar copy, 256-frame liveness -- only exists in the pair-mapped k_sum)
                        wide_ok = wide_ok && !(fam == F_SUM && band_plan.count(vs[e2]));
                        ++e2;
                    }
                    add_launch(fam, off + b * dsz, (int)(e2 - b),
                               term_mode[vs[b]] | (wide_ok ? 0x100u : 0u) | (m4(vs[b]) ? 0x200u : 0u) | ((uint32_t)tpw_of(vs[b]) << 12), lv);
                    b = e2;
                }
                continue;
            }
            add_launch(fam, off, (int)vs.size(), max_nseg, lv);
        }
        for (float2* t : level_tmp) g->free_bufs.push_back(t);
        // release buffers whose last consumer sits at this level
        for (size_t vi : g->order)
            if (g->vbuf[vi] && last_use[vi] == lv && (long)vi != g->output_vertex) {
                g->free_bufs.push_back(g->vbuf[vi]);
                // keep vbuf[vi] for descriptor bookkeeping of this level only
                last_use[vi] = -2;
            }
    }
    for (float2* b : env_bufs) g->free_bufs.push_back(b);
    // un-fused quantise when the output vertex is not a Normalize
    const Vertex& outv = g->vertices[(size_t)g->output_vertex];
    if (pcm_dst && qmode && outv.kind != K_NORMALIZE) {
        std::vector<QuantDesc> d{{g->vbuf[(size_t)g->output_vertex], pcm_dst, amplitude, (uint32_t)qmode}};
        add_launch(F_QUANT, st.put(d), 1, 0u, g->n_levels);
    }
    // the guard's verdict on this chunk: one workgroup adds up what the guarded scan launches estimated (k_band_audit)
    const double guard_thr = (double)g->band_guard_ppb * 1e-9;
    // (sine_mode 2) what k_sine_probe measured at the probed sine vertices' outputs joins the verdict: through the one launch that
    // gives it itself, where every probed vertex reaches the output through that launch's Normalize vertex (BASELINE config 3's
    // shape: no audit launch) -- else as audit sources of their own
    bool in_launch = guard_on && audit_src.size() == 1 && audit_src[0].fused && probe_src.size() <= 2;
    double probe_g2[2] = {0.0, 0.0};
    if (in_launch && !probe_src.empty()) {
        const size_t ni = audit_src[0].from;
        const double to_out = own_gain(g->vertices[ni]) * down[ni].g;   // from the Normalize vertex' raw frames to the graph's output
        for (size_t q = 0; q < probe_src.size(); ++q) {
            const size_t u = probe_src[q].first;
            const double r = down[u].g / to_out;
            if (down[u].norm != (long)ni || !(to_out > 0.0) || !std::isfinite(r)) in_launch = false;
            probe_g2[q] = r * r;
        }
    }
    if (!in_launch)
        for (const auto& ps : probe_src) audit_src.push_back({ps.second, probe_n, ps.first, false, 0, probe_stride, true});
    if (in_launch) {
        // ONE guarded launch and it ends in the Normalize vertex: the launch gives the verdict itself (BandScanDesc::nz_acc)
        const size_t o = audit_src[0].desc_off;
        // One probed vertex, a sample every 256 frames: sixteen samples per tile of the launch = one workgroup of k_sine_probe's --
        // the tiles evaluate their own (BandScanDesc::nz_probe) and the probe launch is taken off the list again.
        bool inl = g->inline_probe && probe_src.size() == 1 && probe_lg == 8u;
        if (inl) {
            size_t at = cb.launches.size();
            for (size_t l = 0; l < cb.launches.size(); ++l)
                if (cb.launches[l].fam == F_PROBE && cb.launches[l].off == probe_desc_off[0] && cb.launches[l].n == 1) at = l;
            if (at == cb.launches.size()) inl = false;
            else cb.launches.erase(cb.launches.begin() + (long)at);
        }
        for (size_t q = 0; q < probe_src.size(); ++q) {
            if (inl) ptr_field(o, offsetof(BandScanDesc, nz_probe), probe_desc_off[q]);
            else scratch_field(o, offsetof(BandScanDesc, nz_extra) + q * sizeof(const float*), probe_src[q].second);
            const float g2 = (float)probe_g2[q];
            memcpy(&st.b[o + offsetof(BandScanDesc, nz_xg2) + q * sizeof(float)], &g2, 4);
        }
        const uint32_t xcnt = (uint32_t)kTileFrames >> probe_lg;   // samples per wave-tile (4 .. 64)
        memcpy(&st.b[o + offsetof(BandScanDesc, nz_xcnt)], &xcnt, 4);
        const double gout = down[audit_src[0].from].g;
        const float scale = (float)(gout * gout / (double)M), thr2 = (float)(guard_thr * guard_thr);
        const uint64_t hw = (uint64_t)(uintptr_t)g->guard.d_word;
        memcpy(&st.b[o + offsetof(BandScanDesc, nz_scale)], &scale, 4);
        memcpy(&st.b[o + offsetof(BandScanDesc, nz_thr2)], &thr2, 4);
        memcpy(&st.b[o + offsetof(BandScanDesc, nz_host)], &hw, 8);
        const uint32_t n_tiles4 = (uint32_t)((M + 4 * (size_t)kTileFrames - 1) / (4 * (size_t)kTileFrames));
        cb.sync_fix.push_back({o + offsetof(BandScanDesc, nz_sync), cb.sync_bytes});   // one granule per tile, zeroed before the launch
        cb.sync_bytes += ((size_t)n_tiles4 * 8 + 63) & ~(size_t)63;
        g->guard.chunk_audited = true;
    } else if ((guard_on || sguard_on) && !audit_src.empty()) {
        g->guard.chunk_audited = true;
        std::vector<AuditDesc> ad;
        for (const AuditSrc& a : audit_src) {
            AuditDesc x{};
            x.n_wt = a.n_wt;
            x.tile_frames = a.tile_frames;
            x.nb = (uint32_t)nb;
            x.bl = (uint32_t)bl;
            x.gain = (float)down[a.from].g;
            x.sampled = a.sampled ? 1u : 0u;
            ad.push_back(x);
        }
        const size_t ao = st.put(ad);
        for (size_t i = 0; i < audit_src.size(); ++i) {
            const size_t o = ao + i * sizeof(AuditDesc);
            scratch_field(o, offsetof(AuditDesc, noise), audit_src[i].noise_off);
            const long nz = audit_src[i].fused ? -1 : down[audit_src[i].from].norm;   // (a fused Normalize vertex: 1 / max already applied by the launch)
            if (nz >= 0) {
                const auto it = audit_norm.find((size_t)nz);
                if (it == audit_norm.end()) return fail("termdaw_amd: internal: the guard lost a Normalize vertex");
                scratch_field(o, offsetof(AuditDesc, peaks), it->second.first);
                // (a scan pass measures against scan_max, graph.rs:222-237: what matters there is the recorded peak's relative error)
                scratch_field(o, offsetof(AuditDesc, init_copy), it->second.second + (is_scan ? sizeof(float) : 0));
            }
        }
        AuditHead hd{};
        hd.n = (uint32_t)audit_src.size();
        hd.frames = (uint32_t)M;
        hd.thr2 = (float)(guard_thr * guard_thr);
        hd.host_word = g->guard.d_word;
        const size_t ho = st.put(std::vector<AuditHead>{hd});
        ptr_field(ho, offsetof(AuditHead, descs), ao);
        add_launch(F_AUDIT, ho, 1, 0u, g->n_levels + 1);
    }
    const auto tp2 = std::chrono::steady_clock::now();
    g->host_ms[0] += ms_between(tp0, tp1);   // event compile
    g->host_ms[1] += ms_between(tp1, tp2);   // descriptors
    g->state_dev_dirty = true;
    return 1;
}

size_t desc_size(int fam) {
    switch (fam) {
        case F_LOOP: return sizeof(LoopDesc);
        case F_MULTI: return sizeof(MultiDesc);
        case F_LERP: return sizeof(LerpDesc);
        case F_SINE: return sizeof(SineDesc);
        case F_SYNTH: return sizeof(SynthDesc);
        case F_SAMPSYN: return sizeof(SampsynDesc);
        case F_ENV: return sizeof(AdsrVDesc);
        case F_PROBE: return sizeof(ProbeDesc);
        case F_SUM: return sizeof(SumDesc);
        case F_SCALE: return sizeof(ScaleDesc);
        case F_NORMFIX: return sizeof(SumDesc);
        case F_ADSR: return sizeof(AdsrVDesc);
        case F_BAND: return sizeof(BandDesc);
        case F_BAND_SPEC:
        case F_BAND_FIX:
        case F_BAND_FILL: return sizeof(BandSpecDesc);
        case F_BAND_SCAN: return sizeof(BandScanDesc);
        case F_QUANT: return sizeof(QuantDesc);
        case F_AUDIT: return sizeof(AuditHead);
        default: return 0;
    }
}
bool is_band_family(int fam) { return fam == F_BAND_SPEC || fam == F_BAND_FIX || fam == F_BAND_FILL; }

}  // namespace tde
