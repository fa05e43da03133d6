// engine.cpp -- graph model, chunk compiler and launch scheduler behind the C ABI.
//
// Reference items are cited as file:line relative to /root/reference/src.  This file holds the
// *host* half of the path: graph construction rules (graph.rs:49-174), the FlowwBank cursor
// (floww.rs:70-141) and the sequential per-vertex bookkeeping (voice lists, f32 envelope clocks,
// one-shot cursors) that the reference interleaves with its per-sample loops.  Everything that is
// per-sample runs on the GPU (kernels.hip).  There is no CPU render fallback: every render entry
// point fails when no HIP device is usable.
#include <chrono>

#include "engine.h"
#include "midi.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <atomic>
#include <thread>
#include <limits>

#include "wav.h"

using namespace tdk;

namespace tde {

thread_local std::string g_error;
int fail(const std::string& msg) {
    g_error = msg;
    return 0;
}

#define TD_HIP(expr)                                                                      \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            g_error = std::string("HIP error: ") + hipGetErrorString(_e) + " at " #expr;  \
            return 0;                                                                     \
        }                                                                                 \
    } while (0)

size_t f32_as_usize(float x) {   // Rust `x as usize`
    if (!(x == x)) return 0;
    if (x <= 0.0f) return 0;
    if (x >= 18446744073709551616.0f) return std::numeric_limits<size_t>::max();
    return (size_t)x;
}

static thread_local int t_device = 0;

static int ensure_device(int dev) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail("termdaw_amd: no HIP device available (the render path has no CPU fallback)");
    if (dev >= n) return fail("termdaw_amd: device index out of range");
    TD_HIP(hipSetDevice(dev));
    return 1;
}

// ------------------------------------------------------------------------------------------------
// sample load pipeline (sample.rs:38-77, 125-147, 252-313) -- decisions on the host, data on the device
// ------------------------------------------------------------------------------------------------
enum LoadMethod { LM_STEREO, LM_LEFT, LM_RIGHT, LM_LOUDEST, LM_NORM, LM_MIX };
static LoadMethod method_from(const char* s) {   // sample.rs:199-210
    std::string m = s ? s : "";
    if (m == "left") return LM_LEFT;
    if (m == "right") return LM_RIGHT;
    if (m == "loudest") return LM_LOUDEST;
    if (m == "normalize-seperate") return LM_NORM;
    if (m == "mix-down") return LM_MIX;
    return LM_STEREO;
}

// Build-defined sinc resampler (DESIGN.md "Resampler"): the reference calls rubato::SincFixedIn<f32> with
// sinc_len 256, f_cutoff 0.95, Linear, oversampling 256, BlackmanHarris2 (sample.rs:152-158,
// state.rs:534-540); rubato's arithmetic is un-vendored, so this is this engine's own specification with
// that parameter set -- parity with the reference is unpinned, parity with oracle/ is bit-exact.
static void build_sinc_table(size_t from, size_t to, std::vector<float>* T) {
    const double ratio = (double)to / (double)from;
    const double fc = 0.95 * (ratio < 1.0 ? ratio : 1.0);
    const double pi = 3.14159265358979323846;
    auto tap = [&](double d) {   // window^2 x sinc at d input frames from the filter centre (|d| <= 128)
        const double z = fc * d;
        const double sinc = z == 0.0 ? 1.0 : sin(pi * z) / (pi * z);
        const double u = (d + 128.0) / 256.0;
        const double bh = 0.35875 - 0.48829 * cos(2.0 * pi * u) + 0.14128 * cos(4.0 * pi * u) - 0.01168 * cos(6.0 * pi * u);
        return sinc * bh * bh;
    };
    // normalisation as rubato's make_sincs does it (from the crate's published source, from memory -- unverified): the sum
    // of ALL sinc_len * oversampling points of the windowed sinc, divided by the oversampling factor
    double sum = 0.0;
    for (int x = 0; x < kSincLen * kSincOver; ++x) sum += tap((double)(x - kSincLen * kSincOver / 2) / (double)kSincOver);
    const double norm = sum / (double)kSincOver;
    T->resize((size_t)(kSincOver + 1) * kSincLen);
    for (int p = 0; p <= kSincOver; ++p)
        for (int k = 0; k < kSincLen; ++k)
            (*T)[(size_t)p * kSincLen + k] = (float)(tap((double)k - 127.0 - (double)p / (double)kSincOver) / norm);
}
// Resamples `len` frames at `in` (device) from rate `from` to rate `to` into a fresh device buffer.
int resample_device(const float2* in, size_t len, size_t from, size_t to, float2** out, size_t* nout_p, hipStream_t st) {
    if (from == 0 || to == 0) return fail("resample: sample rate 0");
    const size_t nout = (size_t)(((unsigned __int128)len * to + from - 1) / from);
    std::vector<float> T;
    build_sinc_table(from, to, &T);
    float* d_T = nullptr;
    float2* d_out = nullptr;
    TD_HIP(hipMalloc(&d_T, T.size() * sizeof(float)));
    if (hipMalloc(&d_out, (nout + (nout & 1) + 17) * sizeof(float2)) != hipSuccess) {
        (void)hipFree(d_T);
        return fail("termdaw_amd: out of device memory for the resampled sample");
    }
    auto bail = [&](hipError_t e, const char* what) {
        (void)hipFree(d_T);
        (void)hipFree(d_out);
        return fail(std::string("HIP error: ") + hipGetErrorString(e) + " at " + what);
    };
    hipError_t he;
    if ((he = hipMemcpyAsync(d_T, T.data(), T.size() * sizeof(float), hipMemcpyHostToDevice, st)) != hipSuccess) return bail(he, "table upload");
    // the pad behind the last frame (odd tail + wrap frames) starts out as zeros
    if ((he = hipMemsetAsync(d_out + nout, 0, ((nout & 1) + 17) * sizeof(float2), st)) != hipSuccess) return bail(he, "pad memset");
    if (nout == 0) {
        (void)hipStreamSynchronize(st);
        (void)hipFree(d_T);
        *out = d_out;
        *nout_p = 0;
        return 1;
    }
    ResampleDesc d{in, d_out, d_T, len, nout, from, to};
    launch_resample(d, st);
    // frames nout .. nout + 14 = the first frames again (wrap frames, as in every bank entry)
    for (size_t i = 0; nout && i < 15;) {
        const size_t src = i % nout, cnt = std::min<size_t>(15 - i, nout - src);
        if ((he = hipMemcpyAsync(d_out + nout + i, d_out + src, cnt * sizeof(float2), hipMemcpyDeviceToDevice, st)) != hipSuccess)
            return bail(he, "wrap frames");
        i += cnt;
    }
    if ((he = hipStreamSynchronize(st)) != hipSuccess) return bail(he, "resample");
    if ((he = hipGetLastError()) != hipSuccess) return bail(he, "resample launch");
    (void)hipFree(d_T);
    *out = d_out;
    *nout_p = nout;
    return 1;
}

}  // namespace tde

void* td_samplebank::alloc(int pool, size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    auto& v = slabs[pool];
    if (v.empty() || v.back().cap - v.back().used < bytes) {
        Slab s;
        s.cap = std::max<size_t>(bytes, (size_t)16 << 20);
        if (hipMalloc(&s.base, s.cap) != hipSuccess) {
            tde::fail("termdaw_amd: out of device memory for the sample bank");
            return nullptr;
        }
        v.push_back(s);
    }
    Slab& s = v.back();
    void* p = s.base + s.used;
    s.used += bytes;
    s.live += 1;
    return p;
}
namespace tde {
// Arenas with a deferred k_norm_fix outstanding (Arena::pending_fix).  The fix, if it runs, gathers again from the sample
// tables: a bank about to give memory back settles every such arena on its device first.
static std::mutex g_fix_mu;
static std::vector<Arena*> g_fix_arenas;
static int settle_arena(Arena& ar, hipStream_t stream);
static int cur_device() {
    int d = 0;
    (void)hipGetDevice(&d);
    return d;
}
static void note_pending(Arena& ar, hipStream_t stream, int device) {
    ar.fix_stream = stream;
    ar.fix_device = device;
    if (ar.listed) return;
    std::lock_guard<std::mutex> lk(g_fix_mu);
    g_fix_arenas.push_back(&ar);
    ar.listed = true;
}
static void unlist_arena(Arena& ar) {
    if (!ar.listed) return;
    std::lock_guard<std::mutex> lk(g_fix_mu);
    g_fix_arenas.erase(std::remove(g_fix_arenas.begin(), g_fix_arenas.end(), &ar), g_fix_arenas.end());
    ar.listed = false;
}
static void drop_pending(Arena& ar) {
    ar.pending_fix.clear();
    unlist_arena(ar);
}
static void settle_device_arenas(int device) {
    std::vector<Arena*> todo;
    {
        std::lock_guard<std::mutex> lk(g_fix_mu);
        for (Arena* a : g_fix_arenas)
            if (a->fix_device == device && !a->pending_fix.empty()) todo.push_back(a);
    }
    for (Arena* a : todo) (void)settle_arena(*a, a->fix_stream);
}
}  // namespace tde

void td_samplebank::release(void* p) {
    if (!p) return;
    for (auto& v : slabs)
        for (size_t i = 0; i < v.size(); ++i) {
            Slab& s = v[i];
            if ((unsigned char*)p >= s.base && (unsigned char*)p < s.base + s.cap) {
                if (s.live) --s.live;
                if (s.live == 0) {               // nothing left in it: reuse from the start, or give it back
                    // (kernels queued by td_graph_render_all_async / td_batch_render_all_async on the engines' non-blocking
                    // streams may still gather from this memory: the per-sample hipFree of the old path synchronised implicitly)
                    (void)hipSetDevice(device);   // (the BANK's device, whatever the calling thread last selected)
                    tde::settle_device_arenas(device);   // (... and a deferred k_norm_fix would gather from it once more)
                    (void)hipDeviceSynchronize();
                    if (i + 1 == v.size()) s.used = 0;
                    else { (void)hipFree(s.base); v.erase(v.begin() + (long)i); }
                }
                return;
            }
        }
    (void)hipFree(p);   // a stand-alone allocation (resampled entries)
}
void td_samplebank::release_all() {
    tde::settle_device_arenas(device);
    (void)hipDeviceSynchronize();
    for (auto& v : slabs) {
        for (auto& s : v) (void)hipFree(s.base);
        v.clear();
    }
}

namespace tde {

// SampleBank::add after the WAV header is known (sample.rs:240-313).  The host only takes the decisions
// that depend on counts (channel / length checks, Sample::from's Err arms); decode, de-interleave, load
// mode, peak scan and normalisation run on the device.  Exactly one of `linear` (already decoded f32 stream)
// and `raw` (little-endian PCM words) is given.
static int bank_add_stream(td_samplebank* sb, const std::string& name, const float* linear, const uint8_t* raw,
                           uint32_t raw_format, size_t n_values, int channels, size_t sr, size_t bd, LoadMethod method) {
    if (sb->names.count(name))
        return fail("TermDaw: SampleBank: there is already a sample with name \"" + name + "\" present.");
    if (method == LM_STEREO && channels != 2)
        return fail("TermDaw: SampleBank: only 2 channel samples are supported for stereo samples.");
    if (method != LM_STEREO && channels > 2)
        return fail("TermDaw: SampleBank: only 1,2 channel samples are supported for left or right samples.");
    if (channels < 1) return fail("TermDaw: SampleBank: sample has no channels.");
    if (n_values > 0xFFFFFFF0ull) return fail("termdaw_amd: sample too long");
    sb->max_sr = std::max(sb->max_sr, sr);
    sb->max_bd = std::max(sb->max_bd, bd);
    // lengths of the de-interleaved l / r (sample.rs:275-292)
    size_t nl, nr;
    if (channels == 1) {
        nl = method == LM_LEFT ? n_values : 0;
        nr = method == LM_LEFT ? 0 : n_values;
    } else {
        nl = n_values / 2 + (n_values & 1);   // a dangling last value goes to l
        nr = n_values / 2;
    }
    // Sample::from (sample.rs:38-77): which decoded channel feeds l and r; src 0/1 = channel index
    uint32_t src_l = 0, src_r = channels == 2 ? 1u : 0u;
    bool pick_loudest = false;
    switch (method) {
        case LM_LEFT:
            if (nl == 0) return fail("TermDaw: Sample::from: l has length 0.");
            nr = nl;
            src_r = src_l = 0;
            break;
        case LM_RIGHT:
            if (nr == 0) return fail("TermDaw: Sample::from: r has length 0.");
            nl = nr;
            src_l = src_r = channels == 2 ? 1u : 0u;
            break;
        case LM_LOUDEST: pick_loudest = true; break;
        default:
            if (nl != nr) return fail("TermDaw: Sample::from: l and r do not have the same length.");
            if (nl == 0) return fail("TermDaw: Sample::from: l and r have length 0.");
    }
    if (!ensure_device(sb->device)) return 0;
    hipStream_t st = nullptr;   // load time: the default stream is fine
    const size_t nmax = std::max(nl, nr);
    float *d_lin = nullptr, *d_l = nullptr, *d_r = nullptr, *d_s = nullptr;
    uint8_t* d_raw = nullptr;
    auto cleanup = [&]() {};   // (the scratch belongs to the bank)
    {
        auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
        const size_t bps_raw = linear ? 0 : (raw_format == PCM_U8 ? 1 : raw_format == PCM_S16 ? 2 : raw_format == PCM_S24 ? 3 : 4);
        const size_t b_lin = up(std::max<size_t>(n_values, 1) * sizeof(float)), b_ch = up(std::max<size_t>(nmax, 1) * sizeof(float));
        const size_t b_raw = up(std::max<size_t>(n_values * bps_raw, 1));
        const size_t need = b_lin + 2 * b_ch + 256 + b_raw;
        if (need > sb->tmp_cap) {
            TD_HIP(hipDeviceSynchronize());
            if (sb->tmp) (void)hipFree(sb->tmp);
            sb->tmp = nullptr;
            sb->tmp_cap = 0;
            TD_HIP(hipMalloc(&sb->tmp, need + need / 2));
            sb->tmp_cap = need + need / 2;
        }
        unsigned char* p = sb->tmp;
        d_lin = reinterpret_cast<float*>(p); p += b_lin;
        d_l = reinterpret_cast<float*>(p); p += b_ch;
        d_r = reinterpret_cast<float*>(p); p += b_ch;
        d_s = reinterpret_cast<float*>(p); p += 256;
        d_raw = p;
    }
    if (linear) {
        TD_HIP(hipMemcpyAsync(d_lin, linear, n_values * sizeof(float), hipMemcpyHostToDevice, st));
    } else {
        const size_t bps = raw_format == PCM_U8 ? 1 : raw_format == PCM_S16 ? 2 : raw_format == PCM_S24 ? 3 : 4;
        TD_HIP(hipMemcpyAsync(d_raw, raw, n_values * bps, hipMemcpyHostToDevice, st));
        launch_pcm_decode(d_raw, d_lin, (uint32_t)n_values, raw_format, st);
    }
    if (pick_loudest) {
        // Loudest (sample.rs:54-62): mean |l| > mean |r| ? l : r, with the means summed left to right in f32
        if (channels == 1) {   // one of the two is empty: mean_energy(empty) = 0
            launch_sample_split(d_lin, 1, 0, 0, d_l, d_r, (uint32_t)nl, (uint32_t)nr, st);
        } else {
            launch_sample_split(d_lin, 2, 0, 1, d_l, d_r, (uint32_t)nl, (uint32_t)nr, st);
        }
        launch_abs_sum_serial(d_l, (uint32_t)nl, d_s + 4, st);
        launch_abs_sum_serial(d_r, (uint32_t)nr, d_s + 5, st);
        float sums[2];
        TD_HIP(hipMemcpyAsync(sums, d_s + 4, 8, hipMemcpyDeviceToHost, st));
        TD_HIP(hipStreamSynchronize(st));
        const float lm = nl ? sums[0] / (float)nl : 0.0f, rm = nr ? sums[1] / (float)nr : 0.0f;
        const bool take_l = lm > rm;
        const size_t n_take = take_l ? nl : nr;
        if (n_take == 0) { cleanup(); return fail("termdaw_amd: loudest channel is empty"); }
        // both outputs become the chosen channel
        TD_HIP(hipMemcpyAsync(take_l ? d_r : d_l, take_l ? d_l : d_r, n_take * sizeof(float), hipMemcpyDeviceToDevice, st));
        nl = nr = n_take;
    } else {
        launch_sample_split(d_lin, (uint32_t)channels, src_l, src_r, d_l, d_r, (uint32_t)nl, (uint32_t)nr, st);
    }
    size_t n = nl;
    const float *p_max_l = d_s, *p_max_r = d_s + 1;
    if (method == LM_NORM) {   // normalize_seperate (sample.rs:132-137)
        launch_absmax(d_l, (uint32_t)nl, d_s, st);
        launch_absmax(d_r, (uint32_t)nr, d_s + 1, st);
    } else if (method == LM_MIX) {   // mix_down (sample.rs:139-147): zip stops at the shorter channel
        n = std::min(nl, nr);
        launch_add_planar(d_l, d_r, d_l, (uint32_t)n, st);
        launch_absmax(d_l, (uint32_t)n, d_s, st);
        TD_HIP(hipMemcpyAsync(d_r, d_l, n * sizeof(float), hipMemcpyDeviceToDevice, st));
        p_max_r = d_s;
    } else {   // normalize(usize::MAX) (sample.rs:125-130): one common peak
        launch_absmax(d_l, (uint32_t)nl, d_s, st);
        launch_absmax(d_r, (uint32_t)nr, d_s + 1, st);
        // max(absmax(l), absmax(r)): both are >= 0, so the uint pattern orders like the float
        launch_absmax(d_s, 2, d_s + 2, st);
        p_max_l = p_max_r = d_s + 2;
    }
    if (nl != nr && method != LM_MIX) { cleanup(); return fail("termdaw_amd: channel lengths differ after load"); }
    if (n == 0) { cleanup(); return fail("TermDaw: Sample::from: l and r have length 0."); }
    SampleEntry e;
    e.len = n;
    // n frames, then the first 15 again (wrap frames for looping readers), padded to an even count
    // (every failure return below hands the entry's memory back to the bank)
    e.d = static_cast<float2*>(sb->alloc(0, (n + 16 + (n & 1)) * sizeof(float2)));
    if (!e.d) return 0;
    struct Guard {
        td_samplebank* sb; SampleEntry* e; bool armed = true;
        ~Guard() { if (armed) { sb->release(e->d); sb->release(e->d16); } }
    } guard{sb, &e};
    TD_HIP(hipMemsetAsync(e.d + n, 0, (16 + (n & 1)) * sizeof(float2), st));
    launch_sample_pack(d_l, d_r, p_max_l, p_max_r, e.d, (uint32_t)n, st);
    // packed 16-bit twin: only when l / r are still the raw integer PCM values times one scale per channel
    // (every mode but mix-down) and no resample follows
    bool want16 = method != LM_MIX && sr == sb->sample_rate && n >= 1 && n < 0x3FFFFFF0u;
    if (want16) {
        e.d16 = static_cast<uint32_t*>(sb->alloc(1, ((n + 18) & ~(size_t)3) * sizeof(uint32_t)));   // the loop + its first 15 frames again
        if (!e.d16) return 0;
        TD_HIP(hipMemsetAsync(d_s + 8, 0, sizeof(uint32_t), st));
        launch_sample_pack16(d_l, d_r, e.d16, (uint32_t)n, reinterpret_cast<uint32_t*>(d_s + 8), st);
    }
    float host_s[12] = {0};
    TD_HIP(hipMemcpyAsync(host_s, d_s, sizeof host_s, hipMemcpyDeviceToHost, st));
    TD_HIP(hipStreamSynchronize(st));
    TD_HIP(hipGetLastError());
    if (want16) {
        uint32_t bad;
        memcpy(&bad, &host_s[8], 4);
        const float ml = host_s[p_max_l - d_s], mr = host_s[p_max_r - d_s];
        if (bad) {
            sb->release(e.d16);
            e.d16 = nullptr;
        } else {
            e.scale_l = 1.0f / ml;   // the same `1.0 / max` k_sample_pack multiplied by
            e.scale_r = 1.0f / mr;
        }
    }
    cleanup();
    if (sr != sb->sample_rate) {   // sample.rs:305-310: Sample::resample after the normalisation
        float2* rs = nullptr;
        size_t nout = 0;
        if (!resample_device(e.d, e.len, sr, sb->sample_rate, &rs, &nout, st)) return 0;
        sb->release(e.d);
        e.d = rs;   // (a stand-alone allocation: release() tells it from slab memory)
        if (nout == 0) return fail("termdaw_amd: resampled sample is empty");
        e.len = nout;
    }
    guard.armed = false;
    sb->samples.push_back(e);
    sb->names[name] = sb->samples.size() - 1;
    return 1;
}

}  // namespace tde

using namespace tde;

// ------------------------------------------------------------------------------------------------
// FlowwBank cursor (floww.rs:70-91)
// ------------------------------------------------------------------------------------------------
uint64_t td_flowwbank::next_version() {
    static std::atomic<uint64_t> counter{1};
    return counter.fetch_add(1);
}
void td_flowwbank::set_start_indices_to_frame(size_t t_frame, bool do_skip) {
    for (size_t i = 0; i < flowws.size(); ++i) {
        const auto& fl = flowws[i];
        for (size_t j = do_skip ? start_indices[i] : 0; j < fl.size(); ++j) {
            if (frame_of(fl[j]) >= t_frame) {
                start_indices[i] = j;
                break;
            }
        }
    }
}
void td_flowwbank::set_time(size_t t) {
    set_start_indices_to_frame(t, false);
    frame = t;
}
void td_flowwbank::set_time_to_next_block() {
    frame += bl;
    set_start_indices_to_frame(frame, true);
}

namespace tde {

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
    {   // tiles with an interval start strictly inside them first (IntervalTab::tile_order)
        const size_t nt = tile_first.size() - 1;
        std::vector<uint32_t> order, light;
        order.reserve(nt);
        for (size_t t = 0; t < nt; ++t) (tile_first[t + 1] > tile_first[t] + (ib.istart[tile_first[t + 1]] == (t + 1) * kTileFrames ? 1u : 0u)
                                             ? order : light).push_back((uint32_t)t);
        order.insert(order.end(), light.begin(), light.end());
        vt.tile_order_off = st.put(order);
    }
    vt.istart_off = st.put(ib.istart);
    vt.ivoff_off = st.put(ib.ivoff);
    vt.voices_off = st.put(ib.voices);
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
                rec.push_back(make_float4(hz, env_t, 0.0f, 0.0f));
                for (int o = 0; o < 3; ++o) {
                    if (conf_of[o] < 0) { rec.push_back(make_float4(0.f, 0.f, 0.f, 0.f)); continue; }
                    const AdsrConfD& c = osc[o]->adsr;
                    const double K = (double)vel * (double)osc[o]->volume * amp * shape[o];
                    rec.push_back(synth_osc_piece(c, synth_piece(c, env_t, rel_t, ia, srf), rel_t, K));
                }
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
    if (v.kind == K_SYNTH && synth_affine_ok(v)) {
        ib.finish();
        std::vector<uint32_t> istart, ivoff;
        std::vector<float4> rec;
        synth_refine_affine(v, ib, bl, sr, istart, ivoff, rec);
        put_intervals(IntervalView{istart, ivoff, rec, ib.limit}, st, vt);
        return 1;
    }
    put_intervals(ib, st, vt);
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
static void save_state(const Vertex& v, std::string& out) {
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
static void load_state(Vertex& v, const std::string& in) {
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
        case K_SYNTH: put_pod(key, v.square); put_pod(key, v.topflat); put_pod(key, v.triangle); break;   // (retain rule: release times)
        case K_SAMPSYN: put_pod(key, v.conf); break;
        case K_ADSR: put_pod(key, (uint8_t)v.use_off); put_pod(key, v.conf); break;
        default: break;
    }
    save_state(v, key);
}
static int upload_tables(td_graph* g, TableCache& tc, const Staging& tmp) {
    const size_t bytes = std::max<size_t>(tmp.b.size(), 16);
    if (bytes > tc.cap) {
        TD_HIP(hipStreamSynchronize(g->stream));   // (kernels of earlier renders may still read the old buffer)
        if (tc.d) { (void)hipFree(tc.d); g->device_bytes -= tc.cap; }
        if (tc.h) (void)hipHostFree(tc.h);
        tc.d = tc.h = nullptr;
        tc.cap = 0;
        tc.inflight = false;
        const size_t cap = bytes + bytes / 4 + 256;
        TD_HIP(hipMalloc(&tc.d, cap));
        TD_HIP(hipHostMalloc(&tc.h, cap, hipHostMallocDefault));
        if (!tc.copied) TD_HIP(hipEventCreateWithFlags(&tc.copied, hipEventDisableTiming));
        tc.cap = cap;
        g->device_bytes += cap;
    }
    if (tc.inflight) TD_HIP(hipEventSynchronize(tc.copied));
    tc.inflight = false;
    memcpy(tc.h, tmp.b.data(), tmp.b.size());
    TD_HIP(hipMemcpyAsync(tc.d, tc.h, tmp.b.size(), hipMemcpyHostToDevice, g->stream));
    TD_HIP(hipEventRecord(tc.copied, g->stream));
    tc.inflight = true;
    return 1;
}
static void free_tables(td_graph* g) {   // (device selected, stream synchronised by the caller)
    for (auto& v : g->vertices)
        if (v.tables) {
            if (v.tables->d) { (void)hipFree(v.tables->d); g->device_bytes -= v.tables->cap; }
            if (v.tables->h) (void)hipHostFree(v.tables->h);
            if (v.tables->copied) (void)hipEventDestroy(v.tables->copied);
            v.tables.reset();
        }
}

// ------------------------------------------------------------------------------------------------
// plan: reachable set, topological levels (graph.rs:98-108 visits exactly the vertices that reach the
// output; others never run and never advance -- quirk Q12)
// ------------------------------------------------------------------------------------------------
static void build_plan(td_graph* g) {
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

static int ensure_graph_device(td_graph* g) {
    if (!ensure_device(g->device)) return 0;
    if (!g->stream) {   // (also after a td_batch_free whose stream creation failed: made on next use)
        TD_HIP(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
        g->owns_stream = true;
    }
    if (!g->ev_fork) {
        TD_HIP(hipEventCreateWithFlags(&g->ev_fork, hipEventDisableTiming));
        TD_HIP(hipMalloc(&g->d_scalar, 256));
    }
    if (!g->guard.h_word) {   // (band_mode 2: k_band_audit's verdict lands here)
        TD_HIP(hipHostMalloc((void**)&g->guard.h_word, 64, hipHostMallocMapped | hipHostMallocCoherent));
        g->guard.h_word[0] = 0u;
        g->guard.h_word[1] = 0u;
        TD_HIP(hipHostGetDevicePointer((void**)&g->guard.d_word, g->guard.h_word, 0));
    }
    if (g->branch_streams && !g->aux[0]) {   // branch streams are made on first use (a batch of 64 graphs never needs them)
        for (int a = 0; a < td_graph::kAuxStreams; ++a) {
            TD_HIP(hipStreamCreateWithFlags(&g->aux[a], hipStreamNonBlocking));
            TD_HIP(hipEventCreateWithFlags(&g->ev_join[a], hipEventDisableTiming));
        }
    }
    return 1;
}

static int ensure_state_slots(td_graph* g) {
    const size_t need = g->hstate.size();
    if (need > g->dstate_cap) {
        const size_t cap = std::max<size_t>(64, need * 2);
        StateSlot* nd = nullptr;
        TD_HIP(hipMalloc(&nd, cap * sizeof(StateSlot)));
        if (g->dstate) {
            TD_HIP(hipStreamSynchronize(g->stream));
            TD_HIP(hipMemcpy(nd, g->dstate, g->dstate_cap * sizeof(StateSlot), hipMemcpyDeviceToDevice));
            TD_HIP(hipFree(g->dstate));
        }
        g->dstate = nd;
        g->dstate_cap = cap;
    }
    if (g->state_host_dirty && need) {
        // host mirror is authoritative only right after construction / explicit host edits
        TD_HIP(hipStreamSynchronize(g->stream));
        TD_HIP(hipMemcpy(g->dstate, g->hstate.data(), need * sizeof(StateSlot), hipMemcpyHostToDevice));
        g->state_host_dirty = false;
    }
    return 1;
}

static int drain(td_graph* g);   // stream drained + deferred k_norm_fix settled (defined with the render loop)
static int pull_state(td_graph* g) {
    if (g->state_dev_dirty && g->dstate && !g->hstate.empty()) {
        if (!drain(g)) return 0;
        TD_HIP(hipMemcpy(g->hstate.data(), g->dstate, g->hstate.size() * sizeof(StateSlot), hipMemcpyDeviceToHost));
        g->state_dev_dirty = false;
        for (const auto& v : g->vertices)   // (a set_time not yet carried to the device by a submission)
            if (v.kind == K_BAND_PASS && v.state_slot >= 0 && v.first_pending) g->hstate[(size_t)v.state_slot].band.first = 1u;
    }
    return 1;
}

static int ensure_buffers(td_graph* g, size_t frames) {
    frames += frames & 1;
    if (frames > g->cap_frames) {
        if (!drain(g)) return 0;
        for (float2* p : g->pool) {
            (void)hipFree(p);
            g->device_bytes -= g->cap_frames * sizeof(float2);
        }
        g->pool.clear();
        g->cap_frames = frames;
    }
    g->free_bufs = g->pool;
    return 1;
}
static float2* take_buffer(td_graph* g) {
    if (!g->free_bufs.empty()) {
        float2* p = g->free_bufs.back();
        g->free_bufs.pop_back();
        return p;
    }
    float2* p = nullptr;
    if (hipMalloc(&p, g->cap_frames * sizeof(float2)) != hipSuccess) return nullptr;
    g->pool.push_back(p);
    g->device_bytes += g->cap_frames * sizeof(float2);
    return p;
}

// Waits for `stream`, then looks at the arena's host-visible word: a single-pass Normalize tile of the last submission gave
// up its bounded wait for an earlier tile (SumDesc modes 4 / 5) and the check launch was not enqueued behind it -> k_norm_fix
// redoes the vertex now, from the block peaks the launch left behind.  The normal case costs one load of page-locked memory.
static int settle_arena(Arena& ar, hipStream_t stream) {
    if (!stream) return 1;
    TD_HIP(hipStreamSynchronize(stream));
    if (!ar.h_flag || !*(volatile uint32_t*)ar.h_flag) {
        drop_pending(ar);   // (settled: the launch needed no fix)
        return 1;
    }
    for (const auto& f : ar.pending_fix) launch_norm_fix((const SumDesc*)(ar.d + f.off), f.n, f.M, f.bl, stream);
    TD_HIP(hipGetLastError());
    TD_HIP(hipStreamSynchronize(stream));
    *(volatile uint32_t*)ar.h_flag = 0u;
    drop_pending(ar);
    ar.fix_runs += 1;
    return 1;
}

static int ensure_arena(Arena& ar, size_t bytes, hipStream_t stream) {
    if (!ar.h_flag) {
        TD_HIP(hipHostMalloc((void**)&ar.h_flag, 64, hipHostMallocMapped | hipHostMallocCoherent));
        *ar.h_flag = 0u;
        TD_HIP(hipHostGetDevicePointer((void**)&ar.d_flag, ar.h_flag, 0));
    }
    if (bytes <= ar.cap) return 1;
    const size_t cap = std::max<size_t>(bytes * 2, 1 << 20);
    if (stream && !settle_arena(ar, stream)) return 0;   // (a pending fix reads descriptors in the buffer about to go)
    drop_pending(ar);
    if (ar.h) (void)hipHostFree(ar.h);
    if (ar.d) (void)hipFree(ar.d);
    ar.h = nullptr;
    ar.d = nullptr;
    ar.cap = 0;
    ar.device_bytes = 0;
    TD_HIP(hipHostMalloc(&ar.h, cap, hipHostMallocDefault));
    TD_HIP(hipMalloc(&ar.d, cap));
    if (!ar.copied) TD_HIP(hipEventCreateWithFlags(&ar.copied, hipEventDisableTiming));
    ar.cap = cap;
    ar.device_bytes = cap;
    ar.inflight = false;
    ar.valid = 0;
    ar.graph_key.clear();
    ar.esync_len = 0;   // (new memory: the epoch-tagged words are zeroed by the next submission that has any)
    return 1;
}
static void free_arena(Arena& ar) {
    drop_pending(ar);
    if (ar.h) (void)hipHostFree(ar.h);
    if (ar.d) (void)hipFree(ar.d);
    if (ar.copied) (void)hipEventDestroy(ar.copied);
    if (ar.graph_exec) (void)hipGraphExecDestroy(ar.graph_exec);
    if (ar.h_flag) (void)hipHostFree(ar.h_flag);
    ar = Arena{};
}

enum Family { F_LOOP, F_MULTI, F_LERP, F_SINE, F_SYNTH, F_SAMPSYN, F_ENV, F_SUM, F_SCALE, F_NORMFIX, F_ADSR, F_BAND, F_BAND_SPEC, F_BAND_FIX, F_BAND_FILL, F_BAND_SCAN, F_QUANT, F_AUDIT,
              F_SOURCES /* (no descriptors of its own: several of the families above as ONE grid, submit_chunk) */, F_COUNT };
static const char* kFamilyName[F_COUNT] = {"k_sample_loop", "k_sample_multi", "k_sample_lerp", "k_debug_sine",
                                           "k_synth",       "k_sampsyn", "k_adsr_env", "k_sum",          "k_scale",       "k_norm_fix",
                                           "k_adsr",        "k_band_pass",    "k_band_spec", "k_band_fix", "k_band_fill", "k_band_scan", "k_quantise", "k_band_audit", "k_sources"};

static hipEvent_t get_event(ProfCtx& pc) {
    if (!pc.free_ev.empty()) {
        hipEvent_t e = pc.free_ev.back();
        pc.free_ev.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

struct Prof {
    ProfCtx& pc;
    int fam;
    hipStream_t s;
    hipEvent_t a = nullptr, b = nullptr;
    Prof(ProfCtx& pc_, int fam_, hipStream_t s_) : pc(pc_), fam(fam_), s(s_) {
        if (pc.now) {
            a = get_event(pc);
            b = get_event(pc);
            (void)hipEventRecord(a, s);
        }
    }
    ~Prof() {
        if (pc.now) {
            (void)hipEventRecord(b, s);
            pc.pending.push_back({a, b, fam});
        }
    }
};

// Branch streams: vertices of one topological level are mutually independent.  Same-kind vertices already
// share one batched launch; launches of DIFFERENT kinds in a level go to separate HIP streams (fork after the
// previous level, join before the next) so independent branches of the graph overlap on the device.  The
// sum -> scale -> band-pass chain of a level stays on the main stream.
static int aux_stream_of(int fam) {
    switch (fam) {
        case F_LOOP: return 0;
        case F_MULTI: return 1;
        case F_LERP: return 2;
        case F_SINE: return 3;
        case F_SYNTH: return 4;
        case F_SAMPSYN: return 4;
        case F_ADSR: return 5;
        case F_BAND: return 6;
        default: return -1;
    }
}

// Band-pass execution plan (DESIGN.md "exact parallel band-pass"): segment length S, warm-up W.  The
// warm-up must outlast the contraction (1 - gamma)^W of the slower chain.  The choice affects speed only --
// k_band_fix verifies every segment bit for bit and repairs what failed.
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
    if (!g->band_parallel) return p;
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
    p.S = 256;
    while ((M + p.S - 1) / p.S > kBandMaxSegs && p.S < kBandMaxS) p.S *= 2;
    p.nseg = (uint32_t)((M + p.S - 1) / p.S);
    if (p.nseg > kBandMaxSegs) return p;
    p.parallel = p.nseg >= 8;        // tiny chunks (block pulls) stay on the serial kernel
    // (the guess pays where the short warm-up is long -- cut-offs below ~75 Hz; elsewhere the walk is a few
    // hundred steps anyway and the block responses would only cost their reduction in the input-sum kernel)
    if (p.parallel && g->band_quick && p.S == 256 && p.Ws >= g->band_guess_min) {
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
        kmax = std::max(kmax, ceil((double)g->band_scan_depth / per_tile));
    }
    if (!(kmax <= (double)kScanMaxK)) return false;
    sp->K = (uint32_t)kmax;
    sp->Kw = sp->nf == 16 ? sp->K : 0u;   // (k_band_chain is built for 16 frames per lane)
    return true;
}

// ------------------------------------------------------------------------------------------------
// one chunk: compile tables + descriptors (compile_chunk), then upload and launch level by level (submit_chunk)
// ------------------------------------------------------------------------------------------------
static double ms_between(std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::milli>(b - a).count();
}

// Steps 1 and 2 for ONE graph, appended to `cb` (which several graphs of a batch may share).
static int compile_chunk(td_graph* g, const td_samplebank* sb, const td_flowwbank* fb,
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
    std::vector<std::vector<size_t>> cons(nv);
    for (size_t vi : g->order)
        for (size_t u : g->edges[vi]) cons[u].push_back(vi);
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
    if (guard_on) {
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
    auto guard_ok = [&](size_t vi, const ScanPlan& sp) {
        return g->vertices[vi].pass && sp.Kw != 0u && down[vi].norm != -2 && !short_up[vi];
    };
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
    struct AuditSrc { size_t noise_off; uint32_t n_wt; size_t from; bool fused; size_t desc_off; };   // from: the vertex whose output the estimate stands at
    std::vector<AuditSrc> audit_src;

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
                case K_DEBUG_SINE: fam_v[F_SINE].push_back(vi); break;
                case K_SYNTH: fam_v[F_SYNTH].push_back(vi); break;
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
                            bp.tmp = take_buffer(g);
                            bp.tmpq = take_buffer(g);
                            if (!bp.tmp || !bp.tmpq) return fail("termdaw_amd: out of device memory for edge buffers");
                            level_tmp.push_back(bp.tmp);
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
                    for (size_t vi : vs) {
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
                        {
                            auto same = [](const AdsrConfD& a, const AdsrConfD& b) { return memcmp(&a, &b, sizeof(AdsrConfD)) == 0; };
                            const bool sq = v.square.volume > 0.0f, tf = v.topflat.volume > 0.0f;
                            x.tf_env_src = (sq && same(v.topflat.adsr, v.square.adsr)) ? 1u : 0u;
                            x.tr_env_src = (sq && same(v.triangle.adsr, v.square.adsr)) ? 1u
                                         : (tf && same(v.triangle.adsr, v.topflat.adsr)) ? 2u : 0u;
                        }
                        d.push_back(x);
                    }
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
                        if (guard_on) {
                            const uint32_t n_wt = (uint32_t)((M + (size_t)kTileFrames - 1) / (size_t)kTileFrames);
                            audit_src.push_back({scratch((size_t)n_wt * sizeof(float)), n_wt, norm_of.count(vi) ? norm_of[vi] : vi, norm_of.count(vi) != 0, 0});
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
    if (guard_on && audit_src.size() == 1 && audit_src[0].fused) {
        // ONE guarded launch and it ends in the Normalize vertex: the launch gives the verdict itself (BandScanDesc::nz_acc)
        const size_t o = audit_src[0].desc_off;
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
    } else if (guard_on && !audit_src.empty()) {
        g->guard.chunk_audited = true;
        std::vector<AuditDesc> ad;
        for (const AuditSrc& a : audit_src) {
            AuditDesc x{};
            x.n_wt = a.n_wt;
            x.nb = (uint32_t)nb;
            x.bl = (uint32_t)bl;
            x.gain = (float)down[a.from].g;
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

static size_t desc_size(int fam) {
    switch (fam) {
        case F_LOOP: return sizeof(LoopDesc);
        case F_MULTI: return sizeof(MultiDesc);
        case F_LERP: return sizeof(LerpDesc);
        case F_SINE: return sizeof(SineDesc);
        case F_SYNTH: return sizeof(SynthDesc);
        case F_SAMPSYN: return sizeof(SampsynDesc);
        case F_ENV: return sizeof(AdsrVDesc);
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
static bool is_band_family(int fam) { return fam == F_BAND_SPEC || fam == F_BAND_FIX || fam == F_BAND_FILL; }

// Steps 3 and 4 for everything compiled into `cb`: patch pointers, upload the tables (skipped when the device
// copy is already byte-identical), launch level by level on `stream`.  With several graphs in `cb` (a batch)
// the launches are first merged: same level, family, launch parameters -> ONE grid whose blockIdx.y runs over
// the descriptors of all the graphs (their descriptors are copied into one contiguous array behind the tables).
// `fork_g`: the graph whose aux streams carry the independent launch families of a level (branch streams; single
// graph only).  *scratch_base = device address the scratch offsets of this submission refer to.
static int submit_chunk(Arena& ar, ChunkBuild& cb, hipStream_t stream, ProfCtx& prof, td_graph* fork_g,
                        const uint8_t** scratch_base, double* host_ms /* [2]: upload, launches */) {
    const auto tp2 = std::chrono::steady_clock::now();
    Staging& st = *cb.st;
    std::vector<Launch>& launches = cb.launches;
    prof.now = prof.every && (prof.count++ % prof.every) == 0;
    struct Copy { size_t dst, src, bytes; };
    std::vector<Copy> copies;
    if (cb.n_graphs > 1) {
        auto key_less = [](const Launch& a, const Launch& b) {
            if (a.level != b.level) return a.level < b.level;
            if (a.fam != b.fam) return a.fam < b.fam;
            if (a.M != b.M) return a.M < b.M;
            if (a.bl != b.bl) return a.bl < b.bl;
            if (a.is_scan != b.is_scan) return a.is_scan < b.is_scan;
            if (!is_band_family(a.fam) && a.aux != b.aux) return a.aux < b.aux;
            return false;
        };
        std::stable_sort(launches.begin(), launches.end(), key_less);
        std::vector<Launch> merged;
        merged.reserve(launches.size());
        for (size_t i = 0; i < launches.size();) {
            size_t j = i + 1;
            int n = launches[i].n;
            uint32_t aux = launches[i].aux;
            while (j < launches.size() && !key_less(launches[i], launches[j]) && !key_less(launches[j], launches[i])) {
                n += launches[j].n;
                aux = std::max(aux, launches[j].aux);   // (equal unless a band family: its largest segment count)
                ++j;
            }
            Launch L = launches[i];
            if (j - i > 1) {
                const size_t dsz = desc_size(L.fam);
                size_t dst = st.alloc((size_t)n * dsz);
                L.off = dst;
                L.n = n;
                L.aux = aux;
                for (size_t q = i; q < j; ++q) {
                    copies.push_back({dst, launches[q].off, (size_t)launches[q].n * dsz});
                    dst += (size_t)launches[q].n * dsz;
                }
            }
            merged.push_back(L);
            i = j;
        }
        launches.swap(merged);
    }
    // ---- 3. upload
    const size_t upload = (st.b.size() + 255) & ~(size_t)255;
    const size_t sync_at = upload + ((cb.scratch_bytes + 255) & ~(size_t)255);
    const size_t esync_at = sync_at + ((cb.sync_bytes + 255) & ~(size_t)255);
    if (!ensure_arena(ar, esync_at + cb.esync_bytes + 256, stream)) return 0;
    for (auto& f : cb.sync_fix) {
        uint64_t p = (uint64_t)(uintptr_t)(ar.d + sync_at + f.off);
        memcpy(&st.b[f.at], &p, 8);
    }
    for (auto& f : cb.esync_fix) {
        uint64_t p = (uint64_t)(uintptr_t)(ar.d + esync_at + f.off);
        memcpy(&st.b[f.at], &p, 8);
    }
    for (size_t at : cb.flag_fix) {
        uint64_t p = (uint64_t)(uintptr_t)ar.d_flag;
        memcpy(&st.b[at], &p, 8);
    }
    for (auto& f : cb.table_fix) {
        uint64_t p = (uint64_t)(uintptr_t)(ar.d + f.off);
        memcpy(&st.b[f.at], &p, 8);
    }
    for (auto& f : cb.scratch_fix) {
        uint64_t p = (uint64_t)(uintptr_t)(ar.d + upload + f.off);
        memcpy(&st.b[f.at], &p, 8);
    }
    for (auto& c : copies) memcpy(&st.b[c.dst], &st.b[c.src], c.bytes);
    *scratch_base = ar.d + upload;
    if (ar.inflight) TD_HIP(hipEventSynchronize(ar.copied));
    ar.inflight = false;
    // re-rendering unchanged projects from the same state compiles to byte-identical tables: the copy
    // already on the device is reused (kernels never write the uploaded region)
    bool arena_same = true;
    if (!(ar.valid == st.b.size() && memcmp(ar.h, st.b.data(), st.b.size()) == 0)) {
        arena_same = false;
        memcpy(ar.h, st.b.data(), st.b.size());
        TD_HIP(hipMemcpyAsync(ar.d, ar.h, st.b.size(), hipMemcpyHostToDevice, stream));
        TD_HIP(hipEventRecord(ar.copied, stream));
        ar.inflight = true;
        ar.valid = st.b.size();
    }
    // HIP-graph replay (td_graph_set_option "graph_replay"): unchanged uploaded bytes + unchanged launch list = the very
    // same kernel launches with the very same arguments; the captured sequence of the last submission is replayed by one
    // hipGraphLaunch instead of being issued launch by launch.
    const bool want_graph = fork_g && fork_g->graph_replay && !fork_g->branch_streams && !prof.now;
    std::vector<uint64_t> gkey;
    if (want_graph) {
        gkey.reserve(launches.size() * 4 + cb.zero.size() * 2 + 2);
        gkey.push_back(upload);
        gkey.push_back(sync_at);
        gkey.push_back(cb.sync_bytes);
        gkey.push_back(cb.esync_bytes);
        gkey.push_back(cb.one_grid_sources ? 1u : 0u);
        for (auto& z : cb.zero) { gkey.push_back(z.off); gkey.push_back(z.bytes); }
        for (auto& L : launches) {
            gkey.push_back(((uint64_t)(uint32_t)L.fam << 32) | (uint32_t)L.n);
            gkey.push_back(L.off);
            gkey.push_back(((uint64_t)L.aux << 32) | (uint32_t)L.level);
            gkey.push_back(((uint64_t)L.M << 32) | ((uint64_t)L.bl << 1) | (uint64_t)(L.is_scan & 1));
        }
        if (arena_same && ar.graph_exec && gkey == ar.graph_key) {
            const auto tp3r = std::chrono::steady_clock::now();
            ar.pending_fix = ar.graph_pending;   // (the replayed launches carry the same deferred check)
            if (!ar.pending_fix.empty()) note_pending(ar, stream, cur_device());
            TD_HIP(hipGraphLaunch(ar.graph_exec, stream));
            host_ms[0] += ms_between(tp2, tp3r);
            host_ms[1] += ms_between(tp3r, std::chrono::steady_clock::now());
            return 1;
        }
        if (ar.graph_exec) { (void)hipGraphExecDestroy(ar.graph_exec); ar.graph_exec = nullptr; }
        ar.graph_key.clear();
        TD_HIP(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
    }
    // An error return between BeginCapture and EndCapture would leave the stream capturing (every later call on it then
    // fails): this guard ends the capture and drops the partial graph on any exit that has not completed it.
    struct CaptureGuard {
        hipStream_t s;
        bool active;
        ~CaptureGuard() {
            if (!active) return;
            hipGraph_t partial = nullptr;
            (void)hipStreamEndCapture(s, &partial);
            if (partial) (void)hipGraphDestroy(partial);
            (void)hipGetLastError();
        }
    } capture_guard{stream, want_graph};
    for (auto& z : cb.zero) TD_HIP(hipMemsetAsync(ar.d + upload + z.off, 0, z.bytes, stream));
    // The launches of a level that read no edge buffer -- affine Synth, wavetable voice, SampleLerp, the envelope buffers -- go
    // out as ONE grid (k_sources, kernels.hip): the first launch of each kind, the longest-running kind first.  `pick`: the
    // launch index per SourceKind, -1: none; returns the common frame count, 0: no such grid for launches [li, lj).
    auto sources_grid = [&](size_t li, size_t lj, long pick[4]) -> uint32_t {
        pick[0] = pick[1] = pick[2] = pick[3] = -1;
        if (!cb.one_grid_sources || lj - li > 64) return 0u;
        uint32_t M0 = 0;
        int found = 0, n_desc = 0;
        for (size_t q = li; q < lj; ++q) {
            const Launch& L = launches[q];
            int kd = -1;
            switch (L.fam) {
                case F_SYNTH: kd = (L.aux & 1u) ? (int)SRC_SYNTH_AFFINE : -1; break;
                case F_SAMPSYN: kd = (int)SRC_SAMPSYN; break;
                case F_LERP: kd = (int)SRC_LERP; break;
                case F_ENV: kd = (int)SRC_ENV; break;
                default: break;
            }
            if (kd < 0 || pick[kd] >= 0 || !L.n || !L.M) continue;
            if (found && L.M != M0) continue;
            M0 = L.M;
            pick[kd] = (long)q;
            n_desc += L.n;
            ++found;
        }
        // (a large batch gains nothing -- its grids have no ramp or tail to speak of -- and the envelope part would run at the
        // Synth part's register budget: 32 config-3 projects measured 2 % slower in one grid, 8 the same)
        return (found >= 2 && n_desc <= 16) ? M0 : 0u;
    };
    // The zeroed hand-off words: by a fill kernel of their own -- or, when the submission opens with a k_sources grid, by that
    // grid's threads (everything that reads the words is behind it on the stream): one launch less in front of a scan render.
    bool zero_in_sources = false;
    if (cb.sync_bytes) {
        size_t l0 = 0;
        while (l0 < launches.size() && launches[l0].level == launches[0].level) ++l0;
        long pick0[4];
        const bool forks = fork_g && fork_g->branch_streams;
        zero_in_sources = !forks && !launches.empty() && cb.sync_bytes <= ((size_t)1 << 28) && sources_grid(0, l0, pick0) != 0u;
        if (!zero_in_sources) TD_HIP(hipMemsetAsync(ar.d + sync_at, 0, cb.sync_bytes, stream));
    }
    // The tile words of the stand-alone single-pass Normalize launches carry the submission's EPOCH beside their value (a
    // kernel argument: the descriptors stay byte-identical from render to render and are not uploaded again).  A word of an
    // earlier submission never compares equal, so the region is not zeroed between launches -- the memset was 2 us of the
    // headline render's 66.  It is zeroed once whenever it lies elsewhere than last time (what was there before is not
    // known to be words), when the arena is new, and before the epoch counter would wrap.  A captured submission
    // (graph_replay) replays its arguments: it keeps zeroed words and tag 1.
    uint32_t sum_tag = 1u;
    if (!cb.esync_bytes || want_graph) ar.esync_len = 0;   // (this submission may write anything where the words lay)
    if (cb.esync_bytes) {
        if (want_graph) {
            TD_HIP(hipMemsetAsync(ar.d + esync_at, 0, cb.esync_bytes, stream));
        } else {
            if (ar.esync_at != esync_at || ar.esync_len != cb.esync_bytes || ar.epoch == 0xFFFFFFFFu) {
                TD_HIP(hipMemsetAsync(ar.d + esync_at, 0, cb.esync_bytes, stream));
                ar.esync_at = esync_at;
                ar.esync_len = cb.esync_bytes;
                if (ar.epoch == 0xFFFFFFFFu) ar.epoch = 1u;
            }
            sum_tag = ++ar.epoch;
        }
    }

    // ---- 4. launch, level by level
    const auto tp3 = std::chrono::steady_clock::now();
    // (what an un-settled earlier submission left: prepare_render has settled it where its result is still needed.  If its
    // host-visible word is already up, the word is cleared ON THE STREAM -- behind that submission's launches, in front of this
    // one's -- so that it does not count against this submission at the next settle; the normal case is one host load.  A word
    // raised after this look costs one unnecessary k_norm_fix later, never a wrong result.)
    if (!ar.pending_fix.empty() && ar.h_flag && *(volatile uint32_t*)ar.h_flag)
        TD_HIP(hipMemsetD32Async((hipDeviceptr_t)ar.d_flag, 0, 1, stream));
    ar.pending_fix.clear();
    size_t li = 0;
    while (li < launches.size()) {
        size_t lj = li;
        uint32_t groups = 0;   // bit 0: main chain, bit 1 + i: aux stream i
        while (lj < launches.size() && launches[lj].level == launches[li].level) {
            const int a = aux_stream_of(launches[lj].fam);
            groups |= a < 0 ? 1u : (2u << a);
            ++lj;
        }
        const bool fork = fork_g && fork_g->branch_streams && __builtin_popcount(groups) > 1;
        if (fork) {
            TD_HIP(hipEventRecord(fork_g->ev_fork, stream));
            for (int a = 0; a < td_graph::kAuxStreams; ++a)
                if (groups & (2u << a)) TD_HIP(hipStreamWaitEvent(fork_g->aux[a], fork_g->ev_fork, 0));
        }
        uint64_t in_one_grid = 0;   // bit q - li: launched as a part of the level's k_sources grid
        if (!fork) {
            long pick[4];
            const uint32_t M0 = sources_grid(li, lj, pick);
            if (M0) {
                SourceParts P{};
                for (int kd = 0; kd < 4; ++kd) {
                    if (pick[kd] < 0) continue;
                    const Launch& L = launches[(size_t)pick[kd]];
                    const void* d = ar.d + L.off;
                    switch (kd) {
                        case SRC_SYNTH_AFFINE: P.synth = (const SynthDesc*)d; P.n_synth = L.n; break;
                        case SRC_SAMPSYN: P.sampsyn = (const SampsynDesc*)d; P.n_sampsyn = L.n; break;
                        case SRC_LERP: P.lerp = (const LerpDesc*)d; P.n_lerp = L.n; break;
                        default: P.env = (const AdsrVDesc*)d; P.n_env = L.n; break;
                    }
                    in_one_grid |= 1ull << ((size_t)pick[kd] - li);
                }
                const bool z = zero_in_sources && li == 0;   // (the submission's first grid clears the hand-off words)
                Prof pr(prof, F_SOURCES, stream);
                launch_sources(P, M0, z ? ar.d + sync_at : nullptr, z ? (cb.sync_bytes + 15) & ~(size_t)15 : 0, stream);
            }
        }
        for (size_t q = li; q < lj; ++q) {
            const Launch& L = launches[q];
            if (in_one_grid & (1ull << (q - li))) continue;
            if (L.fam == F_NORMFIX && (L.aux & 1u)) {   // deferred: launched by settle_arena() only if a tile raised the host-visible word
                ar.pending_fix.push_back({L.off, L.n, L.M, L.bl});
                continue;
            }
            const void* d = ar.d + L.off;
            const int a = aux_stream_of(L.fam);
            hipStream_t s = (fork && a >= 0) ? fork_g->aux[a] : stream;
            Prof pr(prof, L.fam, s);
            switch (L.fam) {
                case F_LOOP: launch_sample_loop((const LoopDesc*)d, L.n, L.M, s); break;
                case F_MULTI: launch_sample_multi((const MultiDesc*)d, L.n, L.M, s); break;
                case F_LERP: launch_sample_lerp((const LerpDesc*)d, L.n, L.M, s); break;
                case F_SINE: launch_debug_sine((const SineDesc*)d, L.n, L.M, L.bl, s); break;
                case F_SYNTH: launch_synth((const SynthDesc*)d, L.n, L.M, (L.aux & 1u) != 0u, s); break;
                case F_SAMPSYN: launch_sampsyn((const SampsynDesc*)d, L.n, L.M, s); break;
                case F_ENV: launch_adsr_env((const AdsrVDesc*)d, L.n, L.M, s); break;
                case F_SUM:
                    if (L.aux >> 12) launch_norm1((const SumDesc*)d, L.n, L.M, L.aux & 0xFFu, (int)(L.aux >> 12), sum_tag, s);   // single-pass Normalize, narrow forms
                    else launch_sum((const SumDesc*)d, L.n, L.M, L.bl, L.aux & 0xFFu, (L.aux & 0x100u) != 0u, (L.aux & 0x200u) != 0u, sum_tag, s);
                    break;
                case F_SCALE: launch_scale((const ScaleDesc*)d, L.n, L.M, L.bl, L.is_scan, s); break;
                case F_NORMFIX: launch_norm_fix((const SumDesc*)d, L.n, L.M, L.bl, s); break;
                case F_ADSR: launch_adsr((const AdsrVDesc*)d, L.n, L.M, L.aux & 0xFFu, s); break;
                case F_BAND: launch_band_pass((const BandDesc*)d, L.n, L.M, s); break;
                case F_BAND_SPEC: launch_band_spec((const BandSpecDesc*)d, L.n, L.M, L.aux, s); break;
                case F_BAND_FIX: launch_band_fix((const BandSpecDesc*)d, L.n, L.M, L.aux, s); break;
                case F_BAND_FILL: break;
                case F_BAND_SCAN:
                    if (L.aux & 0x10000u) launch_band_chain((const BandScanDesc*)d, L.n, L.M, L.aux & 0xFFu, (L.aux & 0x20000u) != 0u, s);
                    else launch_band_scan((const BandScanDesc*)d, L.n, L.M, L.aux & 0xFFu, (int)((L.aux >> 8) & 0xFFu), s);
                    break;
                case F_QUANT: launch_quantise((const QuantDesc*)d, L.n, L.M, s); break;
                case F_AUDIT: launch_band_audit((const AuditHead*)d, L.n, s); break;
            }
        }
        if (fork) {
            for (int a = 0; a < td_graph::kAuxStreams; ++a)
                if (groups & (2u << a)) {
                    TD_HIP(hipEventRecord(fork_g->ev_join[a], fork_g->aux[a]));
                    TD_HIP(hipStreamWaitEvent(stream, fork_g->ev_join[a], 0));
                }
        }
        li = lj;
    }
    if (want_graph) {
        hipGraph_t graph = nullptr;
        capture_guard.active = false;
        TD_HIP(hipStreamEndCapture(stream, &graph));
        hipGraphExec_t exec = nullptr;
        const hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (ie != hipSuccess) return fail(std::string("HIP error: ") + hipGetErrorString(ie) + " at hipGraphInstantiate");
        ar.graph_exec = exec;
        ar.graph_key.swap(gkey);
        ar.graph_pending = ar.pending_fix;
        TD_HIP(hipGraphLaunch(ar.graph_exec, stream));
    }
    TD_HIP(hipGetLastError());
    if (!ar.pending_fix.empty()) note_pending(ar, stream, cur_device());
    const auto tp4 = std::chrono::steady_clock::now();
    host_ms[0] += ms_between(tp2, tp3);   // arena upload (or the compare that skips it)
    host_ms[1] += ms_between(tp3, tp4);   // launches
    return 1;
}

// A render of n_blocks blocks runs in chunks of whole blocks (normally one).  The three pieces below are shared by the
// single-graph path (graph_render_chunks) and the batch path (batch_render_chunks), which interleaves them over its graphs.
struct RenderPlan {
    size_t n_blocks = 0, chunk_blocks = 1, total = 0;
    bool multi = false, want_pcm = false;
    int qmode = 0, bits = 16;
    float amplitude = 0.f;
    size_t word = 0;
};
static int prepare_render(td_graph* g, size_t n_blocks, int bits, bool want_pcm, RenderPlan* rp) {
    if (!ensure_graph_device(g)) return 0;
    if (g->output_vertex < 0) return fail("TermDaw: error: output vertex not found.");
    if (g->plan_dirty) build_plan(g);
    if (!ensure_state_slots(g)) return 0;
    const size_t bl = g->bl;
    if (bl == 0) return fail("termdaw_amd: buffer length 0");
    rp->n_blocks = n_blocks;
    rp->total = n_blocks * bl;
    rp->chunk_blocks = std::max<size_t>(1, g->max_chunk_frames / bl);
    rp->chunk_blocks = std::min(rp->chunk_blocks, std::max<size_t>(n_blocks, 1));
    rp->multi = n_blocks > rp->chunk_blocks;
    rp->want_pcm = want_pcm;
    rp->bits = bits;
    if (want_pcm) {
        if (!(bits == 8 || bits == 16 || bits == 24 || bits == 32))   // state.rs:495-501
            return fail("Bitdepth not supported: choose bitdepth in {8, 16, 24, 32}.");
        rp->qmode = bits > 16 ? 2 : 1;                                    // write_16s / write_32s, state.rs:567-571
        rp->amplitude = bits < 32 ? (float)((1 << (bits - 1)) - 1) : (float)INT32_MAX;   // state.rs:515-516
        rp->word = rp->qmode == 1 ? 2 : 4;
        const size_t need = rp->total * 2 * rp->word + 64;
        if (need > g->pcm_cap) {
            if (!drain(g)) return 0;
            if (g->d_pcm && !g->pcm_borrowed) { (void)hipFree(g->d_pcm); g->device_bytes -= g->pcm_cap; }
            g->d_pcm = nullptr;
            g->pcm_cap = 0;
            g->pcm_borrowed = false;
            TD_HIP(hipMalloc(&g->d_pcm, need));
            g->pcm_cap = need;
            g->device_bytes += need;
        }
        g->pcm_bytes = rp->total * 2 * rp->word;
    }
    if (rp->multi) {
        const size_t need = (rp->total + 2) * sizeof(float2);
        if (need > g->out_f32_cap) {
            if (!drain(g)) return 0;
            if (g->d_out_f32) { (void)hipFree(g->d_out_f32); g->device_bytes -= g->out_f32_cap; }
            g->d_out_f32 = nullptr;
            g->out_f32_cap = 0;
            TD_HIP(hipMalloc(&g->d_out_f32, need));
            g->out_f32_cap = need;
            g->device_bytes += need;
        }
    }
    return 1;
}
// Snapshots the FlowwBank cursor of the next nb blocks (advancing the bank one block at a time exactly like
// state.rs:572 / graph.rs:232) and compiles the chunk into cb.
static int compile_next_chunk(td_graph* g, const td_samplebank* sb, td_flowwbank* fb, const RenderPlan& rp, size_t done,
                              size_t nb, bool is_scan, bool advance_graph_time, size_t scan_t0, ChunkBuild& cb) {
    const size_t bl = g->bl;
    std::vector<BlockCursor>& cur = g->cursor;   // (capacity kept from render to render)
    const size_t nfl = fb->start_indices.size();
    cur.resize(nb);
    g->cursor_starts.resize(nb * nfl + 1);
    for (size_t b = 0; b < nb; ++b) {
        size_t* s = g->cursor_starts.data() + b * nfl;
        for (size_t i = 0; i < nfl; ++i) s[i] = fb->start_indices[i];
        cur[b] = {fb->frame, s, nfl};
        fb->set_time_to_next_block();
    }
    const uint64_t t0 = advance_graph_time ? g->t : scan_t0 + done * bl;
    void* pcm_dst = rp.want_pcm ? (uint8_t*)g->d_pcm + done * bl * 2 * rp.word : nullptr;
    return compile_chunk(g, sb, fb, cur, t0, is_scan, pcm_dst, rp.qmode, rp.amplitude, cb);
}
static int finish_chunk(td_graph* g, const RenderPlan& rp, size_t done, size_t nb, bool advance_graph_time,
                        const uint8_t* scratch_base) {
    g->band_stats_base = scratch_base;
    if (advance_graph_time) g->t += nb * g->bl;
    if (rp.multi)
        TD_HIP(hipMemcpyAsync(g->d_out_f32 + done * g->bl, g->vbuf[(size_t)g->output_vertex], nb * g->bl * sizeof(float2),
                              hipMemcpyDeviceToDevice, g->stream));
    return 1;
}
static void finish_render(td_graph* g, const RenderPlan& rp) {
    g->last_out_f32 = rp.multi ? g->d_out_f32 : (rp.n_blocks ? g->vbuf[(size_t)g->output_vertex] : nullptr);
    if (!g->output_f32 && rp.want_pcm && g->vertices[(size_t)g->output_vertex].kind == K_NORMALIZE)
        g->last_out_f32 = nullptr;   // (the f32 frames of the output were never written)
    g->last_frames = rp.total;
    g->last_bits = rp.bits;
}

// Everything queued for the graph has completed AND a deferred k_norm_fix has run if one was called for (settle_arena): the
// point from which results -- PCM, f32 frames, carried Normalize state -- may be read.
static int guard_settle(td_graph* g);
static int graph_set_time_impl(td_graph* g, size_t time);
static int drain(td_graph* g) {
    if (!g->stream) return 1;
    if (!ensure_device(g->device)) return 0;
    if (!settle_arena(g->arena, g->stream)) return 0;
    if (g->batch && g->batch->stream && !settle_arena(g->batch->arena, g->batch->stream)) return 0;
    return guard_settle(g);
}
// A render whose output Normalize vertex continues from its carried max needs the previous render's deferred fix settled
// first (a render that starts from reset_normalization does not read it: back-to-back fresh renders never wait here).
// ---- the guard (band_mode 2): engine.h tde::Guard
// Does the render about to start read any carried device state?  Not if every reachable vertex with a state slot starts
// afresh: a Normalize vertex with reset_normalization pending, a band-pass vertex with set_time pending.
static bool starts_afresh(const td_graph* g) {
    for (size_t vi : g->order) {
        const Vertex& v = g->vertices[vi];
        if (v.kind == K_NORMALIZE && !v.has_init_override) return false;
        if (v.kind == K_BAND_PASS && v.state_slot >= 0 && !v.first_pending) return false;
    }
    return true;
}
static bool has_reachable_band(const td_graph* g) {
    for (size_t vi : g->order)
        if (g->vertices[vi].kind == K_BAND_PASS) return true;
    return false;
}
// In front of a render that may carry an audit: what it takes to do the render again (called once the plan and the state
// slots are in place, before the first chunk compiles).
static int guard_begin(td_graph* g, const td_samplebank* sb, td_flowwbank* fb, size_t n_blocks, bool is_scan, int bits,
                       bool advance, size_t scan_t0, bool want_pcm) {
    Guard& q = g->guard;
    q.sb = sb; q.fb = fb; q.n_blocks = n_blocks; q.is_scan = is_scan; q.bits = bits; q.advance = advance; q.scan_t0 = scan_t0;
    q.want_pcm = want_pcm;
    q.post = 0;
    q.snap.take(g, fb);
    q.have_backup = false;
    if (!starts_afresh(g) && g->dstate && !g->hstate.empty()) {
        const size_t n = g->hstate.size();
        if (n > q.backup_cap) {
            if (q.d_backup) (void)hipFree(q.d_backup);
            q.d_backup = nullptr;
            q.backup_cap = 0;
            TD_HIP(hipMalloc(&q.d_backup, n * 2 * sizeof(StateSlot)));
            q.backup_cap = n * 2;
        }
        TD_HIP(hipMemcpyAsync(q.d_backup, g->dstate, n * sizeof(StateSlot), hipMemcpyDeviceToDevice, g->stream));
        q.have_backup = true;
    }
    return 1;
}
// The stream has drained: look at the verdict of the last guarded render, and do that render again with the exact kernels
// if its estimate was over the bound.
static int guard_settle(td_graph* g) {
    Guard& q = g->guard;
    if (!q.armed || q.in_redo || !q.h_word) return 1;
    q.armed = false;
    const uint32_t raised = *(volatile uint32_t*)q.h_word;
    const uint32_t bits = *(volatile uint32_t*)(q.h_word + 1);
    memcpy(&q.last_est, &bits, 4);
    if (q.last_est > q.max_est || !(q.last_est == q.last_est)) q.max_est = q.last_est;
    if (!raised) return 1;
    *(volatile uint32_t*)q.h_word = 0u;
    q.in_redo = true;
    const int mode = g->band_mode;
    g->band_mode = 0;
    q.snap.put(g, q.fb);
    int ok = 1;
    if (q.have_backup && hipMemcpyAsync(g->dstate, q.d_backup, g->hstate.size() * sizeof(StateSlot), hipMemcpyDeviceToDevice, g->stream) != hipSuccess)
        ok = fail("HIP error: the guard could not restore the carried state");
    g->state_dev_dirty = true;
    if (ok) ok = graph_render_chunks(g, q.sb, q.fb, q.n_blocks, q.is_scan, q.bits, q.advance, q.scan_t0, q.want_pcm);
    if (ok && q.post == 1) ok = graph_set_time_impl(g, 0);
    if (ok && q.post == 2) { q.fb->frame = q.snap.fb_frame; q.fb->start_indices = q.snap.fb_start; }
    g->band_mode = mode;
    q.in_redo = false;
    q.redos += 1;
    if (!ok) return 0;
    return settle_arena(g->arena, g->stream);   // (the second render's own deferred check, and its completion)
}

static int settle_before_render(td_graph* g) {
    // (a guarded render's verdict is still out and this render continues from the state it left: settle it first)
    if (g->guard.armed && !g->guard.in_redo && !(g->plan_dirty ? false : starts_afresh(g)) && !drain(g)) return 0;
    const bool pending = !g->arena.pending_fix.empty() || (g->batch && !g->batch->arena.pending_fix.empty());
    if (!pending || g->output_vertex < 0) return 1;
    const Vertex& ov = g->vertices[(size_t)g->output_vertex];
    if (ov.kind == K_NORMALIZE && ov.has_init_override) return 1;
    return drain(g);
}

// What compiling a chunk changes on the host side of a project -- loop cursors, the carried state of event-driven vertices,
// a pending reset_normalization, the playhead, the FlowwBank cursor -- so that a step that fails while compiling a LATER project
// of the batch (or a later chunk) can put every project back where the step found it: a retry then renders the same thing.
void HostSnapshot::take(const td_graph* g, const td_flowwbank* fb) {
    t = g->t;
    fb_frame = fb->frame;
    fb_start = fb->start_indices;
    v.resize(g->vertices.size());
    for (size_t i = 0; i < v.size(); ++i) {
        const Vertex& x = g->vertices[i];
        v[i].loop_t = x.loop_t;
        v[i].has_init_override = x.has_init_override;
        v[i].peak_known = x.peak_known;
        v[i].first_pending = x.first_pending;
        v[i].init_override = x.init_override;
        v[i].state.clear();
        save_state(x, v[i].state);
    }
}
void HostSnapshot::put(td_graph* g, td_flowwbank* fb) const {
    g->t = t;
    fb->frame = fb_frame;
    fb->start_indices = fb_start;
    for (size_t i = 0; i < v.size() && i < g->vertices.size(); ++i) {
        Vertex& x = g->vertices[i];
        x.loop_t = v[i].loop_t;
        x.has_init_override = v[i].has_init_override;
        x.peak_known = v[i].peak_known;
        x.first_pending = v[i].first_pending;
        x.init_override = v[i].init_override;
        load_state(x, v[i].state);
    }
}

// Renders n_blocks blocks in chunks.  advance_graph_time: Graph::render semantics (t += bl per block);
// otherwise the scan's explicit j*bl clock starting at scan_t0 (graph.rs:229-233).
int graph_render_chunks(td_graph* g, const td_samplebank* sb, td_flowwbank* fb, size_t n_blocks, bool is_scan,
                        int bits, bool advance_graph_time, size_t scan_t0, bool want_pcm) {
    RenderPlan rp;
    if (!settle_before_render(g)) return 0;
    if (!prepare_render(g, n_blocks, bits, want_pcm, &rp)) return 0;
    g->defer_fix = !rp.multi;   // (a later chunk reads the carried max; the f32 copy of a multi-chunk render reads the frames)
    const bool guarded = g->band_mode == 2 && !g->guard.in_redo && has_reachable_band(g);
    if (guarded && !guard_begin(g, sb, fb, n_blocks, is_scan, bits, advance_graph_time, scan_t0, want_pcm)) return 0;
    bool audited = false;
    ChunkBuild& cb = g->build;
    cb.st = &g->staging;
    // (a graph that belongs to a batch may still render alone: it then uses its own arena on the shared stream)
    size_t done = 0;
    while (done < n_blocks) {
        const size_t nb = std::min(rp.chunk_blocks, n_blocks - done);
        cb.clear();
        g->snapshot.take(g, fb);   // (a chunk that fails to compile or to submit leaves the host state where it found it)
        if (!compile_next_chunk(g, sb, fb, rp, done, nb, is_scan, advance_graph_time, scan_t0, cb)) {
            g->snapshot.put(g, fb);
            return 0;
        }
        audited = audited || (guarded && g->guard.chunk_audited);
        const uint8_t* scratch_base = nullptr;
        if (!submit_chunk(g->arena, cb, g->stream, g->prof, g, &scratch_base, &g->host_ms[2])) {
            g->snapshot.put(g, fb);
            return 0;
        }
        g->host_chunks += 1;
        if (!finish_chunk(g, rp, done, nb, advance_graph_time, scratch_base)) return 0;
        done += nb;
    }
    finish_render(g, rp);
    if (audited) { g->guard.armed = true; g->guard.audits += 1; }
    return 1;
}

static int graph_set_time_impl(td_graph* g, size_t time) {   // graph.rs:123-128 + extensions.rs:196-204
    g->t = time;
    bool any_band = false;
    for (auto& v : g->vertices) {
        switch (v.kind) {
            case K_SAMPLE_LOOP: v.loop_t = time; break;
            case K_DEBUG_SINE: v.sine_notes.clear(); break;
            case K_SYNTH: v.notes.clear(); break;
            case K_BAND_PASS:
                if (v.state_slot >= 0) {
                    g->hstate[v.state_slot].band.first = 1u;
                    v.first_pending = true;
                    any_band = true;
                }
                break;
            default: break;
        }
    }
    (void)any_band;   // (no device work: the next submission's descriptors carry the vertices' first_override)
    return 1;
}

// ------------------------------------------------------------------------------------------------
// batch: many independent projects per submission (BASELINE config 5)
// ------------------------------------------------------------------------------------------------
// The same chunk loop as graph_render_chunks, interleaved over the graphs of the batch: every graph compiles its
// chunk into the batch's ChunkBuild, ONE submission uploads the tables and launches the merged grids.  Graphs
// may differ in everything (structure, block length, chunk cap); launches merge only where level, family and
// launch parameters agree.
// Projects [lo, hi) of the batch.  allow_defer: nothing is submitted through the batch's arena before the caller settles it
// (a deferred k_norm_fix lives in the arena's LAST submission only).
static int batch_render_range(td_batch* b, size_t lo, size_t hi, size_t n_blocks, bool is_scan, int bits, bool advance_graph_time,
                              bool want_pcm, bool allow_defer) {
    if (hi <= lo) return 1;
    if (!ensure_device(b->device)) return 0;
    const size_t P = hi - lo;
    std::vector<RenderPlan> rp(P);
    for (size_t i = 0; i < P; ++i) {
        td_graph* g = b->graphs[lo + i];
        if (!settle_before_render(g)) return 0;
        if (!prepare_render(g, n_blocks, bits, want_pcm, &rp[i])) return 0;
    }
    // A deferred k_norm_fix lives in the arena's LAST submission only: where any project of the range takes several chunks --
    // several submissions through this arena -- no project's check may be deferred (the later submissions would drop it).
    bool any_multi = false;
    for (size_t i = 0; i < P; ++i) any_multi = any_multi || rp[i].multi;
    for (size_t i = 0; i < P; ++i) {
        td_graph* g = b->graphs[lo + i];
        g->defer_fix = allow_defer && !any_multi;
        if (g->band_mode == 2 && !g->guard.in_redo && has_reachable_band(g) &&
            !guard_begin(g, b->sbs[lo + i], b->fbs[lo + i], n_blocks, is_scan, bits, advance_graph_time, 0, want_pcm)) return 0;
    }
    ChunkBuild& cb = b->build;
    cb.st = &b->staging;
    std::vector<size_t> done(P, 0), nb(P, 0);
    for (;;) {
        const auto t0 = std::chrono::steady_clock::now();
        cb.clear();
        bool any = false;
        // (a failing step must not leave the projects compiled before the failure half-advanced: cursors, loop positions,
        // carried voices, a consumed reset_normalization all go back to where this step found them)
        auto roll_back = [&]() {
            for (size_t q = 0; q < P; ++q)
                if (nb[q]) b->graphs[lo + q]->snapshot.put(b->graphs[lo + q], b->fbs[lo + q]);
        };
        for (size_t i = 0; i < P; ++i) nb[i] = std::min(rp[i].chunk_blocks, n_blocks - done[i]);
        for (size_t i = 0; i < P; ++i)
            if (nb[i]) b->graphs[lo + i]->snapshot.take(b->graphs[lo + i], b->fbs[lo + i]);
        for (size_t i = 0; i < P; ++i) {
            if (!nb[i]) continue;
            any = true;
            if (!compile_next_chunk(b->graphs[lo + i], b->sbs[lo + i], b->fbs[lo + i], rp[i], done[i], nb[i], is_scan, advance_graph_time, 0, cb)) {
                roll_back();
                return 0;
            }
            // (this project's chunk carries an audit: its verdict is looked at when the batch is settled)
            if (b->graphs[lo + i]->guard.chunk_audited && !b->graphs[lo + i]->guard.armed) { b->graphs[lo + i]->guard.armed = true; b->graphs[lo + i]->guard.audits += 1; }
        }
        if (!any) break;
        const uint8_t* scratch_base = nullptr;
        b->host_ms[0] += ms_between(t0, std::chrono::steady_clock::now());
        if (!submit_chunk(b->arena, cb, b->stream, b->prof, nullptr, &scratch_base, &b->host_ms[2])) {
            roll_back();   // (host side only: what the device has already run of this step cannot be taken back)
            return 0;
        }
        for (size_t i = 0; i < P; ++i) {
            if (!nb[i]) continue;
            if (!finish_chunk(b->graphs[lo + i], rp[i], done[i], nb[i], advance_graph_time, scratch_base)) return 0;
            done[i] += nb[i];
        }
        b->host_steps += 1;
    }
    for (size_t i = 0; i < P; ++i) finish_render(b->graphs[lo + i], rp[i]);
    return 1;
}
static int batch_render_chunks(td_batch* b, size_t n_blocks, bool is_scan, int bits, bool advance_graph_time, bool want_pcm) {
    return batch_render_range(b, 0, b->graphs.size(), n_blocks, is_scan, bits, advance_graph_time, want_pcm, true);
}

}  // namespace tde

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

const char* td_last_error(void) { return g_error.c_str(); }

int td_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int td_set_device(int device) {
    if (!ensure_device(device)) return 0;
    t_device = device;
    return 1;
}

// ---- SampleBank ----
td_samplebank* td_samplebank_new(size_t sample_rate) {
    td_samplebank* sb = new td_samplebank();
    sb->sample_rate = sample_rate;
    sb->device = t_device;
    return sb;
}
void td_samplebank_free(td_samplebank* sb) {
    if (!sb) return;
    if (hipSetDevice(sb->device) == hipSuccess) {
        for (auto& e : sb->samples) { sb->release(e.d); sb->release(e.d16); }
        sb->release_all();
        if (sb->tmp) (void)hipFree(sb->tmp);
    }
    delete sb;
}
int td_samplebank_add_decoded(td_samplebank* sb, const char* name, const float* linear, size_t n, int channels,
                              size_t sample_rate, size_t bits, const char* method) {
    return bank_add_stream(sb, name, linear, nullptr, PCM_F32, n, channels, sample_rate, bits, method_from(method));
}
int td_samplebank_add_file(td_samplebank* sb, const char* name, const char* path, const char* method) {
    if (sb->names.count(name))
        return fail(std::string("TermDaw: SampleBank: there is already a sample with name \"") + name + "\" present.");
    tdw::WavRaw w;
    std::string err;
    if (!tdw::read_wav_raw(path, &w, &err)) return fail(err);
    const uint32_t fmt = w.is_float ? PCM_F32 : (w.bits == 8 ? PCM_U8 : w.bits == 16 ? PCM_S16 : w.bits == 24 ? PCM_S24 : PCM_S32);
    return bank_add_stream(sb, name, nullptr, w.bytes.data(), fmt, w.n_values, w.channels, w.sample_rate, w.bits, method_from(method));
}
long td_samplebank_get_index(const td_samplebank* sb, const char* name) {
    auto it = sb->names.find(name);
    return it == sb->names.end() ? -1 : (long)it->second;
}
size_t td_samplebank_sample_len(const td_samplebank* sb, size_t index) {
    return index < sb->samples.size() ? sb->samples[index].len : 0;
}
int td_samplebank_read(const td_samplebank* sb, size_t index, float* l, float* r) {
    if (index >= sb->samples.size()) return fail("sample index out of range");
    if (!ensure_device(sb->device)) return 0;
    const SampleEntry& e = sb->samples[index];
    std::vector<float2> tmp(e.len);
    TD_HIP(hipMemcpy(tmp.data(), e.d, e.len * sizeof(float2), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < e.len; ++i) {
        l[i] = tmp[i].x;
        r[i] = tmp[i].y;
    }
    return 1;
}
void td_samplebank_get_max_sr_bd(const td_samplebank* sb, size_t* max_sr, size_t* max_bd) {
    if (max_sr) *max_sr = sb->max_sr;
    if (max_bd) *max_bd = sb->max_bd;
}

// ---- FlowwBank ----
td_flowwbank* td_flowwbank_new(size_t sr, size_t bl) {
    td_flowwbank* fb = new td_flowwbank();
    fb->sr = sr;
    fb->bl = bl;
    return fb;
}
void td_flowwbank_free(td_flowwbank* fb) { delete fb; }
void td_flowwbank_reset(td_flowwbank* fb) {
    fb->frame = 0;
    fb->flowws.clear();
    fb->start_indices.clear();
    fb->names.clear();
    fb->stream_list.clear();
    fb->versions.clear();
}
long td_flowwbank_add_events(td_flowwbank* fb, const char* name, const td_event* events, size_t n) {
    fb->flowws.emplace_back(events, events + n);
    fb->versions.push_back(td_flowwbank::next_version());
    fb->start_indices.push_back(0);
    const size_t index = fb->flowws.size() - 1;
    fb->names[name] = index;
    return (long)index;
}
long td_flowwbank_declare_stream(td_flowwbank* fb, const char* name) {
    long i = td_flowwbank_add_events(fb, name, nullptr, 0);
    fb->stream_list.push_back((size_t)i);
    return i;
}
long td_flowwbank_add_midi(td_flowwbank* fb, const char* name, const char* path) {
    std::vector<td_event> ev;
    std::string err;
    if (!tde::read_midi_file(path, &ev, &err)) {
        fail(std::string("Could not read midi file: \"") + path + "\" (" + err + ").");   // floww.rs:45-46
        return -1;
    }
    return td_flowwbank_add_events(fb, name, ev.data(), ev.size());
}
long td_flowwbank_append_stream(td_flowwbank* fb, const char* name, const td_event* events, size_t n) {
    auto it = fb->names.find(name);
    if (it == fb->names.end()) return -1;
    auto& f = fb->flowws[it->second];
    f.insert(f.end(), events, events + n);
    fb->versions[it->second] = td_flowwbank::next_version();
    return (long)f.size();
}
void td_flowwbank_trim_streams(td_flowwbank* fb) {   // start_indices are not rewound (floww.rs:59-64)
    for (size_t index : fb->stream_list) {
        auto& f = fb->flowws[index];
        f.erase(f.begin(), f.begin() + (long)std::min(fb->start_indices[index], f.size()));
        fb->versions[index] = td_flowwbank::next_version();
    }
}
size_t td_flowwbank_get_events(const td_flowwbank* fb, size_t index, td_event* out, size_t cap) {
    if (index >= fb->flowws.size()) return 0;
    const auto& f = fb->flowws[index];
    for (size_t i = 0; i < f.size() && i < cap; ++i) out[i] = f[i];
    return f.size();
}
long td_flowwbank_get_index(const td_flowwbank* fb, const char* name) {
    auto it = fb->names.find(name);
    return it == fb->names.end() ? -1 : (long)it->second;
}
void td_flowwbank_set_time(td_flowwbank* fb, size_t t) { fb->set_time(t); }
void td_flowwbank_set_time_to_next_block(td_flowwbank* fb) { fb->set_time_to_next_block(); }

// ---- Graph ----
td_graph* td_graph_new(size_t max_buffer_len, size_t sr) {
    td_graph* g = new td_graph();
    g->bl = max_buffer_len;
    g->sr = sr;
    g->device = t_device;
    return g;
}
static void free_prof(ProfCtx& pc) {
    for (auto& e : pc.pending) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto e : pc.free_ev) (void)hipEventDestroy(e);
    pc.pending.clear();
    pc.free_ev.clear();
}
void td_graph_free(td_graph* g) {
    if (!g) return;
    // (a deferred k_norm_fix -- the graph's own arena's or its batch's -- holds pointers into this graph's state: settled first)
    g->guard.armed = false;   // (nobody will read a render of this graph again: its verdict is dropped, not acted on)
    if (g->stream && hipSetDevice(g->device) == hipSuccess) (void)drain(g);
    if (g->batch) {   // leave the batch first: it must not keep a dangling handle
        td_batch* b = g->batch;
        for (size_t i = 0; i < b->graphs.size(); ++i)
            if (b->graphs[i] == g) {
                b->graphs.erase(b->graphs.begin() + (long)i);
                b->sbs.erase(b->sbs.begin() + (long)i);
                b->fbs.erase(b->fbs.begin() + (long)i);
                break;
            }
    }
    const bool has_device_state = g->stream || !g->pool.empty() || !g->wavetables.empty() || g->dstate || g->arena.d || g->d_pcm ||
                                  g->d_out_f32 || g->d_resampled || g->d_scalar || g->ev_fork;
    if (has_device_state && hipSetDevice(g->device) == hipSuccess) {
        if (g->stream) (void)hipStreamSynchronize(g->stream);
        else (void)hipDeviceSynchronize();   // (a graph whose stream could not be re-made after td_batch_free)
        for (float2* p : g->pool) (void)hipFree(p);
        for (float* p : g->wavetables) (void)hipFree(p);
        free_tables(g);
        if (g->dstate) (void)hipFree(g->dstate);
        free_arena(g->arena);
        if (g->d_pcm && !g->pcm_borrowed) (void)hipFree(g->d_pcm);
        if (g->d_out_f32) (void)hipFree(g->d_out_f32);
        if (g->d_resampled) (void)hipFree(g->d_resampled);
        if (g->d_scalar) (void)hipFree(g->d_scalar);
        if (g->guard.d_backup) (void)hipFree(g->guard.d_backup);
        if (g->guard.h_word) (void)hipHostFree(g->guard.h_word);
        free_prof(g->prof);
        if (g->ev_fork) (void)hipEventDestroy(g->ev_fork);
        for (int a = 0; a < td_graph::kAuxStreams; ++a)
            if (g->aux[a]) {
                (void)hipEventDestroy(g->ev_join[a]);
                (void)hipStreamDestroy(g->aux[a]);
            }
        if (g->owns_stream && g->stream) (void)hipStreamDestroy(g->stream);
    }
    drop_pending(g->arena);   // (off the process-wide list whatever happened above: the list holds the arena's address)
    delete g;
}
void td_graph_reset(td_graph* g) {
    g->guard.armed = false;   // (the vertices a pending verdict is about are going)
    if ((g->stream || !g->wavetables.empty()) && hipSetDevice(g->device) == hipSuccess) {
        if (g->stream) (void)drain(g);   // (a deferred k_norm_fix belongs to the vertices about to go)
        for (float* p : g->wavetables) (void)hipFree(p);
        free_tables(g);
    }
    g->wavetables.clear();
    g->vertices.clear();
    g->edges.clear();
    g->name_map.clear();
    g->output_vertex = -1;
    g->t = 0;
    g->hstate.clear();
    g->state_host_dirty = true;
    g->state_dev_dirty = false;
    g->plan_dirty = true;
}

static Vertex& add_vertex(td_graph* g, const char* name, float gain, float angle, float wet, Kind kind) {
    Vertex v;
    v.kind = kind;
    v.name = name;
    v.gain = gain;
    v.angle = fmaxf(fminf(angle, 90.0f), -90.0f);   // graph.rs:255
    v.wet = fmaxf(fminf(wet, 1.0f), 0.0f);          // graph.rs:256
    g->vertices.push_back(v);
    g->edges.emplace_back();
    g->name_map[name] = g->vertices.size() - 1;     // later duplicates overwrite (graph.rs:54)
    g->plan_dirty = true;
    return g->vertices.back();
}
static int new_slot(td_graph* g) {
    // a freshly constructed slot makes the host mirror authoritative: fetch device values first
    pull_state(g);
    StateSlot s;
    memset(&s, 0, sizeof s);
    g->hstate.push_back(s);
    g->state_host_dirty = true;
    return (int)g->hstate.size() - 1;
}
static bool conf_from(const float* arr, int n, AdsrConfD* c) {   // adsr.rs:94-114
    if (n == 0) { *c = AdsrConfD{0, 0, 0, 0, 0, 0, 0, 0, 0}; return true; }
    if (n == 6) { *c = AdsrConfD{0.0f, arr[0], 1.0f, arr[1], arr[2], arr[3], arr[4], arr[5], 0.0f}; return true; }
    if (n == 9) { *c = AdsrConfD{arr[0], arr[1], arr[2], arr[3], arr[4], arr[5], arr[6], arr[7], arr[8]}; return true; }
    return false;
}

int td_graph_add_sum(td_graph* g, const char* name, float gain, float angle) {
    add_vertex(g, name, gain, angle, 0.0f, K_SUM);
    return 1;
}
int td_graph_add_normalize(td_graph* g, const char* name, float gain, float angle) {
    const int slot = new_slot(g);
    Vertex& v = add_vertex(g, name, gain, angle, 0.0f, K_NORMALIZE);
    v.state_slot = slot;
    g->hstate[slot].norm = {0.0f, 0.0f, 0u, 0u};   // extensions.rs:87-92
    return 1;
}
int td_graph_add_sampleloop(td_graph* g, const char* name, float gain, float angle, size_t sample_index) {
    Vertex& v = add_vertex(g, name, gain, angle, 0.0f, K_SAMPLE_LOOP);
    v.sample_index = sample_index;
    return 1;
}
int td_graph_add_sample_multi(td_graph* g, const char* name, float gain, float angle, size_t sample_index,
                              size_t floww_index, int note) {
    Vertex& v = add_vertex(g, name, gain, angle, 0.0f, K_SAMPLE_MULTI);
    v.sample_index = sample_index;
    v.floww_index = floww_index;
    v.has_note = !(note < 0);   // state.rs:358-359
    v.note = v.has_note ? (size_t)note : 0;
    return 1;
}
int td_graph_add_sample_lerp(td_graph* g, const char* name, float gain, float angle, size_t sample_index,
                             size_t floww_index, int note, int lerp_len) {
    Vertex& v = add_vertex(g, name, gain, angle, 0.0f, K_SAMPLE_LERP);
    v.sample_index = sample_index;
    v.floww_index = floww_index;
    v.has_note = !(note < 0);   // state.rs:368-369
    v.note = v.has_note ? (size_t)note : 0;
    v.lerp_len = (size_t)std::max(lerp_len, 0);   // state.rs:370
    return 1;
}
int td_graph_add_debug_sine(td_graph* g, const char* name, float gain, float angle, size_t floww_index) {
    Vertex& v = add_vertex(g, name, gain, angle, 0.0f, K_DEBUG_SINE);
    v.floww_index = floww_index;
    return 1;
}
int td_graph_add_synth(td_graph* g, const char* name, float gain, float angle, size_t floww_index, float square_vel,
                       float square_z, const float* square_adsr, int square_adsr_len, float topflat_vel,
                       float topflat_z, const float* topflat_adsr, int topflat_adsr_len, float triangle_vel,
                       const float* triangle_adsr, int triangle_adsr_len) {
    AdsrConfD sq, tf, tr;
    if (!conf_from(square_adsr, square_adsr_len, &sq) || !conf_from(topflat_adsr, topflat_adsr_len, &tf) ||
        !conf_from(triangle_adsr, triangle_adsr_len, &tr))
        return fail("ADSR config must have 6 or 9 elements");   // state.rs:393 panics
    Vertex& v = add_vertex(g, name, gain, angle, 0.0f, K_SYNTH);
    v.floww_index = floww_index;
    v.square = {square_vel, fmaxf(square_z, 0.0001f), sq};   // state.rs:400
    v.topflat = {topflat_vel, topflat_z, tf};
    v.triangle = {triangle_vel, 0.0f, tr};
    return 1;
}
// Wavetable resource of add_sampsyn: this engine's own format (the reference parses with the un-vendored
// sampsyn crate, state.rs:415-422):  "TDWT" u32 version=1, u32 n_frames, u32 frame_len, f32 table_seconds,
// then n_frames*frame_len f32 little-endian.  Anything else -> the default table (one sine cycle of 2048),
// like the reference's "using default table!" arm.
static bool parse_wavetable(const uint8_t* b, size_t n, uint32_t* nf, uint32_t* fl, float* secs, std::vector<float>* data) {
    if (!b || n < 20 || memcmp(b, "TDWT", 4) != 0) return false;
    uint32_t ver;
    memcpy(&ver, b + 4, 4); memcpy(nf, b + 8, 4); memcpy(fl, b + 12, 4); memcpy(secs, b + 16, 4);
    if (ver != 1 || *nf == 0 || *fl < 2 || (uint64_t)*nf * *fl > (1u << 26) || n < 20 + (size_t)*nf * *fl * 4) return false;
    if (!(*secs > 0.0f)) return false;
    data->resize((size_t)*nf * *fl);
    memcpy(data->data(), b + 20, data->size() * 4);
    return true;
}
int td_graph_add_sampsyn(td_graph* g, const char* name, float gain, float angle, size_t floww_index, const float* adsr,
                         int adsr_len, const void* table_bytes, size_t table_len) {
    AdsrConfD c;
    if (!conf_from(adsr, adsr_len, &c)) return fail("ADSR config must have 6 or 9 elements");   // state.rs:410
    uint32_t nf = 1, fl = 2048;
    float secs = 1.0f;
    std::vector<float> data;
    if (!parse_wavetable((const uint8_t*)table_bytes, table_len, &nf, &fl, &secs, &data)) {
        nf = 1; fl = 2048; secs = 1.0f;
        data.resize(2048);
        for (int i = 0; i < 2048; ++i) data[i] = (float)sin(2.0 * 3.14159265358979323846 * (double)i / 2048.0);
    }
    if (!ensure_device(g->device)) return 0;
    // device layout: one 16-byte quad per (frame, index) holding the four samples the oscillator's two lerps read
    // (tdk::WaveTableD): one gather per voice-frame instead of four
    std::vector<float> quads(data.size() * 4);
    for (uint32_t f = 0; f < nf; ++f) {
        const uint32_t f1 = f + 1u < nf ? f + 1u : nf - 1u;
        for (uint32_t i = 0; i < fl; ++i) {
            const uint32_t i1 = i + 1u == fl ? 0u : i + 1u;
            float* q = &quads[((size_t)f * fl + i) * 4];
            q[0] = data[(size_t)f * fl + i];  q[1] = data[(size_t)f * fl + i1];
            q[2] = data[(size_t)f1 * fl + i]; q[3] = data[(size_t)f1 * fl + i1];
        }
    }
    float* d_t = nullptr;
    TD_HIP(hipMalloc(&d_t, quads.size() * sizeof(float)));
    TD_HIP(hipMemcpy(d_t, quads.data(), quads.size() * sizeof(float), hipMemcpyHostToDevice));
    g->wavetables.push_back(d_t);
    g->device_bytes += quads.size() * sizeof(float);
    Vertex& v = add_vertex(g, name, gain, angle, 0.0f, K_SAMPSYN);
    v.floww_index = floww_index;
    v.conf = c;
    v.wavetable = WaveTableD{(const float4*)d_t, nf, fl, secs, 0u};
    return 1;
}
int td_graph_add_adsr(td_graph* g, const char* name, float gain, float angle, float wet, size_t floww_index,
                      int use_off, int use_max, int note, const float* adsr, int adsr_len) {
    AdsrConfD c;
    if (!conf_from(adsr, adsr_len, &c)) return fail("ADSR config must have 6 or 9 elements");   // state.rs:444
    Vertex& v = add_vertex(g, name, gain, angle, wet, K_ADSR);
    v.floww_index = floww_index;
    v.use_off = use_off != 0;
    v.use_max = use_max != 0;
    v.has_note = !(note < 0);   // state.rs:439-440
    v.note = v.has_note ? (size_t)note : 0;
    v.conf = c;
    return 1;
}
static float band_gamma(float hz, size_t sampling_hz) {   // extensions.rs:176-183
    const float co = fmaxf(fminf(hz, 20000.0f), 0.0f);
    return 1.0f - powf(2.71828182845904523536f, -2.0f * 3.14159274101257324f * co / (float)sampling_hz);
}
int td_graph_add_bandpass(td_graph* g, const char* name, float gain, float angle, float wet, float cut_off_hz_low,
                          float cut_off_hz_high, int pass) {
    const int slot = new_slot(g);
    Vertex& v = add_vertex(g, name, gain, angle, wet, K_BAND_PASS);
    v.lgamma = band_gamma(cut_off_hz_low, g->sr);
    v.hgamma = band_gamma(cut_off_hz_high, g->sr);
    v.pass = pass != 0;
    v.state_slot = slot;
    g->hstate[slot].band = {0.f, 0.f, 0.f, 0.f, 1u, {0, 0, 0}};
    return 1;
}

static bool has_loop(size_t x, size_t b, const std::vector<std::vector<size_t>>& edges) {   // graph.rs:66-72
    if (x == b) return true;
    for (size_t y : edges[x])
        if (has_loop(y, b, edges)) return true;
    return false;
}
int td_graph_connect(td_graph* g, const char* a, const char* b) {   // graph.rs:58-96
    auto ia = g->name_map.find(a), ib = g->name_map.find(b);
    if (ia == g->name_map.end()) return fail(std::string("TermDaw: warning: vertex \"") + a + "\" cannot be found and thus can't be connected.");
    if (ib == g->name_map.end()) return fail(std::string("TermDaw: warning: vertex \"") + b + "\" cannot be found and thus can't be connected to.");
    const size_t ai = ia->second, bi = ib->second;
    if (ai == bi) return fail("connect: self edge");
    if (!g->vertices[bi].has_input()) return fail("connect: target vertex takes no input");
    if (has_loop(ai, bi, g->edges)) return fail("connect: edge would close a loop");
    g->edges[bi].push_back(ai);
    g->plan_dirty = true;
    return 1;
}
int td_graph_set_output(td_graph* g, const char* vertex) {
    auto it = g->name_map.find(vertex);
    if (it == g->name_map.end()) return fail("set_output: vertex not found");
    g->output_vertex = (long)it->second;
    g->plan_dirty = true;
    return 1;
}
int td_graph_check(const td_graph* g) {   // graph.rs:150-174
    if (g->output_vertex < 0) return fail("TermDaw: error: output vertex not found.");
    const size_t out = (size_t)g->output_vertex;
    if (g->edges[out].empty() && g->vertices[out].has_input()) return fail("TermDaw: error: output receives no inputs.");
    return 1;
}
void td_graph_set_time(td_graph* g, size_t time) { graph_set_time_impl(g, time); }
size_t td_graph_change_time(td_graph* g, size_t delta, int plus) {   // graph.rs:130-135
    const size_t nt = plus ? g->t + delta : g->t - std::min(delta, g->t);
    graph_set_time_impl(g, nt);
    return nt;
}
size_t td_graph_get_time(const td_graph* g) { return g->t; }
void td_graph_reset_normalize_vertices(td_graph* g) {   // extensions.rs:295-299
    // No device traffic: the value is handed to the next render as the initial max (SumDesc::init_max).
    for (auto& vx : g->vertices) {
        if (vx.kind != K_NORMALIZE) continue;
        vx.has_init_override = true;
        vx.init_override = 0.000001f;
        vx.peak_known = false;
    }
}
float td_graph_get_normalization_value(const td_graph* gc, const char* name) {
    td_graph* g = const_cast<td_graph*>(gc);
    auto it = g->name_map.find(name);
    if (it == g->name_map.end()) return -1.0f;
    const Vertex& v = g->vertices[it->second];
    if (v.kind != K_NORMALIZE) return -1.0f;
    if (v.has_init_override) return v.init_override;
    pull_state(g);
    return g->hstate[v.state_slot].norm.max;
}
size_t td_graph_vertex_count(const td_graph* g) { return g->vertices.size(); }

int td_graph_render_block(td_graph* g, const td_samplebank* sb, td_flowwbank* fb, float* l, float* r) {
    if (g->output_vertex < 0) return 0;   // None
    // Graph::render leaves the FlowwBank alone: run one block on a cursor snapshot
    const size_t frame = fb->frame;
    const std::vector<size_t> starts = fb->start_indices;
    const int ok = graph_render_chunks(g, sb, fb, 1, false, 16, true, 0, false);
    fb->frame = frame;
    fb->start_indices = starts;
    g->guard.post = 2;   // (a guarded block that is done again leaves the cursor where this call found it, too)
    if (!ok) return -1;
    std::vector<float2> tmp(g->bl);
    if (!drain(g)) return -1;
    if (hipMemcpyAsync(tmp.data(), g->last_out_f32, g->bl * sizeof(float2), hipMemcpyDeviceToHost, g->stream) != hipSuccess ||
        hipStreamSynchronize(g->stream) != hipSuccess) {
        fail("HIP error: block read-back failed");
        return -1;
    }
    for (size_t i = 0; i < g->bl; ++i) {
        if (l) l[i] = tmp[i].x;
        if (r) r[i] = tmp[i].y;
    }
    return 1;
}

// Graph::true_normalize_scan (graph.rs:222-237) around the dry run itself
static int scan_begin(td_graph* g, td_flowwbank* fb) {
    if (!ensure_graph_device(g)) return 0;
    if (g->plan_dirty) build_plan(g);
    if (!ensure_state_slots(g)) return 0;
    for (auto& v : g->vertices)   // reset_scan_normalization
        if (v.kind == K_NORMALIZE)
            TD_HIP(hipMemsetD32Async((hipDeviceptr_t)&g->dstate[v.state_slot].norm.scan_max, 0, 1, g->stream));
    fb->set_time(0);
    return 1;
}
static int scan_end(td_graph* g, td_flowwbank* fb) {
    for (auto& v : g->vertices)   // apply_scan_normalization: max = scan_max (every Normalize vertex, reached or not)
        if (v.kind == K_NORMALIZE) {
            v.has_init_override = false;
            v.peak_known = true;
            TD_HIP(hipMemcpyAsync(&g->dstate[v.state_slot].norm.max, &g->dstate[v.state_slot].norm.scan_max, 4,
                                  hipMemcpyDeviceToDevice, g->stream));
        }
    g->state_dev_dirty = true;
    if (!graph_set_time_impl(g, 0)) return 0;
    fb->set_time(0);
    return 1;
}
int td_graph_normalize_scan(td_graph* g, const td_samplebank* sb, td_flowwbank* fb, size_t chunks) {   // graph.rs:222-237
    if (g->output_vertex < 0) return 1;
    if (!scan_begin(g, fb)) return 0;
    if (!graph_render_chunks(g, sb, fb, chunks, true, 16, false, 0, false)) return 0;
    if (g->guard.armed && !drain(g)) return 0;   // (band_mode 2: a dry run over the bound is done again BEFORE its peaks are applied)
    if (!scan_end(g, fb)) return 0;
    return drain(g);
}

size_t td_graph_render_all_async(td_graph* g, const td_samplebank* sb, td_flowwbank* fb, size_t n_blocks, int bits) {
    if (!graph_render_chunks(g, sb, fb, n_blocks, false, bits, true, 0, true)) return 0;
    if (!graph_set_time_impl(g, 0)) return 0;   // state.rs:575
    g->guard.post = 1;
    return n_blocks * g->bl;
}
int td_graph_sync(td_graph* g) { return drain(g); }
size_t td_graph_norm_fix_runs(const td_graph* g) { return g->arena.fix_runs + (g->batch ? g->batch->arena.fix_runs : 0); }
size_t td_graph_render_all(td_graph* g, const td_samplebank* sb, td_flowwbank* fb, size_t n_blocks, int bits) {
    const size_t n = td_graph_render_all_async(g, sb, fb, n_blocks, bits);
    if (!n) return 0;
    if (!td_graph_sync(g)) return 0;
    return n;
}
// State::render's `psr > render_sr` arm (state.rs:533-561): render, then resample the whole timeline with
// the build-defined resampler (the reference streams rubato block by block -- parity unpinned), quantise.
size_t td_graph_render_all_resampled(td_graph* g, const td_samplebank* sb, td_flowwbank* fb, size_t n_blocks, int bits,
                                     size_t psr, size_t render_sr) {
    if (!(bits == 8 || bits == 16 || bits == 24 || bits == 32)) {
        fail("Bitdepth not supported: choose bitdepth in {8, 16, 24, 32}.");
        return 0;
    }
    if (!graph_render_chunks(g, sb, fb, n_blocks, false, bits, true, 0, false)) return 0;
    if (!graph_set_time_impl(g, 0)) return 0;
    g->guard.post = 1;
    if (!drain(g)) return 0;   // (the resampler reads the output vertex' frames)
    const size_t total = n_blocks * g->bl;
    float2* rs = nullptr;
    size_t nout = 0;
    if (!resample_device(g->last_out_f32, total, psr, render_sr, &rs, &nout, g->stream)) return 0;
    if (g->d_resampled) (void)hipFree(g->d_resampled);
    g->d_resampled = rs;
    const int qmode = bits > 16 ? 2 : 1;
    const size_t word = qmode == 1 ? 2 : 4;
    const float amplitude = bits < 32 ? (float)((1 << (bits - 1)) - 1) : (float)INT32_MAX;
    const size_t need = nout * 2 * word + 64;
    if (need > g->pcm_cap) {
        if (hipStreamSynchronize(g->stream) != hipSuccess) return 0;
        if (g->d_pcm && !g->pcm_borrowed) { (void)hipFree(g->d_pcm); g->device_bytes -= g->pcm_cap; }
        g->d_pcm = nullptr;
        g->pcm_cap = 0;
        g->pcm_borrowed = false;
        if (hipMalloc(&g->d_pcm, need) != hipSuccess) { fail("out of device memory"); return 0; }
        g->pcm_cap = need;
        g->device_bytes += need;
    }
    QuantDesc qd{rs, g->d_pcm, amplitude, (uint32_t)qmode};
    if (hipMemcpyAsync(g->d_scalar + 8, &qd, sizeof qd, hipMemcpyHostToDevice, g->stream) != hipSuccess) return 0;
    if (hipStreamSynchronize(g->stream) != hipSuccess) return 0;   // qd lives on this stack frame
    launch_quantise((const QuantDesc*)(g->d_scalar + 8), 1, (uint32_t)nout, g->stream);
    if (hipStreamSynchronize(g->stream) != hipSuccess) return 0;
    g->pcm_bytes = nout * 2 * word;
    g->last_out_f32 = rs;
    g->last_frames = nout;
    g->last_bits = bits;
    return nout;
}
const void* td_graph_output_pcm_device(const td_graph* g) { return g->d_pcm; }
const float* td_graph_output_f32_device(const td_graph* g) { return (const float*)g->last_out_f32; }
int td_graph_read_pcm(const td_graph* g, void* out, size_t bytes) {
    if (!g->d_pcm || bytes > g->pcm_bytes) return fail("read_pcm: nothing rendered / size too large");
    if (!drain(const_cast<td_graph*>(g))) return 0;
    TD_HIP(hipMemcpy(out, g->d_pcm, bytes, hipMemcpyDeviceToHost));
    return 1;
}
int td_graph_read_f32(const td_graph* g, float* out, size_t n_floats) {
    if (!g->last_out_f32 || n_floats > g->last_frames * 2) return fail("read_f32: nothing rendered / size too large");
    if (!drain(const_cast<td_graph*>(g))) return 0;
    TD_HIP(hipMemcpy(out, g->last_out_f32, n_floats * sizeof(float), hipMemcpyDeviceToHost));
    return 1;
}
float td_graph_output_peak(const td_graph* gc) {
    td_graph* g = const_cast<td_graph*>(gc);
    if (!g->last_out_f32 || !g->last_frames) return 0.0f;
    if (!drain(g)) return 0.0f;
    launch_absmax((const float*)g->last_out_f32, (uint32_t)std::min<size_t>(g->last_frames * 2, 0xFFFFFFFFu), g->d_scalar,
                  g->stream);
    float v = 0.0f;
    if (hipMemcpyAsync(&v, g->d_scalar, 4, hipMemcpyDeviceToHost, g->stream) != hipSuccess) return 0.0f;
    (void)hipStreamSynchronize(g->stream);
    return v;
}
size_t td_graph_host_times(td_graph* g, double* ms4, int reset) {
    for (int i = 0; i < 4; ++i) ms4[i] = g->host_ms[i];
    const size_t n = g->host_chunks;
    if (reset) { for (double& v : g->host_ms) v = 0.0; g->host_chunks = 0; }
    return n;
}
static void prof_set(ProfCtx& pc, int on) {
    pc.every = on > 0 ? (unsigned)on : 0u;
    pc.count = 0;
    pc.now = false;
    for (auto& e : pc.pending) { pc.free_ev.push_back(e.a); pc.free_ev.push_back(e.b); }
    pc.pending.clear();
    pc.last_times.clear();
}
// (the caller has synchronised the stream the events were recorded on)
static size_t prof_collect(ProfCtx& pc, const char** names, float* ms, size_t* launches, size_t cap) {
    if (pc.last_times.empty()) {
        pc.last_times.resize(F_COUNT);
        for (int f = 0; f < F_COUNT; ++f) pc.last_times[f].name = kFamilyName[f];
    }
    for (auto& e : pc.pending) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, e.a, e.b) == hipSuccess) {
            pc.last_times[e.fam].ms += t;
            pc.last_times[e.fam].launches += 1;
        }
        pc.free_ev.push_back(e.a);
        pc.free_ev.push_back(e.b);
    }
    pc.pending.clear();
    size_t n = 0;
    for (auto& kt : pc.last_times) {
        if (!kt.launches) continue;
        if (n < cap) {
            names[n] = kt.name.c_str();
            ms[n] = kt.ms;
            launches[n] = kt.launches;
        }
        ++n;
    }
    return std::min(n, cap);
}
void td_graph_set_profiling(td_graph* g, int on) { prof_set(g->prof, on); }
size_t td_graph_last_kernel_times(const td_graph* gc, const char** names, float* ms, size_t* launches, size_t cap) {
    td_graph* g = const_cast<td_graph*>(gc);
    if (!(g->stream && hipSetDevice(g->device) == hipSuccess)) return 0;
    (void)hipStreamSynchronize(g->stream);
    return prof_collect(g->prof, names, ms, launches, cap);
}
size_t td_graph_device_bytes(const td_graph* g) { return g->device_bytes + g->arena.device_bytes; }

int td_graph_band_guard_stats(const td_graph* g, double out[4]) {
    out[0] = (double)g->guard.audits;
    out[1] = (double)g->guard.redos;
    out[2] = (double)g->guard.last_est;
    out[3] = (double)g->guard.max_est;
    return 1;
}
int td_graph_band_stats(const td_graph* gc, uint32_t out[3]) {
    td_graph* g = const_cast<td_graph*>(gc);
    out[0] = out[1] = out[2] = 0;
    if (g->band_stats_off.empty() || !g->band_stats_base) return 1;
    if (!ensure_device(g->device)) return 0;
    TD_HIP(hipStreamSynchronize(g->stream));
    for (size_t so : g->band_stats_off) {
        uint32_t s[4];
        TD_HIP(hipMemcpy(s, g->band_stats_base + so, 16, hipMemcpyDeviceToHost));
        for (int i = 0; i < 3; ++i) out[i] += s[i];
    }
    return 1;
}

int td_graph_set_option(td_graph* g, const char* key, long value) {
    const std::string k = key ? key : "";
    if (k == "fuse_sources") { g->fuse_sources = value != 0; return 1; }
    if (k == "band_parallel") { g->band_parallel = value != 0; return 1; }
    if (k == "band_mode") {   // 0: exact (default, the parity mode), 1: blocked affine scan (tolerance class), 2: the scan under the guard
        if (value != 0 && value != 1 && value != 2) return fail("band_mode must be 0 (exact), 1 (scan) or 2 (guarded scan)");
        if (g->guard.armed && !drain(g)) return 0;   // (a verdict still out belongs to the mode it was rendered in)
        g->band_mode = (int)value;
        return 1;
    }
    if (k == "band_guard_ppb") { g->band_guard_ppb = value > 0 ? (unsigned)std::min<long>(value, 1000000000L) : 0u; return 1; }
    if (k == "band_scan_nf") {
        if (value != 8 && value != 16) return fail("band_scan_nf must be 8 or 16");
        g->band_scan_nf = (int)value;
        return 1;
    }
    if (k == "band_scan_debug") { g->band_scan_debug = (int)value; return 1; }
    if (k == "band_chain") { g->band_chain = value != 0; return 1; }   // scan mode: chains of band-pass vertices in one launch
    if (k == "band_live_exp") { g->band_live_thr = value >= 38 ? 0.0f : powf(10.0f, -(float)value); return 1; }
    if (k == "band_short") { g->band_short = value > 0 ? (unsigned)value : 64u; return 1; }
    if (k == "band_quick") { g->band_quick = value > 0 ? (unsigned)value : 0u; return 1; }
    if (k == "band_medium") { g->band_medium = value > 0 ? (unsigned)value : 30u; return 1; }
    if (k == "band_guess_min") { g->band_guess_min = value > 0 ? (unsigned)value : 0u; return 1; }
    if (k == "band_depth") { g->band_depth = value > 0 ? (unsigned)value : 100u; return 1; }
    if (k == "band_scan_depth") { g->band_scan_depth = value > 0 ? (unsigned)value : 64u; return 1; }
    if (k == "band_warmup") { g->band_warmup = value > 0 ? (unsigned)value : 150u; return 1; }
    if (k == "packed_samples") { g->packed_samples = value != 0; return 1; }
    if (k == "inline_adsr") { g->inline_adsr = value != 0; return 1; }
    if (k == "spec_normalize") { g->spec_normalize = value != 0; return 1; }
    if (k == "single_pass_normalize") { g->single_pass_normalize = value != 0; return 1; }
    if (k == "norm_debug") { g->norm_debug = (int)value; return 1; }
    if (k == "fuse_normalize") { g->fuse_normalize = value != 0; return 1; }
    if (k == "one_grid_sources") { g->one_grid_sources = value != 0; return 1; }
    if (k == "output_f32") { g->output_f32 = value != 0; return 1; }
    if (k == "table_cache") { g->table_cache = value != 0; return 1; }
    if (k == "graph_replay") { g->graph_replay = value != 0; return 1; }
    if (k == "branch_streams") { g->branch_streams = value != 0; return 1; }
    if (k == "max_chunk_frames") {
        if (value < 1) return fail("max_chunk_frames must be >= 1");
        g->max_chunk_frames = (size_t)value;
        return 1;
    }
    return fail("unknown option \"" + k + "\"");
}

// ---- Batch (no reference counterpart: the reference renders one project per process; this is the loop a
// batch driver would run State::render, state.rs:563-575, in for many independent States) ----
td_batch* td_batch_new(void) {
    td_batch* b = new td_batch();
    b->device = t_device;
    return b;
}
void td_batch_free(td_batch* b) {
    if (!b) return;
    const bool dev_ok = hipSetDevice(b->device) == hipSuccess;
    if (b->stream && dev_ok) (void)settle_arena(b->arena, b->stream);   // (the projects' results stay readable through their own handles)
    for (td_graph* g : b->graphs) {   // the projects outlive the batch: give each its own stream back (made on next use)
        // (a member rendered on its own queues on the batch's stream with its own arena: its deferred check, if one is
        // outstanding, must run before that stream goes)
        if (b->stream && dev_ok) (void)settle_arena(g->arena, b->stream);
        drop_pending(g->arena);
        g->batch = nullptr;
        g->stream = nullptr;
        g->owns_stream = true;
        g->band_stats_base = nullptr;   // (it pointed into the batch arena's scratch, freed below)
        g->band_stats_off.clear();
        if (g->pcm_borrowed) { g->d_pcm = nullptr; g->pcm_cap = 0; g->pcm_bytes = 0; g->pcm_borrowed = false; }   // (a slice of the batch's PCM arena)
        if (dev_ok && hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) != hipSuccess) g->stream = nullptr;
    }
    if (dev_ok) {
        free_arena(b->arena);
        free_prof(b->prof);
        if (b->d_peaks) (void)hipFree(b->d_peaks);
        if (b->copy_stream) { (void)hipStreamSynchronize(b->copy_stream); (void)hipStreamDestroy(b->copy_stream); }
        if (b->d_pcm_arena) (void)hipFree(b->d_pcm_arena);
        for (hipEvent_t e : b->ev_pool) (void)hipEventDestroy(e);
        for (hipEvent_t e : b->ev_mark) if (e) (void)hipEventDestroy(e);
        if (b->host_pcm) (void)hipHostFree(b->host_pcm);
        if (b->stream) (void)hipStreamDestroy(b->stream);
    }
    drop_pending(b->arena);
    delete b;
}
long td_batch_add(td_batch* b, td_graph* g, const td_samplebank* sb, td_flowwbank* fb) {
    if (!g || !sb || !fb) { fail("td_batch_add: null handle"); return -1; }
    if (g->batch) { fail("td_batch_add: the graph already belongs to a batch"); return -1; }
    if (g->device != b->device || sb->device != b->device) { fail("td_batch_add: project and batch live on different devices"); return -1; }
    if (!ensure_device(b->device)) return -1;
    if (!b->stream && hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess) {
        fail("td_batch_add: cannot create a HIP stream");
        return -1;
    }
    // the graph's launches, state copies and memsets move to the batch's stream
    if (g->stream) {
        (void)drain(g);   // (its own arena's deferred check, if any, runs on the stream about to go)
        if (g->owns_stream) (void)hipStreamDestroy(g->stream);
    }
    g->stream = b->stream;
    g->owns_stream = false;
    g->batch = b;
    b->graphs.push_back(g);
    b->sbs.push_back(sb);
    b->fbs.push_back(fb);
    return (long)b->graphs.size() - 1;
}
size_t td_batch_size(const td_batch* b) { return b->graphs.size(); }
/* reset_normalize_vertices (state.rs:467) + FlowwBank::set_time(0) for every project: the state right after refresh */
void td_batch_rewind(td_batch* b) {
    for (size_t i = 0; i < b->graphs.size(); ++i) {
        td_graph_reset_normalize_vertices(b->graphs[i]);
        b->fbs[i]->set_time(0);
    }
}
size_t td_batch_render_all_async(td_batch* b, size_t n_blocks, int bits) {
    if (!batch_render_chunks(b, n_blocks, false, bits, true, true)) return 0;
    for (td_graph* g : b->graphs) {
        if (!graph_set_time_impl(g, 0)) return 0;   // state.rs:575
        g->guard.post = 1;
    }
    return b->graphs.empty() ? 0 : n_blocks * b->graphs[0]->bl;
}
int td_batch_sync(td_batch* b) {
    if (!b->stream) return 1;
    if (!ensure_device(b->device)) return 0;
    if (!settle_arena(b->arena, b->stream)) return 0;
    for (td_graph* g : b->graphs)   // (band_mode 2: a project whose estimate was over the bound renders again, alone, exact)
        if (g->guard.armed && !guard_settle(g)) return 0;
    return 1;
}
size_t td_batch_render_all(td_batch* b, size_t n_blocks, int bits) {
    const size_t n = td_batch_render_all_async(b, n_blocks, bits);
    if (!n) return 0;
    if (!td_batch_sync(b)) return 0;
    return n;
}

// State::render (state.rs:477-577) for every project of the batch, END TO END: render, PCM to the host, the WAV file.
// The projects render in groups of `group` (one submission each, queued back to back on the batch's stream) into ONE device
// arena, project after project; a copy stream takes each group's PCM -- one contiguous transfer -- to page-locked host memory
// as soon as the group has rendered, while the next groups render; `writers` host threads write project i's file -- hound's
// header (wav.cpp) + the PCM words -- as soon as its group's copy has landed.  Returns when every file is written.
// (Measured: one copy per project leaves ~38 us between transfers, 0.84 of the pinned copy rate; two copy streams make the
// runtime copy with shader kernels that take the CUs from the renders: 27 GB/s and renders twice as slow.)
int td_batch_render_to_files(td_batch* b, size_t n_blocks, int bits, size_t render_sr, const char* const* paths, int group,
                             int writers, double* times) {
    const size_t P = b->graphs.size();
    if (times) for (int i = 0; i < 8; ++i) times[i] = 0.0;
    if (!P) return 1;
    if (!ensure_device(b->device)) return 0;
    if (!(bits == 8 || bits == 16 || bits == 24 || bits == 32)) return fail("Bitdepth not supported: choose bitdepth in {8, 16, 24, 32}.");
    const size_t G = group > 0 ? (size_t)group : 8;
    const size_t n_groups = (P + G - 1) / G;
    const auto w0 = std::chrono::steady_clock::now();
    if (!b->copy_stream) TD_HIP(hipStreamCreateWithFlags(&b->copy_stream, hipStreamNonBlocking));
    // events: [g] render of group g done; a timed pair around every group's copy; two timed ones around the renders
    const size_t n_ev = 3 * n_groups + 2;
    while (b->ev_pool.size() < n_ev) {
        hipEvent_t e = nullptr;
        TD_HIP(hipEventCreate(&e));
        b->ev_pool.push_back(e);
    }
    hipEvent_t* ev_group = b->ev_pool.data();
    hipEvent_t* ev_c0 = ev_group + n_groups;
    hipEvent_t* ev_c1 = ev_c0 + n_groups;
    hipEvent_t ev_r0 = ev_c1[n_groups], ev_r1 = ev_c1[n_groups + 1];
    // one slice per project, the same layout on the device and in page-locked host memory (kept from call to call)
    const size_t word = bits > 16 ? 4 : 2;
    b->host_pcm_off.assign(P, 0);
    b->host_pcm_bytes.assign(P, 0);
    std::vector<size_t> slice(P);
    size_t need = 0;
    for (size_t i = 0; i < P; ++i) {
        b->host_pcm_off[i] = need;
        b->host_pcm_bytes[i] = n_blocks * b->graphs[i]->bl * 2 * word;
        slice[i] = (b->host_pcm_bytes[i] + 64 + 4095) & ~(size_t)4095;   // (+ 64: the engine's own pad behind a PCM buffer)
        need += slice[i];
    }
    if (need > b->host_pcm_cap) {
        if (b->host_pcm) (void)hipHostFree(b->host_pcm);
        b->host_pcm = nullptr;
        b->host_pcm_cap = 0;
        TD_HIP(hipHostMalloc((void**)&b->host_pcm, need, hipHostMallocDefault));
        b->host_pcm_cap = need;
    }
    bool relayout = need > b->d_pcm_arena_cap;
    for (size_t i = 0; i < P && !relayout; ++i)
        relayout = !(b->graphs[i]->pcm_borrowed && b->graphs[i]->d_pcm == b->d_pcm_arena + b->host_pcm_off[i] && b->graphs[i]->pcm_cap >= slice[i]);
    if (relayout) {
        if (!settle_arena(b->arena, b->stream)) return 0;   // (nothing queued may still write an old PCM buffer)
        if (need > b->d_pcm_arena_cap) {
            for (td_graph* g : b->graphs)
                if (g->pcm_borrowed) { g->d_pcm = nullptr; g->pcm_cap = 0; g->pcm_bytes = 0; g->pcm_borrowed = false; }
            if (b->d_pcm_arena) (void)hipFree(b->d_pcm_arena);
            b->d_pcm_arena = nullptr;
            b->d_pcm_arena_cap = 0;
            TD_HIP(hipMalloc((void**)&b->d_pcm_arena, need));
            b->d_pcm_arena_cap = need;
        }
        for (size_t i = 0; i < P; ++i) {
            td_graph* g = b->graphs[i];
            if (g->d_pcm && !g->pcm_borrowed) { (void)hipFree(g->d_pcm); g->device_bytes -= g->pcm_cap; }
            g->d_pcm = b->d_pcm_arena + b->host_pcm_off[i];
            g->pcm_cap = slice[i];
            g->pcm_bytes = 0;
            g->pcm_borrowed = true;
        }
    }
    const auto w1 = std::chrono::steady_clock::now();
    // writer threads: project i is theirs once its group's copy has completed
    std::atomic<size_t> next{0};
    std::atomic<int> failed{0};
    const size_t nw = paths ? (size_t)std::max(writers, 1) : 0;   // (files asked for: at least one writer, whatever `writers` says)
    std::vector<std::string> errs(nw);
    std::vector<double> first_write(nw, -1.0), last_write(nw, 0.0);
    std::atomic<size_t> queued{0};   // groups whose copy has been enqueued (their events are recorded)
    const int dev = b->device;
    // a file is written in `parts` slices by as many threads (pwrite at their own offsets): the last group's files -- nothing
    // renders or copies under them any more -- are then finished by all the writers, not by one thread per file
    const size_t parts = (bits == 16 || bits == 32) ? (size_t)std::max<long>(1, std::min<long>(8, (long)nw * 2 / (long)std::max<size_t>(G, 1))) : 1;
    auto writer = [&](size_t w) {
        (void)hipSetDevice(dev);
        for (;;) {
            const size_t task = next.fetch_add(1);
            if (task >= P * parts) return;
            const size_t i = task / parts, part = task % parts;
            const size_t gi = i / G;
            while (queued.load(std::memory_order_acquire) <= gi) {
                if (failed.load()) return;
                std::this_thread::yield();
            }
            if (hipEventSynchronize(ev_c1[gi]) != hipSuccess) { failed = 1; errs[w] = "copy event failed"; return; }
            const double t_a = ms_between(w0, std::chrono::steady_clock::now());
            if (first_write[w] < 0) first_write[w] = t_a;
            std::string err;
            const void* words = b->host_pcm + b->host_pcm_off[i];
            const size_t frames = n_blocks * b->graphs[i]->bl;
            const bool ok_w = parts > 1 ? tdw::write_wav_int_part(paths[i], words, frames, 2, render_sr, bits, (int)part, (int)parts, &err)
                                        : tdw::write_wav_int(paths[i], words, frames, 2, render_sr, bits, &err);
            if (!ok_w) {
                failed = 1;
                errs[w] = err;
                return;
            }
            last_write[w] = ms_between(w0, std::chrono::steady_clock::now());
        }
    };
    std::vector<std::thread> pool;
    for (size_t w = 0; w < nw; ++w) pool.emplace_back(writer, w);
    int ok = 1;
    if (hipEventRecord(ev_r0, b->stream) != hipSuccess) ok = fail("HIP error: event");
    for (size_t gi = 0; gi < n_groups && ok; ++gi) {
        const size_t lo = gi * G, hi = std::min(P, lo + G);
        ok = batch_render_range(b, lo, hi, n_blocks, false, bits, true, true, false);
        for (size_t i = lo; i < hi && ok; ++i) {
            ok = graph_set_time_impl(b->graphs[i], 0);   // state.rs:575
            b->graphs[i]->guard.post = 1;
        }
        {   // (band_mode 2: a group with guarded projects is settled -- verdicts looked at, a project over the bound done again -- before its PCM leaves)
            bool any_armed = false;
            for (size_t i = lo; i < hi; ++i) any_armed = any_armed || b->graphs[i]->guard.armed;
            if (ok && any_armed) {
                ok = settle_arena(b->arena, b->stream);
                for (size_t i = lo; i < hi && ok; ++i) ok = guard_settle(b->graphs[i]);
            }
        }
        if (!ok) break;
        const size_t bytes = b->host_pcm_off[hi - 1] + b->host_pcm_bytes[hi - 1] - b->host_pcm_off[lo];
        if (hipEventRecord(ev_group[gi], b->stream) != hipSuccess || hipStreamWaitEvent(b->copy_stream, ev_group[gi], 0) != hipSuccess ||
            hipEventRecord(ev_c0[gi], b->copy_stream) != hipSuccess ||
            hipMemcpyAsync(b->host_pcm + b->host_pcm_off[lo], b->d_pcm_arena + b->host_pcm_off[lo], bytes, hipMemcpyDeviceToHost, b->copy_stream) != hipSuccess ||
            hipEventRecord(ev_c1[gi], b->copy_stream) != hipSuccess) { ok = fail("HIP error: PCM copy to the host"); break; }
        queued.store(gi + 1, std::memory_order_release);
    }
    if (ok && hipEventRecord(ev_r1, b->stream) != hipSuccess) ok = fail("HIP error: event");
    if (!ok) failed = 1;
    const auto w2 = std::chrono::steady_clock::now();
    for (auto& t : pool) t.join();
    if (hipStreamSynchronize(b->copy_stream) != hipSuccess || hipStreamSynchronize(b->stream) != hipSuccess) ok = ok && fail("HIP error: stream");
    const auto w3 = std::chrono::steady_clock::now();
    if (ok && failed.load()) {
        std::string e = "td_batch_render_to_files: ";
        for (auto& x : errs) if (!x.empty()) { e += x; break; }
        return fail(e);
    }
    if (!ok) return 0;
    if (times) {
        float ms = 0.f;
        times[0] = ms_between(w0, w3);                         // wall: whole call
        times[1] = ms_between(w0, w1);                         // of which: buffers (first call only) + events
        if (hipEventElapsedTime(&ms, ev_r0, ev_r1) == hipSuccess) times[2] = ms;                    // GPU: first render start -> last render end
        if (hipEventElapsedTime(&ms, ev_c0[0], ev_c1[n_groups - 1]) == hipSuccess) times[3] = ms;   // copy stream: first copy start -> last copy end
        double busy = 0.0, bytes = 0.0;
        for (size_t gi = 0; gi < n_groups; ++gi)
            if (hipEventElapsedTime(&ms, ev_c0[gi], ev_c1[gi]) == hipSuccess) busy += ms;
        for (size_t i = 0; i < P; ++i) bytes += (double)b->host_pcm_bytes[i];
        times[4] = busy;                                       // sum of the copies' own durations
        times[5] = bytes;
        double fw = -1.0, lw = 0.0;
        for (size_t w = 0; w < nw; ++w) {
            if (first_write[w] >= 0 && (fw < 0 || first_write[w] < fw)) fw = first_write[w];
            lw = std::max(lw, last_write[w]);
        }
        times[6] = fw < 0 ? 0.0 : lw - fw;                     // host: first file opened -> last file closed
        times[7] = ms_between(w1, w2);                         // host: time to enqueue everything
    }
    return 1;
}
const void* td_batch_host_pcm(const td_batch* b, size_t i, size_t* bytes) {
    if (bytes) *bytes = 0;
    if (!b->host_pcm || i >= b->host_pcm_off.size()) return nullptr;
    if (bytes) *bytes = b->host_pcm_bytes[i];
    return b->host_pcm + b->host_pcm_off[i];
}
int td_batch_normalize_scan(td_batch* b, size_t chunks) {   // State::scan_exact (state.rs:473-475) for every project
    for (size_t i = 0; i < b->graphs.size(); ++i) {
        if (b->graphs[i]->output_vertex < 0) return fail("TermDaw: error: output vertex not found.");
        if (!scan_begin(b->graphs[i], b->fbs[i])) return 0;
    }
    if (!batch_render_chunks(b, chunks, true, 16, false, false)) return 0;
    {
        bool any_armed = false;
        for (td_graph* g : b->graphs) any_armed = any_armed || g->guard.armed;
        if (any_armed && !td_batch_sync(b)) return 0;   // (a dry run over the bound is done again before its peaks are applied)
    }
    for (size_t i = 0; i < b->graphs.size(); ++i)
        if (!scan_end(b->graphs[i], b->fbs[i])) return 0;
    return td_batch_sync(b);
}
// Per-project peak after the last render: the output Normalize vertex' running peak (`max`, extensions.rs:323 --
// the project's pre-normalisation peak), or the absolute peak of the output buffer when the output vertex is no
// Normalize.  Fills a table of n_total floats in DEVICE memory: project i of this batch goes to entry
// first + i * stride, every other entry is written as 0 -- ready for one all-reduce(max) across the ranks.
int td_batch_peak_table_device(td_batch* b, float* d_table, size_t n_total, size_t first, size_t stride) {
    const size_t P = b->graphs.size();
    if (stride == 0 || (P && first + (P - 1) * stride >= n_total)) return fail("td_batch_peak_table_device: table too small");
    if (n_total > 0xFFFFFFFFull) return fail("td_batch_peak_table_device: table too large");
    if (!ensure_device(b->device)) return 0;
    if (!b->stream) TD_HIP(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    // The table kernel goes out right behind whatever is queued -- no host wait in front of it.  The carried max of a single-pass
    // Normalize is final only once a deferred k_norm_fix has had its chance: with one outstanding the stream is drained
    // afterwards, and in the rare case that the check then redid a vertex the table is made again.
    auto enqueue_table = [&]() -> int {
        if (P > b->peaks_cap) {
            TD_HIP(hipStreamSynchronize(b->stream));
            if (b->d_peaks) (void)hipFree(b->d_peaks);
            b->d_peaks = nullptr;
            b->peaks_cap = 0;
            TD_HIP(hipMalloc(&b->d_peaks, (P + 10) * (sizeof(float) + sizeof(float*))));
            b->peaks_cap = P + 8;
            b->peak_src.clear();   // (the device copy of the pointer table went with the old allocation)
        }
        const float** d_src = reinterpret_cast<const float**>(b->d_peaks + ((b->peaks_cap + 1) & ~(size_t)1));
        std::vector<const float*> src(P);
        for (size_t i = 0; i < P; ++i) {
            td_graph* g = b->graphs[i];
            const Vertex* outv = g->output_vertex >= 0 ? &g->vertices[(size_t)g->output_vertex] : nullptr;
            if (outv && outv->kind == K_NORMALIZE && g->dstate && !outv->has_init_override) {
                src[i] = &g->dstate[outv->state_slot].norm.max;
            } else {
                src[i] = b->d_peaks + i;
                if (g->last_out_f32 && g->last_frames)
                    launch_absmax((const float*)g->last_out_f32, (uint32_t)std::min<size_t>(g->last_frames * 2, 0xFFFFFFFFu), b->d_peaks + i, b->stream);
                else
                    TD_HIP(hipMemsetAsync(b->d_peaks + i, 0, sizeof(float), b->stream));
            }
        }
        // the pointer table rarely changes (carried normalize states keep their addresses from render to render): the
        // device copy is reused, and the exchange then costs one small launch and no synchronisation
        if (src != b->peak_src) {
            b->peak_src = src;
            if (P) TD_HIP(hipMemcpyAsync(d_src, b->peak_src.data(), P * sizeof(float*), hipMemcpyHostToDevice, b->stream));
            TD_HIP(hipStreamSynchronize(b->stream));
        }
        if (n_total) launch_peak_table(d_src, d_table, (uint32_t)n_total, (uint32_t)P, (uint32_t)first, (uint32_t)stride, b->stream);
        TD_HIP(hipGetLastError());
        return 1;
    };
    const bool pending = !b->arena.pending_fix.empty();
    const size_t runs0 = b->arena.fix_runs;
    if (!enqueue_table()) return 0;
    if (pending) {
        if (!settle_arena(b->arena, b->stream)) return 0;
        if (b->arena.fix_runs != runs0 && !enqueue_table()) return 0;
    }
    return 1;
}
int td_batch_peaks(td_batch* b, float* out) {   // host copy of this batch's own entries, in td_batch_add order
    const size_t P = b->graphs.size();
    if (!P) return 1;
    if (!ensure_device(b->device)) return 0;
    float* d_tab = nullptr;
    TD_HIP(hipMalloc(&d_tab, P * sizeof(float)));
    int ok = td_batch_peak_table_device(b, d_tab, P, 0, 1);
    if (ok && (hipMemcpyAsync(out, d_tab, P * sizeof(float), hipMemcpyDeviceToHost, b->stream) != hipSuccess ||
               hipStreamSynchronize(b->stream) != hipSuccess))
        ok = fail("td_batch_peaks: read-back failed");
    (void)hipFree(d_tab);
    return ok;
}
void td_batch_set_profiling(td_batch* b, int on) { prof_set(b->prof, on); }
int td_batch_mark(td_batch* b, int which) {
    if (which < 0 || which > 1) return fail("td_batch_mark: which is 0 or 1");
    if (!ensure_device(b->device)) return 0;
    if (!b->stream) TD_HIP(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    if (!b->ev_mark[which]) TD_HIP(hipEventCreate(&b->ev_mark[which]));
    TD_HIP(hipEventRecord(b->ev_mark[which], b->stream));
    b->mark_set[which] = true;
    if (which == 0) b->mark_set[1] = false;
    return 1;
}
double td_batch_marked_ms(td_batch* b) {
    if (!(b->mark_set[0] && b->mark_set[1]) || hipSetDevice(b->device) != hipSuccess) return -1.0;
    float ms = -1.0f;
    if (hipEventSynchronize(b->ev_mark[1]) != hipSuccess || hipEventElapsedTime(&ms, b->ev_mark[0], b->ev_mark[1]) != hipSuccess) return -1.0;
    return (double)ms;
}
size_t td_batch_last_kernel_times(td_batch* b, const char** names, float* ms, size_t* launches, size_t cap) {
    if (!(b->stream && hipSetDevice(b->device) == hipSuccess)) return 0;
    (void)hipStreamSynchronize(b->stream);
    return prof_collect(b->prof, names, ms, launches, cap);
}
size_t td_batch_host_times(td_batch* b, double* ms4, int reset) {
    for (int i = 0; i < 4; ++i) ms4[i] = b->host_ms[i];
    const size_t n = b->host_steps;
    if (reset) { for (double& v : b->host_ms) v = 0.0; b->host_steps = 0; }
    return n;
}

}  // extern "C"
