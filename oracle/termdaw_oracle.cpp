// termdaw_oracle.cpp -- CPU restatement of termdaw's per-block vertex/graph render path.
//
// THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py may load it.  The shipped path (termdaw_amd/) never links, imports
// or calls anything in this directory.
//
// What it is: a block-serial, single-threaded, planar-f32 restatement of the reference algorithm,
// written from the reference source text (every function cites the file:line it follows; paths are
// relative to /root/reference/).  It deliberately keeps the reference's structure (one recursive
// memoised DFS per block, one bl-frame buffer per vertex, per-sample event pulls) so that it can
// double as the CPU baseline ("port") in bench.py.
//
// Parity status of this oracle:
//   * PINNED by the reference's own tests: apply_adsr / apply_ads / apply_r / apply_r_rt
//     (src/adsr.rs:120-204, 5 tests, 51 assertions, tol 1e-3) -- see tests/test_oracle_adsr.py.
//   * Everything else: the reference ships no golden vectors, fixtures or integration tests, and
//     it cannot be compiled here (no cargo/rustc, un-vendored crates).  Those parts are restated
//     from source text and cross-checked only by hand-derived known answers
//     and by an independent numpy twin (tests/np_twin.py).  Treat them as "parity unpinned vs a
//     reference executable".
//   * Arithmetic that lives in un-vendored crates cannot be restated: rubato 0.15.0 resampling
//     (sample.rs:150-175, state.rs:533-561) is REPLACED by a build-defined sinc resampler with rubato's
//     visible parameter set (see "Build-defined sinc resampler" below; parity unpinned by construction);
//     the sampsyn 0.1.4 wavetable oscillator + table format (extensions.rs:532-578, state.rs:415-422) are
//     REPLACED by build-defined ones (see "Build-defined wavetable voice"; parity unpinned by construction);
//     the floww 0.1.10 MIDI reader (floww.rs:40-48) and LV2 hosting (extensions.rs:580-590) fail loudly.
//
// Build: g++ -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -shared (see oracle/Makefile) --
// the optimisation level of a Rust release build, no fast-math, no FMA contraction.
// Rust `as` casts are emulated (saturating, truncating, NaN->0); f32::max/min -> fmaxf/fminf;
// f32::sin/cos/powf/floor/ceil -> glibc sinf/cosf/powf/floorf/ceilf (what Rust lowers to on
// x86_64-unknown-linux-gnu).

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <limits>
#include <map>
#include <string>
#include <vector>

namespace {

// ---------------------------------------------------------------------------------------------
// Rust cast emulation
// ---------------------------------------------------------------------------------------------
inline size_t f32_as_usize(float x) {  // `x as usize`
    if (!(x == x)) return 0;
    if (x <= 0.0f) return 0;
    if (x >= 18446744073709551616.0f) return std::numeric_limits<size_t>::max();
    return (size_t)x;
}
inline int16_t f32_as_i16(float x) {  // `x as i16`
    if (!(x == x)) return 0;
    if (x <= -32768.0f) return INT16_MIN;
    if (x >= 32767.0f) return INT16_MAX;
    return (int16_t)x;  // truncates toward zero
}
inline int32_t f32_as_i32(float x) {  // `x as i32`
    if (!(x == x)) return 0;
    if (x <= -2147483648.0f) return INT32_MIN;
    if (x >= 2147483648.0f) return INT32_MAX;
    return (int32_t)x;
}

const float PI_F32 = 3.14159274101257324f;           // core::f32::consts::PI
const float E_F32 = 2.71828182845904523536f;         // std::f32::consts::E
const float FRAC_1_SQRT_2_F32 = 0.707106781186547524400844362104849039f;

// ---------------------------------------------------------------------------------------------
// adsr.rs
// ---------------------------------------------------------------------------------------------
struct AdsrConf {  // adsr.rs:1-12
    float std_vel = 0, attack_sec = 0, attack_vel = 0, decay_sec = 0, decay_vel = 0,
          sustain_sec = 0, sustain_vel = 0, release_sec = 0, release_vel = 0;
    float max_vel() const {  // adsr.rs:32-38
        return fmaxf(fmaxf(fmaxf(fmaxf(std_vel, attack_vel), decay_vel), sustain_vel), release_vel);
    }
};

inline float lerp(float a, float b, float t) { return a + t * (b - a); }  // adsr.rs:41-44

float apply_ads_internal(const AdsrConf& c, float t) {  // adsr.rs:46-60
    if (t <= c.attack_sec) {
        return lerp(c.std_vel, c.attack_vel, t / c.attack_sec);
    } else if (t <= c.attack_sec + c.decay_sec) {
        return lerp(c.attack_vel, c.decay_vel, (t - c.attack_sec) / c.decay_sec);
    } else if (t <= c.attack_sec + c.decay_sec + c.sustain_sec) {
        return lerp(c.decay_vel, c.sustain_vel, (t - c.attack_sec - c.decay_sec) / c.sustain_sec);
    } else {
        return -1000.0f;
    }
}

float apply_ads(const AdsrConf& c, float t) {  // adsr.rs:62-69
    float res = apply_ads_internal(c, t);
    if (res <= -1.0f) return c.sustain_vel;
    return res;
}

float apply_r(const AdsrConf& c, float t, float old_val) {  // adsr.rs:71-73
    return lerp(old_val, c.release_vel, fminf(t / c.release_sec, 1.0f));
}

float apply_adsr(const AdsrConf& c, float t) {  // adsr.rs:75-86
    float res = apply_ads_internal(c, t);
    if (res <= -1.0f) {
        return lerp(c.sustain_vel, c.release_vel,
                    fminf((t - c.attack_sec - c.decay_sec - c.sustain_sec) / c.release_sec, 1.0f));
    }
    return res;
}

float apply_r_rt(const AdsrConf& c, float t, float rt) {  // adsr.rs:89-92
    float rv = apply_ads(c, rt);
    return apply_r(c, t, rv);
}

bool build_adsr_conf(const float* arr, int n, AdsrConf* out) {  // adsr.rs:94-114 (+ hit_conf 15-30)
    if (n == 0) {
        *out = AdsrConf();
        return true;
    } else if (n == 6) {
        AdsrConf c;
        c.std_vel = 0.0f;
        c.attack_sec = arr[0];
        c.attack_vel = 1.0f;
        c.decay_sec = arr[1];
        c.decay_vel = arr[2];
        c.sustain_sec = arr[3];
        c.sustain_vel = arr[4];
        c.release_sec = arr[5];
        c.release_vel = 0.0f;
        *out = c;
        return true;
    } else if (n == 9) {
        AdsrConf c;
        c.std_vel = arr[0];
        c.attack_sec = arr[1];
        c.attack_vel = arr[2];
        c.decay_sec = arr[3];
        c.decay_vel = arr[4];
        c.sustain_sec = arr[5];
        c.sustain_vel = arr[6];
        c.release_sec = arr[7];
        c.release_vel = arr[8];
        *out = c;
        return true;
    }
    return false;
}

// ---------------------------------------------------------------------------------------------
// synth.rs
// ---------------------------------------------------------------------------------------------
struct OscConf {  // synth.rs:5-9
    float volume = 0, param = 0;
    AdsrConf adsr;
};

inline float square_sine_sample(float t, float hz, float z) {  // synth.rs:21-24
    return fminf(fmaxf(sinf(t * hz * 2.0f * PI_F32), -z), z) * (1.0f / z);
}
inline float topflat_sine_sample(float t, float hz, float z) {  // synth.rs:26-29
    return (fminf(sinf(t * hz * 2.0f * PI_F32), z) + ((1.0f - z) / 2.0f)) * (2.0f / (1.0f + z));
}
inline float triangle_sample(float t, float hz) {  // synth.rs:31-34
    return 4.0f * fabsf((t * hz) - floorf((t * hz) + 0.5f)) - 1.0f;
}
inline float note_hz(float note) {  // extensions.rs:451,503
    return 440.0f * powf(2.0f, (note - 69.0f) / 12.0f);
}

// ---------------------------------------------------------------------------------------------
// sample.rs
// ---------------------------------------------------------------------------------------------
float absmaxlen(const std::vector<float>& s, size_t len) {  // sample.rs:12-14
    float max = 0.0f;
    size_t n = s.size() < len ? s.size() : len;
    for (size_t i = 0; i < n; ++i) {
        float a = fabsf(s[i]);
        if (a > max) max = a;
    }
    return max;
}
float absmax(const std::vector<float>& s) { return absmaxlen(s, SIZE_MAX); }  // sample.rs:8-10

float mean_energy(const std::vector<float>& s) {  // sample.rs:16-22
    if (s.empty()) return 0.0f;
    float sum = 0.0f;
    for (float v : s) sum += fabsf(v);
    return sum / (float)s.size();
}

enum LoadMethod { M_STEREO, M_LEFT, M_RIGHT, M_LOUDEST, M_NORM, M_MIX };  // sample.rs:196-197

LoadMethod load_method_from(const char* s) {  // sample.rs:199-210
    if (!strcmp(s, "left")) return M_LEFT;
    if (!strcmp(s, "right")) return M_RIGHT;
    if (!strcmp(s, "loudest")) return M_LOUDEST;
    if (!strcmp(s, "normalize-seperate")) return M_NORM;
    if (!strcmp(s, "mix-down")) return M_MIX;
    return M_STEREO;
}

struct Sample {  // sample.rs:24-28
    std::vector<float> l, r;
    Sample() {}
    explicit Sample(size_t bl) : l(bl, 0.0f), r(bl, 0.0f) {}  // sample.rs:31-36
    size_t len() const { return l.size(); }                    // sample.rs:79-81

    void zero() {  // sample.rs:87-95
        for (auto& v : l) v = 0.0f;
        for (auto& v : r) v = 0.0f;
    }
    void apply_angle(float angle, size_t len) {  // sample.rs:97-106
        if (fabsf(angle) < 0.001f) return;
        float angle_rad = angle * 0.5f * 0.01745329f;
        float l_amp = FRAC_1_SQRT_2_F32 * (cosf(angle_rad) + sinf(angle_rad));
        float r_amp = FRAC_1_SQRT_2_F32 * (cosf(angle_rad) - sinf(angle_rad));
        for (size_t i = 0; i < len; ++i) {
            l[i] *= l_amp;
            r[i] *= r_amp;
        }
    }
    void apply_gain(float gain, size_t len) {  // sample.rs:108-114
        if (fabsf(gain - 1.0f) < 0.001f) return;
        size_t n = len < this->len() ? len : this->len();
        for (size_t i = 0; i < n; ++i) {
            l[i] *= gain;
            r[i] *= gain;
        }
    }
    float scan_max(size_t len) const {  // sample.rs:116-118
        return fmaxf(absmaxlen(l, len), absmaxlen(r, len));
    }
    void scale(size_t len, float scalar) {  // sample.rs:120-123
        size_t nl = l.size() < len ? l.size() : len;
        size_t nr = r.size() < len ? r.size() : len;
        for (size_t i = 0; i < nl; ++i) l[i] *= scalar;
        for (size_t i = 0; i < nr; ++i) r[i] *= scalar;
    }
    void normalize(size_t len) {  // sample.rs:125-130
        len = len < this->len() ? len : this->len();
        float max = scan_max(len);
        float scalar = 1.0f / max;
        scale(len, scalar);
    }
    void normalize_seperate() {  // sample.rs:132-137
        float scalel = 1.0f / absmax(l);
        float scaler = 1.0f / absmax(r);
        for (auto& v : l) v *= scalel;
        for (auto& v : r) v *= scaler;
    }
    void mix_down() {  // sample.rs:139-147
        size_t n = l.size() < r.size() ? l.size() : r.size();  // zip stops at the shorter
        std::vector<float> mix(n);
        for (size_t i = 0; i < n; ++i) mix[i] = l[i] + r[i];
        float scale = 1.0f / absmax(mix);
        for (auto& v : mix) v *= scale;
        l = mix;
        r = mix;
    }
};

// Sample::from (sample.rs:38-77). Returns false + message on the reference's Err arms.
bool sample_from(std::vector<float> l, std::vector<float> r, LoadMethod m, Sample* out,
                 std::string* err) {
    switch (m) {
        case M_LEFT:
            if (l.empty()) { *err = "Sample::from: l has length 0."; return false; }
            out->l = l; out->r = l; return true;
        case M_RIGHT:
            if (r.empty()) { *err = "Sample::from: r has length 0."; return false; }
            out->l = r; out->r = r; return true;
        case M_LOUDEST: {
            float lm = mean_energy(l), rm = mean_energy(r);
            if (lm > rm) { out->l = l; out->r = l; } else { out->l = r; out->r = r; }
            return true;
        }
        default:
            if (l.size() != r.size()) {
                *err = "Sample::from: l and r do not have the same length.";
                return false;
            }
            if (l.empty()) { *err = "Sample::from: l and r have length 0."; return false; }
            out->l = l; out->r = r; return true;
    }
}

// ---------------------------------------------------------------------------------------------
// Build-defined sinc resampler -- stands in for rubato 0.15.0 (un-vendored), PARITY UNPINNED.
// ---------------------------------------------------------------------------------------------
// The reference resamples with rubato::SincFixedIn<f32> (sample.rs:150-175, state.rs:533-561); its
// parameter set is visible (sinc_len 256, f_cutoff 0.95, Linear interpolation, oversampling 256,
// BlackmanHarris2) but its arithmetic, delay and output length are not.  This engine defines its own
// resampler with those parameters (specification: DESIGN.md "Resampler"); oracle and HIP kernel implement
// the same specification and must agree bit for bit -- neither is claimed to match rubato.
//   out[j] = sum_{k=0..255} in[i0 - 255 + k] * c_k,   x = j * from / to = i0 + frac   (zero outside the input):
//   the filter centre sits sinc_len / 2 = 128 input frames behind x -- rubato's SincFixedIn delays its output by that
//   much because it only looks at frames it has been handed (state.rs:545-560 feeds it block by block, never flushes)
//   out[j] = (1 - a) y0 + a y1,  y_q = sum_k in[..] * T[p+q][k] (f32, taps in order) -- the two neighbouring phases are
//            convolved and the RESULTS interpolated, rubato's interp_lin order;  p = floor(frac * 256), a = frac * 256 - p
//   T[p][k] = (f32)( sinc(fc * d) * bh(u)^2 / norm ),  d = k - 127 - p/256,  u = (d + 128) / 256,
//             fc = 0.95 * min(1, to/from),  bh = 4-term Blackman-Harris,  norm = (sum of the windowed sinc over all
//             256 * 256 grid points) / 256 -- rubato's make_sincs normalisation;  table in f64, rounded once
//   len_out = ceil(len * to / from); zero delay.
const int kSincLen = 256, kSincOver = 256;
void build_sinc_table(size_t from, size_t to, std::vector<float>* T) {
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
void resample_planar(const std::vector<float>& l, const std::vector<float>& r, size_t from, size_t to,
                     std::vector<float>* ol, std::vector<float>* orr) {
    std::vector<float> T;
    build_sinc_table(from, to, &T);
    const size_t len = l.size();
    const size_t nout = (size_t)(((unsigned __int128)len * to + from - 1) / from);
    ol->assign(nout, 0.0f);
    orr->assign(nout, 0.0f);
    for (size_t j = 0; j < nout; ++j) {
        const unsigned __int128 num = (unsigned __int128)j * from;
        const int64_t i0 = (int64_t)(num / to);
        const uint64_t rem = (uint64_t)(num % to);
        const uint64_t ph = rem * (uint64_t)kSincOver;
        const size_t p = (size_t)(ph / to);
        const float a = (float)(ph % to) / (float)to;
        const float* t0 = &T[p * kSincLen];
        const float* t1 = &T[(p + 1) * kSincLen];
        // the two neighbouring phases' convolutions, then ONE interpolation of the results (rubato's interp_lin order:
        // (1 - a) y0 + a y1 -- from the crate's published source, from memory, unverified)
        float al0 = 0.0f, ar0 = 0.0f, al1 = 0.0f, ar1 = 0.0f;
        for (int k = 0; k < kSincLen; ++k) {
            const int64_t idx = i0 - (127 + 128) + k;   // delayed by sinc_len / 2 input frames (SincFixedIn's output delay)
            if (idx < 0 || idx >= (int64_t)len) continue;
            const float xl = l[(size_t)idx], xr = r[(size_t)idx];
            al0 += xl * t0[k];
            ar0 += xr * t0[k];
            al1 += xl * t1[k];
            ar1 += xr * t1[k];
        }
        (*ol)[j] = (1.0f - a) * al0 + a * al1;
        (*orr)[j] = (1.0f - a) * ar0 + a * ar1;
    }
}

struct SampleBank {  // sample.rs:187-194
    size_t sample_rate;
    std::vector<Sample> samples;
    std::map<std::string, size_t> names;
    size_t max_sr = 0, max_bd = 0;
    explicit SampleBank(size_t sr) : sample_rate(sr) {}

    // Body of SampleBank::add after the WAV reader produced `linear` (sample.rs:252-313).
    // `linear` holds the decoded stream: ints cast `as f32` (NOT scaled; sample.rs:269-273) or floats.
    bool add_decoded(const std::string& name, const std::vector<float>& linear, int channels,
                     size_t sr, size_t bd, LoadMethod method, std::string* err) {
        if (names.count(name)) {  // sample.rs:225-230
            *err = "SampleBank: there is already a sample with name \"" + name + "\" present.";
            return false;
        }
        if (method == M_STEREO && channels != 2) {  // sample.rs:240-245
            *err = "SampleBank: only 2 channel samples are supported for stereo samples.";
            return false;
        }
        if (method != M_STEREO && channels > 2) {  // sample.rs:246-251
            *err = "SampleBank: only 1,2 channel samples are supported for left or right samples.";
            return false;
        }
        if (sr > max_sr) max_sr = sr;  // sample.rs:254-255
        if (bd > max_bd) max_bd = bd;
        std::vector<float> l, r;
        if (channels == 1) {  // sample.rs:277-282
            if (method == M_LEFT) l = linear; else r = linear;
        } else {  // sample.rs:283-292
            size_t half_len = linear.size() / 2;
            for (size_t i = 0; i < half_len; ++i) {
                l.push_back(linear[i * 2]);
                r.push_back(linear[i * 2 + 1]);
            }
            if (linear.size() > half_len * 2) l.push_back(linear[linear.size() - 1]);
        }
        Sample sample;
        if (!sample_from(l, r, method, &sample, err)) return false;  // sample.rs:293-296
        if (method == M_NORM) sample.normalize_seperate();           // sample.rs:297-303
        else if (method == M_MIX) sample.mix_down();
        else sample.normalize(SIZE_MAX);
        if (sr != sample_rate) {  // sample.rs:305-310: Sample::resample -> build-defined resampler (parity unpinned)
            Sample rs;
            resample_planar(sample.l, sample.r, sr, sample_rate, &rs.l, &rs.r);
            sample = rs;
        }
        samples.push_back(sample);  // sample.rs:311-312
        names[name] = samples.size() - 1;
        return true;
    }
    long get_index(const std::string& n) const {  // sample.rs:338-340
        auto it = names.find(n);
        return it == names.end() ? -1 : (long)it->second;
    }
    const Sample& get_sample(size_t i) const { return samples[i]; }  // sample.rs:342-344
};

// Minimal RIFF/WAVE PCM reader standing in for hound::WavReader (sample.rs:231-274): integer
// samples are returned as their integer value cast to f32, float samples as-is.
bool read_wav(const char* path, std::vector<float>* linear, int* channels, size_t* sr, size_t* bd,
              std::string* err) {
    FILE* f = fopen(path, "rb");
    if (!f) { *err = std::string("SampleBank: could not open file \"") + path + "\"."; return false; }
    std::vector<uint8_t> buf;
    uint8_t tmp[65536];
    size_t n;
    while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    fclose(f);
    auto u16 = [&](size_t o) { return (uint32_t)buf[o] | ((uint32_t)buf[o + 1] << 8); };
    auto u32 = [&](size_t o) { return u16(o) | (u16(o + 2) << 16); };
    if (buf.size() < 12 || memcmp(&buf[0], "RIFF", 4) || memcmp(&buf[8], "WAVE", 4)) {
        *err = "SampleBank: not a RIFF/WAVE file."; return false;
    }
    size_t pos = 12;
    int fmt = 0, ch = 0, bits = 0; size_t rate = 0; bool have_fmt = false;
    while (pos + 8 <= buf.size()) {
        uint32_t sz = u32(pos + 4);
        if (!memcmp(&buf[pos], "fmt ", 4) && pos + 8 + 16 <= buf.size()) {
            fmt = u16(pos + 8); ch = u16(pos + 10); rate = u32(pos + 12); bits = u16(pos + 22);
            if (fmt == 0xFFFE && sz >= 26) fmt = u16(pos + 8 + 24);  // WAVE_FORMAT_EXTENSIBLE
            have_fmt = true;
        } else if (!memcmp(&buf[pos], "data", 4)) {
            if (!have_fmt) { *err = "SampleBank: data chunk before fmt."; return false; }
            size_t end = pos + 8 + sz; if (end > buf.size()) end = buf.size();
            size_t bps = bits / 8;
            for (size_t o = pos + 8; o + bps <= end; o += bps) {
                if (fmt == 3 && bits == 32) { float v; memcpy(&v, &buf[o], 4); linear->push_back(v); }
                else if (fmt == 1 && bits == 8) linear->push_back((float)((int)buf[o] - 128));
                else if (fmt == 1 && bits == 16) linear->push_back((float)(int16_t)u16(o));
                else if (fmt == 1 && bits == 24) {
                    int32_t v = (int32_t)(u16(o) | ((uint32_t)buf[o + 2] << 16));
                    if (v & 0x800000) v |= ~0xFFFFFF;
                    linear->push_back((float)v);
                } else if (fmt == 1 && bits == 32) linear->push_back((float)(int32_t)u32(o));
                else { *err = "SampleBank: unsupported WAV encoding."; return false; }
            }
            *channels = ch; *sr = rate; *bd = (size_t)bits;
            return true;
        }
        pos += 8 + sz + (sz & 1);
    }
    *err = "SampleBank: no data chunk."; return false;
}

// ---------------------------------------------------------------------------------------------
// floww.rs
// ---------------------------------------------------------------------------------------------
struct Event { float t, note, vel; };  // floww crate tuple (_, t, note, vel); field 0 unused here

struct FlowwBank {  // floww.rs:6-16
    size_t sr, bl, frame = 0, block_index = 0;
    std::vector<std::vector<Event>> flowws;
    std::vector<size_t> start_indices;
    std::map<std::string, size_t> names;
    FlowwBank(size_t sr_, size_t bl_) : sr(sr_), bl(bl_) {}

    size_t declare_floww(const std::string& name, const std::vector<Event>& fl) {  // floww.rs:32-38
        flowws.push_back(fl);
        start_indices.push_back(0);
        size_t index = flowws.size() - 1;
        names[name] = index;
        return index;
    }
    std::vector<size_t> stream_list;
    size_t declare_stream(const std::string& name) {  // floww.rs:50-53
        size_t index = declare_floww(name, {});
        stream_list.push_back(index);
        return index;
    }
    // floww.rs:55-57 hands FlowwPackets to the floww crate's `unpacket` (un-vendored); its visible effect on
    // the bank -- events appended to the named floww -- is restated here with the events given directly.
    long append_stream(const std::string& name, const std::vector<Event>& ev) {
        long i = get_index(name);
        if (i < 0) return -1;
        flowws[(size_t)i].insert(flowws[(size_t)i].end(), ev.begin(), ev.end());
        return (long)flowws[(size_t)i].size();
    }
    void trim_streams() {  // floww.rs:59-64: drain(..start_index); start_indices are NOT rewound here
        for (size_t index : stream_list) {
            size_t k = std::min(start_indices[index], flowws[index].size());
            flowws[index].erase(flowws[index].begin(), flowws[index].begin() + (long)k);
        }
    }
    long get_index(const std::string& n) const {  // floww.rs:66-68
        auto it = names.find(n);
        return it == names.end() ? -1 : (long)it->second;
    }
    size_t frame_of(const Event& e) const { return f32_as_usize(e.t * (float)sr); }  // floww.rs:75

    void set_start_indices_to_frame(size_t t_frame, bool do_skip) {  // floww.rs:70-81
        for (size_t i = 0; i < flowws.size(); ++i) {
            size_t skip = do_skip ? start_indices[i] : 0;
            for (size_t j = skip; j < flowws[i].size(); ++j) {
                if (frame_of(flowws[i][j]) >= t_frame) {
                    start_indices[i] = j;
                    break;
                }
            }
        }
    }
    void set_time(size_t t) {  // floww.rs:83-86
        set_start_indices_to_frame(t, false);
        frame = t;
    }
    void set_time_to_next_block() {  // floww.rs:88-91
        frame += bl;
        set_start_indices_to_frame(frame, true);
    }
    void start_block(size_t index) {  // floww.rs:93-96
        if (index >= flowws.size()) return;
        block_index = start_indices[index];
    }
    // floww.rs:99-121 -- Option<(note, vel)>
    bool get_block_drum(size_t index, size_t offset_frame, float* note, float* vel) {
        if (index >= flowws.size()) return false;
        for (;;) {
            if (block_index >= flowws[index].size()) return false;
            Event next_event = flowws[index][block_index];
            if (frame_of(next_event) < frame + offset_frame) {
                block_index += 1;
                continue;
            }
            if (frame_of(next_event) == frame + offset_frame) {
                block_index += 1;
                if (next_event.vel > 0.001f) {
                    *note = next_event.note;
                    *vel = next_event.vel;
                    return true;
                }
            } else {
                return false;
            }
        }
    }
    struct Simple { bool on; float note, vel; };
    // floww.rs:124-141 -- Vec<(on?, note, vel)>; heap-allocates per sample like the reference.
    std::vector<Simple> get_block_simple(size_t index, size_t offset_frame) {
        std::vector<Simple> res;
        if (index >= flowws.size()) return res;
        for (;;) {
            if (block_index >= flowws[index].size()) break;
            Event next_event = flowws[index][block_index];
            if (frame_of(next_event) == frame + offset_frame) {
                block_index += 1;
                bool on = next_event.vel > 0.001f;
                res.push_back({on, next_event.note, next_event.vel});
            } else {
                break;
            }
        }
        return res;
    }
};

// ---------------------------------------------------------------------------------------------
// extensions.rs
// ---------------------------------------------------------------------------------------------
enum Kind { K_SUM, K_NORMALIZE, K_SAMPLE_LOOP, K_SAMPLE_MULTI, K_SAMPLE_LERP, K_DEBUG_SINE, K_SYNTH,
            K_SAMPSYN, K_ADSR, K_BAND_PASS };

// ---------------------------------------------------------------------------------------------
// Build-defined wavetable voice -- stands in for the sampsyn 0.1.4 crate (un-vendored), PARITY UNPINNED.
// ---------------------------------------------------------------------------------------------
// The reference's SampSyn vertex (extensions.rs:532-578) calls sampsyn::wavetable_act_state(table, &mut
// state, hz, t, sr) on a table parsed by sampsyn::parse_wavetable_from_buffer (state.rs:415-422).  Neither
// the file format nor the oscillator are visible.  This engine defines both (DESIGN.md "Wavetable voice");
// the voice bookkeeping around it (note on/off, envelope clocks, retain rule) follows the reference text.
//   file:  "TDWT" u32 version=1, u32 n_frames, u32 frame_len, f32 table_seconds, n_frames*frame_len f32 (LE)
//   voice: ph = frac(t * hz); sample position ph * frame_len, frame position min(t / table_seconds, 1) *
//          (n_frames - 1); bilinear interpolation (within the frame with wrap-around, then across frames)
struct WaveTable {
    uint32_t n_frames = 1, frame_len = 2048;
    float table_seconds = 1.0f;
    std::vector<float> data;
};
WaveTable default_wavetable() {   // WaveTable::default() stand-in: one frame, one sine cycle
    WaveTable t;
    t.data.resize(2048);
    for (int i = 0; i < 2048; ++i) t.data[i] = (float)sin(2.0 * 3.14159265358979323846 * (double)i / 2048.0);
    return t;
}
bool parse_wavetable(const uint8_t* b, size_t n, WaveTable* out) {
    if (n < 20 || memcmp(b, "TDWT", 4) != 0) return false;
    uint32_t ver, nf, fl;
    float secs;
    memcpy(&ver, b + 4, 4); memcpy(&nf, b + 8, 4); memcpy(&fl, b + 12, 4); memcpy(&secs, b + 16, 4);
    if (ver != 1 || nf == 0 || fl < 2 || (uint64_t)nf * fl > (1u << 26) || n < 20 + (size_t)nf * fl * 4) return false;
    if (!(secs > 0.0f)) return false;
    out->n_frames = nf; out->frame_len = fl; out->table_seconds = secs;
    out->data.resize((size_t)nf * fl);
    memcpy(out->data.data(), b + 20, (size_t)nf * fl * 4);
    return true;
}
inline float wavetable_act(const WaveTable& w, float hz, float t) {
    float ph = t * hz;
    ph = ph - floorf(ph);
    const float pos = ph * (float)w.frame_len;
    uint32_t i0 = (uint32_t)f32_as_usize(pos);
    if (i0 >= w.frame_len) i0 = w.frame_len - 1;
    const float a = pos - (float)i0;
    const uint32_t i1 = i0 + 1 == w.frame_len ? 0 : i0 + 1;
    float fp = fminf(t / w.table_seconds, 1.0f) * (float)(w.n_frames - 1);
    if (!(fp >= 0.0f)) fp = 0.0f;
    uint32_t f0 = (uint32_t)f32_as_usize(fp);
    if (f0 >= w.n_frames) f0 = w.n_frames - 1;
    const float b = fp - (float)f0;
    const uint32_t f1 = f0 + 1 < w.n_frames ? f0 + 1 : w.n_frames - 1;
    const float* r0 = &w.data[(size_t)f0 * w.frame_len];
    const float* r1 = &w.data[(size_t)f1 * w.frame_len];
    const float s0 = lerp(r0[i0], r0[i1], a);
    const float s1 = lerp(r1[i0], r1[i1], a);
    return lerp(s0, s1, b);
}

// sampsyn's per-voice oscillator state (extensions.rs:542 pushes `initial_state(wave_table, 0.0)` with every voice, :569 hands
// `&mut state` to wavetable_act_state per sample).  Its content is the crate's; the build-defined oscillator's state is the
// phase the voice starts at -- created at note-on, carried with the voice, never changed: the oscillator is the closed form
// of a phase accumulator (ph = phase0 + t hz, t the time since note-on), so every frame stays independent.
struct WtState { float phase0 = 0.0f; };
inline WtState initial_state(const WaveTable&, float phase) { return WtState{phase}; }
struct SynthNote { float note, vel, env_t, rel_t; WtState st = WtState{}; };
// wavetable_act_state (extensions.rs:569): the voice's state is its starting phase; phase0 == 0.0 adds nothing (x + 0.0 == x
// for every x the oscillator sees but -0.0, whose fractional part is the same 0)
inline float wavetable_act_state(const WaveTable& w, WtState& st, float hz, float t) {
    return st.phase0 == 0.0f ? wavetable_act(w, hz, t) : wavetable_act(w, hz, t + st.phase0 / hz);
}
struct SineNote { float note, vel; };
struct Voice3 { float t, vel, rel; };  // Adsr primary/ghost: (t_off, vel, release_val)

struct VertexExt {  // extensions.rs:15-80
    Kind kind = K_SUM;
    // Normalize
    float max = 0.0f, scan_max = 0.0f;
    // SampleLoop / SampleMulti / SampleLerp
    size_t sample_index = 0, t = 0, floww_index = 0;
    bool has_note = false; size_t note = 0;
    std::deque<std::pair<int64_t, float>> ts;
    size_t lerp_len = 0, countdown = 0;
    std::pair<int64_t, float> primary{0, 0.0f}, ghost{0, 0.0f};
    // DebugSine / Synth
    std::vector<SineNote> sine_notes;
    std::vector<SynthNote> notes;
    OscConf square, topflat, triangle;
    // SampSyn (uses `notes` and `conf`)
    WaveTable wave_table;
    // Adsr
    bool use_off = false, use_max = false;
    AdsrConf conf;
    Voice3 aprimary{0, 0, 0}, aghost{0, 0, 0};
    // BandPass
    float lgamma = 0, hgamma = 0, lprevl = 0, lprevr = 0, hprevl = 0, hprevr = 0;
    bool first = true, pass = true;

    bool has_input() const {  // extensions.rs:266-281
        switch (kind) {
            case K_SUM: case K_NORMALIZE: case K_ADSR: case K_BAND_PASS: return true;
            default: return false;
        }
    }
    void set_time(size_t time) {  // extensions.rs:196-204
        switch (kind) {
            case K_SAMPLE_LOOP: t = time; break;
            case K_DEBUG_SINE: sine_notes.clear(); break;
            case K_SYNTH: notes.clear(); break;
            case K_BAND_PASS: first = true; break;
            default: break;
        }
    }
};

void sum_inputs(Sample& buf, size_t len, const std::vector<const Sample*>& res) {  // extensions.rs:310-319
    buf.zero();
    for (const Sample* r : res) {
        size_t l = r->len() < len ? r->len() : len;
        for (size_t i = 0; i < l; ++i) {
            buf.l[i] += r->l[i];
            buf.r[i] += r->r[i];
        }
    }
}

void normalize_gen(Sample& buf, size_t len, float* max, float* scan_max, bool is_scan) {  // :321-329
    float buf_max = buf.scan_max(len);
    if (is_scan) *scan_max = fmaxf(buf_max, *scan_max);
    else *max = fmaxf(buf_max, *max);
    buf.scale(len, 1.0f / *max);
}

void sample_loop_gen(Sample& buf, const SampleBank& sb, size_t len, size_t* t, size_t si) {  // :331-341
    const Sample& sample = sb.get_sample(si);
    size_t l = sample.len();
    for (size_t i = 0; i < len; ++i) {
        buf.l[i] = sample.l[(*t + i) % l];
        buf.r[i] = sample.r[(*t + i) % l];
    }
    *t += len;
}

void sample_multi_gen(Sample& buf, const SampleBank& sb, FlowwBank& fb, size_t len, VertexExt& e) {  // :344-381
    const Sample& sample = sb.get_sample(e.sample_index);
    fb.start_block(e.floww_index);
    for (size_t i = 0; i < len; ++i) {
        float note, v;
        if (fb.get_block_drum(e.floww_index, i, &note, &v)) {
            bool ok = e.has_note ? fabsf(note - (float)e.note) < 0.01f : true;
            if (ok) e.ts.push_back({-(int64_t)i, v});
        }
        buf.l[i] = 0.0f;
        buf.r[i] = 0.0f;
        size_t pops = 0;
        for (auto& tv : e.ts) {
            int64_t p = tv.first + (int64_t)i;
            size_t pos = (size_t)(p > 0 ? p : 0);
            if (pos >= sample.len()) {
                pops += 1;
            } else {
                buf.l[i] += sample.l[pos] * tv.second;
                buf.r[i] += sample.r[pos] * tv.second;
            }
        }
        for (size_t k = 0; k < pops; ++k) e.ts.pop_front();
    }
    for (auto& tv : e.ts) tv.first += (int64_t)len;
}

void sample_lerp_gen(Sample& buf, const SampleBank& sb, FlowwBank& fb, size_t len, VertexExt& e) {  // :384-421
    const Sample& sample = sb.get_sample(e.sample_index);
    fb.start_block(e.floww_index);
    for (size_t i = 0; i < len; ++i) {
        float note, v;
        if (fb.get_block_drum(e.floww_index, i, &note, &v)) {
            bool ok = e.has_note ? fabsf(note - (float)e.note) < 0.01f : true;
            if (ok) {
                e.ghost = e.primary;
                e.primary = {-(int64_t)i, v};
                e.countdown = e.lerp_len;
            }
        }
        auto clamp_pos = [&](int64_t off) {
            int64_t p = off + (int64_t)i;
            size_t pos = (size_t)(p > 0 ? p : 0);
            size_t last = sample.len() - 1;
            return pos < last ? pos : last;
        };
        size_t primary_pos = clamp_pos(e.primary.first);
        float l = sample.l[primary_pos] * e.primary.second;
        float r = sample.r[primary_pos] * e.primary.second;
        if (e.countdown > 0) {
            e.countdown -= 1;
            float t = (float)e.countdown / (float)e.lerp_len;
            size_t ghost_pos = clamp_pos(e.ghost.first);
            float gl = sample.l[ghost_pos] * e.ghost.second;
            float gr = sample.r[ghost_pos] * e.ghost.second;
            l = gl * t + l * (1.0f - t);
            r = gr * t + r * (1.0f - t);
        }
        buf.l[i] = l;
        buf.r[i] = r;
    }
    e.primary.first += (int64_t)len;
    e.ghost.first += (int64_t)len;
}

void debug_sine_gen(Sample& buf, FlowwBank& fb, size_t len, VertexExt& e, size_t t, size_t sr) {  // :423-457
    fb.start_block(e.floww_index);
    for (size_t i = 0; i < len; ++i) {
        for (auto& ev : fb.get_block_simple(e.floww_index, i)) {
            if (ev.on) {
                bool has = false;
                for (auto& nv : e.sine_notes) {
                    if (fabsf(nv.note - ev.note) < 0.001f) {
                        nv.vel = ev.vel;
                        has = true;
                        break;
                    }
                }
                if (!has) e.sine_notes.push_back({ev.note, ev.vel});
            } else {
                std::vector<SineNote> kept;
                for (auto& x : e.sine_notes)
                    if (fabsf(x.note - ev.note) > 0.001f) kept.push_back(x);
                e.sine_notes.swap(kept);
            }
        }
        buf.l[i] = 0.0f;
        buf.r[i] = 0.0f;
        for (auto& nv : e.sine_notes) {
            float time = (float)(t + i) / (float)sr;
            float hz = 440.0f * powf(2.0f, (nv.note - 69.0f) / 12.0f);
            float s = sinf(time * hz * 2.0f * PI_F32) * nv.vel;
            buf.l[i] += s;
            buf.r[i] += s;
        }
    }
}

void synth_gen(Sample& buf, FlowwBank& fb, size_t len, VertexExt& e, size_t t, size_t sr) {  // :460-529
    const OscConf& square = e.square; const OscConf& topflat = e.topflat; const OscConf& triangle = e.triangle;
    float osc_amp_multiplier = 1.0f / (square.volume * square.adsr.max_vel() +
                                       topflat.volume * topflat.adsr.max_vel() +
                                       triangle.volume * triangle.adsr.max_vel());
    float release_sec = 0.0f;
    if (square.volume > 0.0f) release_sec = square.adsr.release_sec;
    if (topflat.volume > 0.0f) release_sec = fmaxf(release_sec, topflat.adsr.release_sec);
    if (triangle.volume > 0.0f) release_sec = fmaxf(release_sec, triangle.adsr.release_sec);
    fb.start_block(e.floww_index);
    for (size_t i = 0; i < len; ++i) {
        for (auto& ev : fb.get_block_simple(e.floww_index, i)) {
            if (ev.on) {
                e.notes.push_back({ev.note, ev.vel, -((float)i / (float)sr), 0.0f});
            } else {
                std::vector<SynthNote> kept;
                for (auto& x : e.notes)
                    if (fabsf(x.note - ev.note) > 0.001f || x.rel_t == 0.0f) kept.push_back(x);
                e.notes.swap(kept);
                for (auto& x : e.notes) {
                    if (fabsf(x.note - ev.note) > 0.001f) continue;
                    if (x.rel_t == 0.0f) {
                        x.rel_t = x.env_t + ((float)i / (float)sr);
                        x.env_t = -((float)i / (float)sr);
                    } else {
                        fprintf(stderr, "Synth: impossible release stage note\n");  // :492 panic
                        abort();
                    }
                }
            }
        }
        buf.l[i] = 0.0f;
        buf.r[i] = 0.0f;
        for (auto& x : e.notes) {
            float time = (float)(t + i) / (float)sr;
            float env_time = x.env_t + ((float)i / (float)sr);
            float hz = 440.0f * powf(2.0f, (x.note - 69.0f) / 12.0f);
            auto env_vel = [&](const AdsrConf& c) {
                return x.rel_t == 0.0f ? apply_ads(c, env_time) : apply_r_rt(c, env_time, x.rel_t);
            };
            float s = 0.0f;
            if (square.volume > 0.0f)
                s += square_sine_sample(time, hz, square.param) * x.vel * env_vel(square.adsr) * square.volume;
            if (topflat.volume > 0.0f)
                s += topflat_sine_sample(time, hz, topflat.param) * x.vel * env_vel(topflat.adsr) * topflat.volume;
            if (triangle.volume > 0.0f)
                s += triangle_sample(time, hz) * x.vel * env_vel(triangle.adsr) * triangle.volume;
            s *= osc_amp_multiplier;
            buf.l[i] += s;
            buf.r[i] += s;
        }
    }
    for (auto& x : e.notes) x.env_t += (float)len / (float)sr;
    std::vector<SynthNote> kept;
    for (auto& x : e.notes)
        if (x.rel_t == 0.0f || x.env_t <= release_sec) kept.push_back(x);
    e.notes.swap(kept);
}

void sampsyn_gen(Sample& buf, FlowwBank& fb, size_t len, VertexExt& e, size_t sr) {  // :532-578
    const AdsrConf& adsr = e.conf;
    float amp_multiplier = 1.0f / adsr.max_vel();
    fb.start_block(e.floww_index);
    for (size_t i = 0; i < len; ++i) {
        for (auto& ev : fb.get_block_simple(e.floww_index, i)) {
            if (ev.on) {
                e.notes.push_back({ev.note, ev.vel, -((float)i / (float)sr), 0.0f, initial_state(e.wave_table, 0.0f)});   // :542
            } else {
                std::vector<SynthNote> kept;
                for (auto& x : e.notes)
                    if (fabsf(x.note - ev.note) > 0.001f || x.rel_t == 0.0f) kept.push_back(x);
                e.notes.swap(kept);
                for (auto& x : e.notes) {
                    if (fabsf(x.note - ev.note) > 0.001f) continue;
                    if (x.rel_t == 0.0f) {
                        x.rel_t = x.env_t + ((float)i / (float)sr);
                        x.env_t = -((float)i / (float)sr);
                    } else {
                        fprintf(stderr, "Synth: impossible release stage note\n");  // :552 panic
                        abort();
                    }
                }
            }
        }
        buf.l[i] = 0.0f;
        buf.r[i] = 0.0f;
        for (auto& x : e.notes) {
            float env_time = x.env_t + ((float)i / (float)sr);
            float hz = 440.0f * powf(2.0f, (x.note - 69.0f) / 12.0f);
            float env = x.rel_t == 0.0f ? apply_ads(adsr, env_time) : apply_r_rt(adsr, env_time, x.rel_t);
            float s = 0.0f;
            float vel = x.vel * env * amp_multiplier;
            s += wavetable_act_state(e.wave_table, x.st, hz, env_time + x.rel_t) * vel;   // :569, build-defined oscillator
            buf.l[i] += s;
            buf.r[i] += s;
        }
    }
    for (auto& x : e.notes) x.env_t += (float)len / (float)sr;
    std::vector<SynthNote> kept;
    for (auto& x : e.notes)
        if (x.rel_t == 0.0f || x.env_t <= adsr.release_sec) kept.push_back(x);
    e.notes.swap(kept);
}

void adsr_gen(Sample& buf, size_t len, FlowwBank& fb, float wet, VertexExt& e, size_t sr) {  // :593-651
    if (wet < 0.0001f) return;
    const AdsrConf& conf = e.conf;
    Voice3& primary = e.aprimary; Voice3& ghost = e.aghost;
    float maxmul = e.use_max ? 1.0f : 0.0f;
    float minmul = 1.0f - maxmul;
    fb.start_block(e.floww_index);
    if (e.use_off) {
        for (size_t i = 0; i < len; ++i) {
            float offset = (float)i / (float)sr;
            for (auto& ev : fb.get_block_simple(e.floww_index, i)) {
                if (e.has_note) {
                    if (fabsf((float)e.note - ev.note) > 0.01f) continue;
                }
                if (ev.on) {
                    ghost = primary;
                    primary = {-((float)i / (float)sr), ev.vel, 0.0f};
                } else if (ghost.rel == 0.0f) {
                    ghost.t = -((float)i / (float)sr);
                    ghost.rel = apply_ads(conf, ghost.t + offset) * ghost.vel;
                } else {
                    primary.t = -((float)i / (float)sr);
                    primary.rel = apply_ads(conf, primary.t + offset) * primary.vel;
                }
            }
            float pvel = primary.rel == 0.0f ? apply_ads(conf, primary.t + offset) * primary.vel
                                             : apply_r(conf, primary.t + offset, primary.rel) * primary.vel;
            float gvel = ghost.rel == 0.0f ? apply_ads(conf, ghost.t + offset) * ghost.vel
                                           : apply_r(conf, ghost.t + offset, ghost.rel) * ghost.vel;
            float adsr_vel = fmaxf(pvel, gvel) * maxmul + fminf(pvel, gvel) * minmul;
            float vel = lerp(1.0f, adsr_vel, wet);
            buf.l[i] *= vel;
            buf.r[i] *= vel;
        }
    } else {
        for (size_t i = 0; i < len; ++i) {
            float n, v;
            if (fb.get_block_drum(e.floww_index, i, &n, &v)) {
                if (e.has_note) {
                    if (fabsf((float)e.note - n) > 0.01f) continue;  // :632-635 skips the whole frame
                }
                ghost = primary;
                primary = {-((float)i / (float)sr), v, 0.0f};
            }
            float offset = (float)i / (float)sr;
            float pvel = apply_adsr(conf, primary.t + offset) * primary.vel;
            float gvel = apply_adsr(conf, ghost.t + offset) * ghost.vel;
            float adsr_vel = fmaxf(pvel, gvel) * maxmul + fminf(pvel, gvel) * minmul;
            float vel = lerp(1.0f, adsr_vel, wet);
            buf.l[i] *= vel;
            buf.r[i] *= vel;
        }
    }
    primary.t += (float)len / (float)sr;
    ghost.t += (float)len / (float)sr;
}

void band_pass_gen(Sample& buf, size_t len, float wet, VertexExt& e) {  // :654-689
    if (wet < 0.0001f) return;
    float lgamma = e.lgamma, hgamma = e.hgamma;
    if (lgamma == 0.0f && hgamma == 0.0f) return;
    float lmul = lgamma == 0.0f ? 0.0f : 1.0f;
    float hmul = hgamma == 0.0f ? 0.0f : 1.0f;
    float pass_mul = e.pass ? 1.0f : 0.0f;
    float cut_mul = 1.0f - pass_mul;
    if (e.first) {
        e.lprevl = buf.l[0];
        e.lprevr = buf.r[0];
        e.hprevl = buf.l[0];
        e.hprevr = buf.r[0];
        e.first = false;
    }
    for (size_t i = 0; i < len; ++i) {
        float l = buf.l[i];
        float r = buf.r[i];
        float ll = e.lprevl + lgamma * (l - e.lprevl);
        float lr = e.lprevr + lgamma * (r - e.lprevr);
        float hl = e.hprevl + hgamma * (l - e.hprevl);
        float hr = e.hprevr + hgamma * (r - e.hprevr);
        e.lprevl = ll;
        e.lprevr = lr;
        e.hprevl = hl;
        e.hprevr = hr;
        float cutl = (lmul * ll + hmul * (l - hl)) * 0.5f;
        float cutr = (lmul * lr + hmul * (r - hr)) * 0.5f;
        float passl = l - cutl;
        float passr = r - cutl;  // :685 uses the LEFT cut (quirk Q7) -- reproduced on purpose
        buf.l[i] = cutl * cut_mul + passl * pass_mul;
        buf.r[i] = cutr * cut_mul + passr * pass_mul;
    }
}

// VertexExt::generate (extensions.rs:207-264)
void ext_generate(VertexExt& e, size_t t, size_t sr, size_t len, bool is_scan, const SampleBank& sb,
                  FlowwBank& fb, float gain, float angle, float wet, Sample& buf,
                  const std::vector<const Sample*>& res) {
    if (e.has_input()) sum_inputs(buf, len, res);
    switch (e.kind) {
        case K_SUM: break;
        case K_NORMALIZE: normalize_gen(buf, len, &e.max, &e.scan_max, is_scan); break;
        case K_SAMPLE_LOOP: sample_loop_gen(buf, sb, len, &e.t, e.sample_index); break;  // vertex-own t (:220-221)
        case K_SAMPLE_MULTI: sample_multi_gen(buf, sb, fb, len, e); break;
        case K_SAMPLE_LERP: sample_lerp_gen(buf, sb, fb, len, e); break;
        case K_DEBUG_SINE: debug_sine_gen(buf, fb, len, e, t, sr); break;
        case K_SYNTH: synth_gen(buf, fb, len, e, t, sr); break;
        case K_SAMPSYN: sampsyn_gen(buf, fb, len, e, sr); break;
        case K_ADSR: adsr_gen(buf, len, fb, wet, e, sr); break;
        case K_BAND_PASS: band_pass_gen(buf, len, wet, e); break;
    }
    buf.apply_angle(angle, len);
    buf.apply_gain(gain, len);
}

// ---------------------------------------------------------------------------------------------
// graph.rs
// ---------------------------------------------------------------------------------------------
struct Vertex {  // graph.rs:242-259
    Sample buf;
    float gain, angle, wet;
    VertexExt ext;
    Vertex(size_t bl, float gain_, float angle_, float wet_, const VertexExt& e)
        : buf(bl), gain(gain_), angle(fmaxf(fminf(angle_, 90.0f), -90.0f)),
          wet(fmaxf(fminf(wet_, 1.0f), 0.0f)), ext(e) {}
};

struct Graph {  // graph.rs:12-22
    std::vector<Vertex> vertices;
    std::vector<std::vector<size_t>> edges;
    std::vector<std::string> names;
    std::map<std::string, size_t> name_map;
    std::vector<bool> ran_status;
    long output_vertex = -1;
    size_t max_buffer_len, sr, t = 0;
    Graph(size_t bl, size_t sr_) : max_buffer_len(bl), sr(sr_) {}

    void add(const Vertex& v, const std::string& name) {  // graph.rs:49-56
        vertices.push_back(v);
        ran_status.push_back(false);
        edges.emplace_back();
        name_map[name] = vertices.size() - 1;
        names.push_back(name);
    }
    static bool has_loop(size_t x, size_t b, const std::vector<std::vector<size_t>>& edges) {  // :66-72
        if (x == b) return true;
        for (size_t y : edges[x]) if (has_loop(y, b, edges)) return true;
        return false;
    }
    bool connect_internal(size_t a, size_t b) {  // graph.rs:58-78
        if (a == b) return false;
        size_t len = vertices.size();
        if (a >= len) return false;
        if (b >= len) return false;
        if (!vertices[b].ext.has_input()) return false;
        if (has_loop(a, b, edges)) return false;
        edges[b].push_back(a);
        return true;
    }
    bool connect(const std::string& a, const std::string& b) {  // graph.rs:80-96
        auto ia = name_map.find(a), ib = name_map.find(b);
        if (ia == name_map.end() || ib == name_map.end()) return false;
        return connect_internal(ia->second, ib->second);
    }
    void run_vertex(size_t t_, const SampleBank& sb, FlowwBank& fb, size_t index, bool is_scan) {  // :98-121
        if (index >= vertices.size()) return;
        if (ran_status[index]) return;
        ran_status[index] = true;
        std::vector<size_t> es = edges[index];  // clone per vertex per block, like the reference
        for (size_t incoming : es) run_vertex(t_, sb, fb, incoming, is_scan);
        std::vector<const Sample*> ins;
        for (size_t incoming : es) ins.push_back(&vertices[incoming].buf);
        Vertex& v = vertices[index];
        size_t len = v.buf.len() < max_buffer_len ? v.buf.len() : max_buffer_len;  // Vertex::generate :270
        ext_generate(v.ext, t_, sr, len, is_scan, sb, fb, v.gain, v.angle, v.wet, v.buf, ins);
    }
    void set_time(size_t time) {  // graph.rs:123-128
        t = time;
        for (auto& v : vertices) v.ext.set_time(time);
    }
    bool set_output(const std::string& v) {  // graph.rs:141-148
        auto it = name_map.find(v);
        if (it == name_map.end()) return false;
        output_vertex = (long)it->second;
        return true;
    }
    bool check_graph() const {  // graph.rs:150-174
        if (output_vertex < 0) return false;
        size_t out = (size_t)output_vertex;
        if (edges[out].empty() && vertices[out].ext.has_input()) return false;
        return true;  // unreachable vertices only warn
    }
    void reset_ran_stati() { for (size_t i = 0; i < ran_status.size(); ++i) ran_status[i] = false; }  // :176-180
    const Sample* render(const SampleBank& sb, FlowwBank& fb) {  // graph.rs:182-193
        reset_ran_stati();
        if (output_vertex < 0) return nullptr;
        run_vertex(t, sb, fb, (size_t)output_vertex, false);
        t += max_buffer_len;
        return &vertices[(size_t)output_vertex].buf;
    }
    void reset_normalize_vertices() {  // graph.rs:207-211 / extensions.rs:295-299
        for (auto& v : vertices) if (v.ext.kind == K_NORMALIZE) v.ext.max = 0.000001f;
    }
    void true_normalize_scan(const SampleBank& sb, FlowwBank& fb, size_t chunks) {  // graph.rs:222-237
        if (output_vertex < 0) return;
        for (auto& v : vertices) if (v.ext.kind == K_NORMALIZE) v.ext.scan_max = 0.0f;  // :195-199
        fb.set_time(0);
        for (size_t j = 0; j < chunks; ++j) {
            reset_ran_stati();
            run_vertex(j * max_buffer_len, sb, fb, (size_t)output_vertex, true);
            fb.set_time_to_next_block();
        }
        for (auto& v : vertices) if (v.ext.kind == K_NORMALIZE) v.ext.max = v.ext.scan_max;  // :201-205
        set_time(0);
        fb.set_time(0);
    }
};

thread_local std::string g_err;

}  // namespace

// =================================================================================================
// C API (ctypes surface for tests / smoke / cpu_baseline)
// =================================================================================================
extern "C" {

const char* orc_last_error() { return g_err.c_str(); }

// ---- scalar known-answer hooks ------------------------------------------------------------------
int orc_build_adsr_conf(const float* arr, int n, float* out9) {
    AdsrConf c;
    if (!build_adsr_conf(arr, n, &c)) return 0;
    memcpy(out9, &c, sizeof(float) * 9);
    return 1;
}
static AdsrConf conf_from9(const float* c9) { AdsrConf c; memcpy((void*)&c, c9, sizeof(float) * 9); return c; }
float orc_apply_ads(const float* c9, float t) { return apply_ads(conf_from9(c9), t); }
float orc_apply_adsr(const float* c9, float t) { return apply_adsr(conf_from9(c9), t); }
float orc_apply_r(const float* c9, float t, float old_val) { return apply_r(conf_from9(c9), t, old_val); }
float orc_apply_r_rt(const float* c9, float t, float rt) { return apply_r_rt(conf_from9(c9), t, rt); }
float orc_adsr_max_vel(const float* c9) { return conf_from9(c9).max_vel(); }
float orc_lerp(float a, float b, float t) { return lerp(a, b, t); }
float orc_square_sine_sample(float t, float hz, float z) { return square_sine_sample(t, hz, z); }
float orc_topflat_sine_sample(float t, float hz, float z) { return topflat_sine_sample(t, hz, z); }
float orc_triangle_sample(float t, float hz) { return triangle_sample(t, hz); }
float orc_note_hz(float note) { return note_hz(note); }
// state.rs:104  cs = (psr as f32 * seconds / bl as f32).ceil() as usize
size_t orc_chunk_count(size_t psr, float seconds, size_t bl) {
    return f32_as_usize(ceilf((float)psr * seconds / (float)bl));
}
void orc_pan_amps(float angle, float* l_amp, float* r_amp) {  // sample.rs:99-101
    float angle_rad = angle * 0.5f * 0.01745329f;
    *l_amp = FRAC_1_SQRT_2_F32 * (cosf(angle_rad) + sinf(angle_rad));
    *r_amp = FRAC_1_SQRT_2_F32 * (cosf(angle_rad) - sinf(angle_rad));
}
float orc_bandpass_gamma(float cut_off_hz, size_t sampling_hz) {  // extensions.rs:176-183
    float co = fmaxf(fminf(cut_off_hz, 20000.0f), 0.0f);
    return 1.0f - powf(E_F32, -2.0f * PI_F32 * co / (float)sampling_hz);
}
float orc_amplitude(size_t bd) {  // state.rs:515-516
    return bd < 32 ? (float)((1 << (bd - 1)) - 1) : (float)INT32_MAX;
}
int16_t orc_quantise16(float x, float amplitude) { return f32_as_i16(x * amplitude); }  // state.rs:521
int32_t orc_quantise32(float x, float amplitude) { return f32_as_i32(x * amplitude); }  // state.rs:529
size_t orc_frame_of(float t_sec, size_t sr) { return f32_as_usize(t_sec * (float)sr); }  // floww.rs:75

// ---- SampleBank -----------------------------------------------------------------------------------
void* orc_sb_new(size_t sr) { return new SampleBank(sr); }
void orc_sb_free(void* sb) { delete (SampleBank*)sb; }
int orc_sb_add_decoded(void* sb, const char* name, const float* linear, size_t n, int channels, size_t sr,
                       size_t bd, const char* method) {
    std::vector<float> v(linear, linear + n);
    return ((SampleBank*)sb)->add_decoded(name, v, channels, sr, bd, load_method_from(method), &g_err) ? 1 : 0;
}
int orc_sb_add_file(void* sb, const char* name, const char* path, const char* method) {
    std::vector<float> linear; int ch = 0; size_t sr = 0, bd = 0;
    SampleBank* b = (SampleBank*)sb;
    if (b->names.count(name)) { g_err = "SampleBank: there is already a sample with that name."; return 0; }
    if (!read_wav(path, &linear, &ch, &sr, &bd, &g_err)) return 0;
    return b->add_decoded(name, linear, ch, sr, bd, load_method_from(method), &g_err) ? 1 : 0;
}
long orc_sb_get_index(void* sb, const char* name) { return ((SampleBank*)sb)->get_index(name); }
size_t orc_sb_len(void* sb, size_t idx) { return ((SampleBank*)sb)->samples[idx].len(); }
void orc_sb_read(void* sb, size_t idx, float* l, float* r) {
    const Sample& s = ((SampleBank*)sb)->samples[idx];
    memcpy(l, s.l.data(), s.l.size() * 4);
    memcpy(r, s.r.data(), s.r.size() * 4);
}

// ---- FlowwBank ------------------------------------------------------------------------------------
void* orc_fb_new(size_t sr, size_t bl) { return new FlowwBank(sr, bl); }
void orc_fb_free(void* fb) { delete (FlowwBank*)fb; }
// events: n triples (t_sec, note, vel); replaces read_floww_from_midi (floww.rs:40-48, un-vendored)
long orc_fb_add_events(void* fb, const char* name, const float* triples, size_t n) {
    std::vector<Event> ev(n);
    for (size_t i = 0; i < n; ++i) ev[i] = {triples[3 * i], triples[3 * i + 1], triples[3 * i + 2]};
    return (long)((FlowwBank*)fb)->declare_floww(name, ev);
}
long orc_fb_get_index(void* fb, const char* name) { return ((FlowwBank*)fb)->get_index(name); }
long orc_fb_declare_stream(void* fb, const char* name) { return (long)((FlowwBank*)fb)->declare_stream(name); }
long orc_fb_append_stream(void* fb, const char* name, const float* triples, size_t n) {
    std::vector<Event> ev(n);
    for (size_t i = 0; i < n; ++i) ev[i] = {triples[3 * i], triples[3 * i + 1], triples[3 * i + 2]};
    return ((FlowwBank*)fb)->append_stream(name, ev);
}
void orc_fb_trim_streams(void* fb) { ((FlowwBank*)fb)->trim_streams(); }
size_t orc_fb_get_events(void* fb, size_t index, float* out, size_t cap) {
    FlowwBank* b = (FlowwBank*)fb;
    if (index >= b->flowws.size()) return 0;
    const auto& f = b->flowws[index];
    for (size_t i = 0; i < f.size() && i < cap; ++i) { out[3 * i] = f[i].t; out[3 * i + 1] = f[i].note; out[3 * i + 2] = f[i].vel; }
    return f.size();
}
void orc_fb_set_time(void* fb, size_t t) { ((FlowwBank*)fb)->set_time(t); }
void orc_fb_set_time_to_next_block(void* fb) { ((FlowwBank*)fb)->set_time_to_next_block(); }
void orc_fb_start_block(void* fb, size_t index) { ((FlowwBank*)fb)->start_block(index); }
int orc_fb_get_block_drum(void* fb, size_t index, size_t offset, float* note, float* vel) {
    return ((FlowwBank*)fb)->get_block_drum(index, offset, note, vel) ? 1 : 0;
}
// out: cap triples (on, note, vel); returns the count
size_t orc_fb_get_block_simple(void* fb, size_t index, size_t offset, float* out, size_t cap) {
    auto v = ((FlowwBank*)fb)->get_block_simple(index, offset);
    for (size_t i = 0; i < v.size() && i < cap; ++i) {
        out[3 * i] = v[i].on ? 1.0f : 0.0f;
        out[3 * i + 1] = v[i].note;
        out[3 * i + 2] = v[i].vel;
    }
    return v.size();
}

// ---- Graph ----------------------------------------------------------------------------------------
void* orc_graph_new(size_t bl, size_t sr) { return new Graph(bl, sr); }
void orc_graph_free(void* g) { delete (Graph*)g; }
static void g_add(void* g, const char* name, float gain, float angle, float wet, const VertexExt& e) {
    Graph* gr = (Graph*)g;
    gr->add(Vertex(gr->max_buffer_len, gain, angle, wet, e), name);
}
// The add_* functions follow state.rs:341-457 (argument fix-ups) + extensions.rs:83-194 (constructors).
void orc_graph_add_sum(void* g, const char* name, float gain, float angle) {
    VertexExt e; e.kind = K_SUM; g_add(g, name, gain, angle, 0.0f, e);
}
void orc_graph_add_normalize(void* g, const char* name, float gain, float angle) {
    VertexExt e; e.kind = K_NORMALIZE; e.max = 0.0f; e.scan_max = 0.0f; g_add(g, name, gain, angle, 0.0f, e);
}
void orc_graph_add_sampleloop(void* g, const char* name, float gain, float angle, size_t sample) {
    VertexExt e; e.kind = K_SAMPLE_LOOP; e.sample_index = sample; e.t = 0; g_add(g, name, gain, angle, 0.0f, e);
}
void orc_graph_add_sample_multi(void* g, const char* name, float gain, float angle, size_t sample,
                                size_t floww, int note) {
    VertexExt e; e.kind = K_SAMPLE_MULTI; e.sample_index = sample; e.floww_index = floww;
    e.has_note = !(note < 0); e.note = e.has_note ? (size_t)note : 0;  // state.rs:358-359
    g_add(g, name, gain, angle, 0.0f, e);
}
void orc_graph_add_sample_lerp(void* g, const char* name, float gain, float angle, size_t sample,
                               size_t floww, int note, int lerp_len) {
    VertexExt e; e.kind = K_SAMPLE_LERP; e.sample_index = sample; e.floww_index = floww;
    e.has_note = !(note < 0); e.note = e.has_note ? (size_t)note : 0;  // state.rs:368-369
    e.lerp_len = (size_t)(lerp_len > 0 ? lerp_len : 0);                // state.rs:370
    g_add(g, name, gain, angle, 0.0f, e);
}
void orc_graph_add_debug_sine(void* g, const char* name, float gain, float angle, size_t floww) {
    VertexExt e; e.kind = K_DEBUG_SINE; e.floww_index = floww; g_add(g, name, gain, angle, 0.0f, e);
}
int orc_graph_add_synth(void* g, const char* name, float gain, float angle, size_t floww, float sq_vel,
                        float sq_z, const float* sq_arr, int sq_n, float tf_vel, float tf_z,
                        const float* tf_arr, int tf_n, float tr_vel, const float* tr_arr, int tr_n) {
    VertexExt e; e.kind = K_SYNTH; e.floww_index = floww;
    if (!build_adsr_conf(sq_arr, sq_n, &e.square.adsr) || !build_adsr_conf(tf_arr, tf_n, &e.topflat.adsr) ||
        !build_adsr_conf(tr_arr, tr_n, &e.triangle.adsr)) {
        g_err = "ADSR config must have 6 or 9 elements";  // state.rs:393 (panic in the reference)
        return 0;
    }
    e.square.volume = sq_vel; e.square.param = fmaxf(sq_z, 0.0001f);  // state.rs:400
    e.topflat.volume = tf_vel; e.topflat.param = tf_z;                  // state.rs:401
    e.triangle.volume = tr_vel; e.triangle.param = 0.0f;                // state.rs:402
    g_add(g, name, gain, angle, 0.0f, e);
    return 1;
}
// add_sampsyn (state.rs:406-426): table = parse(resource bytes) or the default table (state.rs:415-422)
int orc_graph_add_sampsyn(void* g, const char* name, float gain, float angle, size_t floww, const float* arr, int n,
                          const uint8_t* table_bytes, size_t table_len) {
    VertexExt e; e.kind = K_SAMPSYN; e.floww_index = floww;
    if (!build_adsr_conf(arr, n, &e.conf)) { g_err = "ADSR config must have 6 or 9 elements"; return 0; }
    if (!table_bytes || !parse_wavetable(table_bytes, table_len, &e.wave_table)) e.wave_table = default_wavetable();
    g_add(g, name, gain, angle, 0.0f, e);
    return 1;
}
float orc_wavetable_act(const uint8_t* table_bytes, size_t table_len, float hz, float t) {
    WaveTable w;
    if (!table_bytes || !parse_wavetable(table_bytes, table_len, &w)) w = default_wavetable();
    return wavetable_act(w, hz, t);
}
int orc_graph_add_adsr(void* g, const char* name, float gain, float angle, float wet, size_t floww,
                       int use_off, int use_max, int note, const float* arr, int n) {
    VertexExt e; e.kind = K_ADSR; e.floww_index = floww; e.use_off = use_off != 0; e.use_max = use_max != 0;
    e.has_note = !(note < 0); e.note = e.has_note ? (size_t)note : 0;  // state.rs:439-440
    if (!build_adsr_conf(arr, n, &e.conf)) { g_err = "ADSR config must have 6 or 9 elements"; return 0; }
    g_add(g, name, gain, angle, wet, e);
    return 1;
}
void orc_graph_add_bandpass(void* g, const char* name, float gain, float angle, float wet, float lo_hz,
                            float hi_hz, int pass) {
    VertexExt e; e.kind = K_BAND_PASS;
    e.lgamma = orc_bandpass_gamma(lo_hz, ((Graph*)g)->sr);  // extensions.rs:173-194; psr from state.rs:455
    e.hgamma = orc_bandpass_gamma(hi_hz, ((Graph*)g)->sr);
    e.first = true; e.pass = pass != 0;
    g_add(g, name, gain, angle, wet, e);
}
int orc_graph_connect(void* g, const char* a, const char* b) { return ((Graph*)g)->connect(a, b) ? 1 : 0; }
int orc_graph_set_output(void* g, const char* v) { return ((Graph*)g)->set_output(v) ? 1 : 0; }
int orc_graph_check(void* g) { return ((Graph*)g)->check_graph() ? 1 : 0; }
void orc_graph_set_time(void* g, size_t t) { ((Graph*)g)->set_time(t); }
size_t orc_graph_get_time(void* g) { return ((Graph*)g)->t; }
void orc_graph_reset_normalize_vertices(void* g) { ((Graph*)g)->reset_normalize_vertices(); }
float orc_graph_get_normalization_value(void* g, const char* name) {  // extensions.rs:301-307
    Graph* gr = (Graph*)g;
    auto it = gr->name_map.find(name);
    if (it == gr->name_map.end()) return -1.0f;
    const VertexExt& e = gr->vertices[it->second].ext;
    return e.kind == K_NORMALIZE ? e.max : -1.0f;
}
// Graph::render -> copies the output block into l/r (bl floats each). 0 when there is no output vertex.
int orc_graph_render(void* g, void* sb, void* fb, float* l, float* r) {
    const Sample* s = ((Graph*)g)->render(*(SampleBank*)sb, *(FlowwBank*)fb);
    if (!s) return 0;
    if (l) memcpy(l, s->l.data(), s->l.size() * 4);
    if (r) memcpy(r, s->r.data(), s->r.size() * 4);
    return 1;
}
void orc_graph_true_normalize_scan(void* g, void* sb, void* fb, size_t chunks) {
    ((Graph*)g)->true_normalize_scan(*(SampleBank*)sb, *(FlowwBank*)fb, chunks);
}

// State::render main loop, `psr <= render_sr` arm (state.rs:562-575): cs blocks, quantise to
// interleaved integer PCM in memory (write_16s for bd<=16 -> int16 words; write_32s otherwise ->
// int32 words), fb.set_time_to_next_block per block, g.set_time(0) at the end.  `out_f32` (optional)
// receives the un-quantised interleaved float frames for tolerance-class comparisons.
// Returns the number of frames rendered.
size_t orc_state_render(void* g, void* sb, void* fb, size_t cs, size_t bd, void* out_pcm, float* out_f32) {
    Graph* gr = (Graph*)g;
    float amplitude = orc_amplitude(bd);
    size_t frames = 0;
    for (size_t c = 0; c < cs; ++c) {
        const Sample* chunk = gr->render(*(SampleBank*)sb, *(FlowwBank*)fb);
        if (!chunk) continue;
        size_t len = chunk->len();
        for (size_t i = 0; i < len; ++i) {
            if (out_pcm) {
                if (bd > 16) {
                    ((int32_t*)out_pcm)[2 * (frames + i)] = f32_as_i32(chunk->l[i] * amplitude);
                    ((int32_t*)out_pcm)[2 * (frames + i) + 1] = f32_as_i32(chunk->r[i] * amplitude);
                } else {
                    ((int16_t*)out_pcm)[2 * (frames + i)] = f32_as_i16(chunk->l[i] * amplitude);
                    ((int16_t*)out_pcm)[2 * (frames + i) + 1] = f32_as_i16(chunk->r[i] * amplitude);
                }
            }
            if (out_f32) {
                out_f32[2 * (frames + i)] = chunk->l[i];
                out_f32[2 * (frames + i) + 1] = chunk->r[i];
            }
        }
        frames += len;
        ((FlowwBank*)fb)->set_time_to_next_block();
    }
    gr->set_time(0);
    return frames;
}

// State::render's `psr > render_sr` arm (state.rs:533-561) with the build-defined resampler applied to the
// whole rendered timeline (the reference streams rubato block by block; parity unpinned).  Returns the
// number of output frames; out_pcm / out_f32 may be NULL to query it.
size_t orc_state_render_resampled(void* g, void* sb, void* fb, size_t cs, size_t bd, size_t psr, size_t render_sr,
                                  void* out_pcm, float* out_f32) {
    Graph* gr = (Graph*)g;
    const size_t frames = cs * gr->max_buffer_len;
    const size_t nout = (size_t)(((unsigned __int128)frames * render_sr + psr - 1) / psr);
    if (!out_pcm && !out_f32) return nout;
    std::vector<float> l(frames), r(frames);
    size_t done = 0;
    for (size_t c = 0; c < cs; ++c) {
        const Sample* chunk = gr->render(*(SampleBank*)sb, *(FlowwBank*)fb);
        if (!chunk) continue;
        memcpy(&l[done], chunk->l.data(), chunk->len() * 4);
        memcpy(&r[done], chunk->r.data(), chunk->len() * 4);
        done += chunk->len();
        ((FlowwBank*)fb)->set_time_to_next_block();
    }
    gr->set_time(0);
    std::vector<float> ol, orr;
    resample_planar(l, r, psr, render_sr, &ol, &orr);
    const float amplitude = orc_amplitude(bd);
    for (size_t i = 0; i < nout; ++i) {
        if (out_pcm) {
            if (bd > 16) {
                ((int32_t*)out_pcm)[2 * i] = f32_as_i32(ol[i] * amplitude);
                ((int32_t*)out_pcm)[2 * i + 1] = f32_as_i32(orr[i] * amplitude);
            } else {
                ((int16_t*)out_pcm)[2 * i] = f32_as_i16(ol[i] * amplitude);
                ((int16_t*)out_pcm)[2 * i + 1] = f32_as_i16(orr[i] * amplitude);
            }
        }
        if (out_f32) { out_f32[2 * i] = ol[i]; out_f32[2 * i + 1] = orr[i]; }
    }
    return nout;
}

}  // extern "C"
