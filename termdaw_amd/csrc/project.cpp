// project.cpp -- the project front-end: State (state.rs:27-578) behind td_state_*.
//
// State::refresh (state.rs:50-471): run the project script -- whose 23 registered globals only RECORD
// their arguments (state.rs:83-157) -- then diff samples, rebuild the FlowwBank, rebuild the graph
// with vertices grouped by type in the fixed order of state.rs:341-457, connect edges in call order,
// set the output, check the graph and reset the normalize vertices.  State::render (state.rs:477-577)
// drives the whole-timeline GPU render and writes the integer WAV.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <fstream>
#include <sstream>
#include <thread>

#include <hip/hip_runtime.h>

#include "engine.h"
#include "devmem.h"
#include "lua_subset.h"
#include "wav.h"
#include "midi.h"

using namespace tde;
using tdl::LuaError;
using tdl::Value;

namespace {

// ---- mlua-style argument conversion (FromLua for f32 / usize / i32 / String / bool / Vec<f32>) ----
const Value& arg_at(const std::vector<Value>& a, size_t i) {
    static const Value nil;
    return i < a.size() ? a[i] : nil;
}
[[noreturn]] void bad(const char* fn, size_t i, const char* want, const Value& got) {
    static const char* tn[] = {"nil", "boolean", "integer", "number", "string", "table", "function"};
    throw LuaError{std::string("bad argument #") + std::to_string(i + 1) + " to '" + fn + "': error converting Lua " +
                   tn[got.type] + " to " + want};
}
double to_double(const char* fn, const std::vector<Value>& a, size_t i, const char* want) {
    const Value& v = arg_at(a, i);
    if (v.type == Value::INT) return (double)v.i;
    if (v.type == Value::FLT) return v.d;
    if (v.type == Value::STR) {
        char* end = nullptr;
        double d = strtod(v.s.c_str(), &end);
        if (end && *end == 0 && !v.s.empty()) return d;
    }
    bad(fn, i, want, v);
}
float to_f32(const char* fn, const std::vector<Value>& a, size_t i) { return (float)to_double(fn, a, i, "f32"); }
long long to_int(const char* fn, const std::vector<Value>& a, size_t i, const char* want, long long lo, long long hi) {
    const Value& v = arg_at(a, i);
    long long r;
    if (v.type == Value::INT) r = v.i;
    else {
        double d = to_double(fn, a, i, want);
        if (d != floor(d) || d < -9.2e18 || d > 9.2e18) bad(fn, i, want, v);
        r = (long long)d;
    }
    if (r < lo || r > hi) bad(fn, i, want, v);
    return r;
}
std::string to_str(const char* fn, const std::vector<Value>& a, size_t i) {
    const Value& v = arg_at(a, i);
    if (v.type == Value::STR) return v.s;
    if (v.is_number()) return tdl::tostring(v);
    bad(fn, i, "String", v);
}
bool to_bool(const std::vector<Value>& a, size_t i) { return arg_at(a, i).truthy(); }
std::vector<float> to_vecf(const char* fn, const std::vector<Value>& a, size_t i) {
    const Value& v = arg_at(a, i);
    if (v.type != Value::TAB) bad(fn, i, "Vec<f32>", v);
    std::vector<float> out;
    for (auto& e : v.tab->arr) {
        if (e.type == Value::INT) out.push_back((float)e.i);
        else if (e.type == Value::FLT) out.push_back((float)e.d);
        else bad(fn, i, "Vec<f32>", v);
    }
    return out;
}

std::string fnum(float v) {
    char b[64];
    snprintf(b, sizeof b, "%.9g", (double)v);
    return b;
}
std::string fvec(const std::vector<float>& v) {
    std::string s = "{";
    for (size_t i = 0; i < v.size(); ++i) s += (i ? "," : "") + fnum(v[i]);
    return s + "}";
}

struct SumCall { std::string name; float gain, angle; };
struct LoopCall { std::string name; float gain, angle; std::string sample; };
struct MultiCall { std::string name; float gain, angle; std::string sample, floww; int note; };
struct LerpCall { std::string name; float gain, angle; std::string sample, floww; int note, lerp_len; };
struct SineCall { std::string name; float gain, angle; std::string floww; };
struct SynthCall { std::string name; float gain, angle; std::string floww; float sq_vel, sq_z; std::vector<float> sq; float tf_vel, tf_z; std::vector<float> tf; float tr_vel; std::vector<float> tr; };
struct SampsynCall { std::string name; float gain, angle; std::string floww; std::vector<float> adsr; std::string resource; };
struct Lv2fxCall { std::string name; float gain, angle, wet; std::string plugin; };
struct AdsrCall { std::string name; float gain, angle, wet; std::string floww; bool use_off, use_max; int note; std::vector<float> conf; };
struct BandCall { std::string name; float gain, angle, wet, lo, hi; bool pass; };

using Triple = std::tuple<std::string, std::string, std::string>;

}  // namespace

struct td_state {
    std::string wdir;
    std::string main_file = "project.lua";
    size_t psr = 44100, bl = 1024;   // config.rs:58-64 defaults
    td_samplebank* sb = nullptr;
    td_flowwbank* fb = nullptr;
    td_graph* g = nullptr;
    bool loaded = false;
    size_t cs = 0, render_sr = 48000, bd = 16;   // main.rs:88-90
    std::string output_vertex, output_file = "outp.wav";   // main.rs:91-92
    std::vector<Triple> cur_samples;
    std::string dump;
    // read-back buffer of the last render: page-locked, so the device-to-host copy runs at PCIe speed
    struct Pinned {
        uint8_t* p = nullptr;
        size_t n = 0, cap = 0;
        uint8_t* data() { return p; }
        size_t size() const { return n; }
        void clear() { n = 0; }
        bool resize(size_t want) {
            if (want > cap) {
                if (p) (void)hipHostFree(p);
                p = nullptr;
                cap = 0;
                if (hipHostMalloc((void**)&p, want + want / 4 + 4096, hipHostMallocDefault) != hipSuccess) return false;
                cap = want + want / 4 + 4096;
            }
            n = want;
            return true;
        }
        ~Pinned() { if (p) (void)hipHostFree(p); }
    } host_pcm;
    size_t out_frames = 0;
};

namespace {

void bank_remove(td_samplebank* sb, const std::string& name) {   // mark_dead + refresh, sample.rs:316-336
    auto it = sb->names.find(name);
    if (it == sb->names.end()) return;
    const size_t idx = it->second;
    if (hipSetDevice(sb->device) == hipSuccess) {
        sb->release(sb->samples[idx].d);
        sb->release(sb->samples[idx].d16);
    }
    sb->samples.erase(sb->samples.begin() + (long)idx);
    sb->names.erase(it);
    for (auto& kv : sb->names)
        if (kv.second > idx) --kv.second;
}

// Event lists for load_midi_floww: the reference reads MIDI through the un-vendored floww crate
// (floww.rs:40-48).  This front-end reads .mid / .midi files with its own SMF reader (midi.h) and, for any
// other name, a plain-text list: one event per line, "<t_sec> <note> <vel>" (decimal or C99 hex floats),
// '#' comments.  Anything else fails loudly.
bool read_event_file(const std::string& path, std::vector<td_event>* out, std::string* err) {
    std::ifstream f(path);
    if (!f) { *err = "Could not read midi file: \"" + path + "\"."; return false; }
    if (path.size() >= 4) {
        std::string ext = path.substr(path.size() - 4);
        for (auto& c : ext) c = (char)tolower(c);
        if (ext == ".mid" || ext == "midi") {   // Standard MIDI File: this library's own reader (midi.h)
            std::string merr;
            if (tde::read_midi_file(path, out, &merr)) return true;
            *err = "Could not read midi file: \"" + path + "\" (" + merr + ").";
            return false;
        }
    }
    std::string line;
    while (std::getline(f, line)) {
        size_t h = line.find('#');
        if (h != std::string::npos) line.resize(h);
        std::istringstream ss(line);
        std::string a, b, c;
        if (!(ss >> a)) continue;
        if (!(ss >> b >> c)) { *err = "malformed event line in \"" + path + "\""; return false; }
        out->push_back({strtof(a.c_str(), nullptr), strtof(b.c_str(), nullptr), strtof(c.c_str(), nullptr)});
    }
    return true;
}

std::string join_path(const std::string& dir, const std::string& f) {
    if (dir.empty() || (!f.empty() && f[0] == '/')) return f;
    return dir + (dir.back() == '/' ? "" : "/") + f;
}

int do_refresh(td_state* s, const std::string& contents) {
    s->loaded = false;
    s->dump.clear();
    const size_t psr = s->psr, bl = s->bl;
    std::vector<Triple> new_samples;
    std::vector<std::pair<std::string, std::string>> new_resources, midis, new_lv2plugins, edges;
    std::vector<std::string> streams;
    std::vector<SumCall> sums, norms;
    std::vector<LoopCall> sampleloops;
    std::vector<MultiCall> samplemultis;
    std::vector<LerpCall> samplelerps;
    std::vector<SineCall> debugsines;
    std::vector<SynthCall> synths;
    std::vector<SampsynCall> sampsyns;
    std::vector<Lv2fxCall> lv2fxs;
    std::vector<AdsrCall> adsrs;
    std::vector<BandCall> bandpasses;
    size_t cs = s->cs, render_sr = s->render_sr, bd = s->bd;
    // std::mem::take (state.rs:79-80): the locals start from the previous values, the State's own fields are left
    // EMPTY until the script has run (state.rs:169-170) -- so they stay empty when the script fails
    std::string output_file = std::move(s->output_file), output_vertex = std::move(s->output_vertex);
    s->output_file.clear();
    s->output_vertex.clear();
    std::string& dump = s->dump;
    const long long IMAX = 2147483647LL, IMIN = -2147483648LL, UMAX = 9223372036854775807LL;

    tdl::Interp lua;
    auto V = [](const std::vector<Value>&) { return Value::nil(); };
    (void)V;
    lua.set_function("set_length", [&](const std::vector<Value>& a) {   // state.rs:103-106
        const float seconds = to_f32("set_length", a, 0);
        cs = f32_as_usize(ceilf((float)psr * seconds / (float)bl));
        dump += "set_length(" + fnum(seconds) + ")\n";
        return Value::nil();
    });
    lua.set_function("set_render_samplerate", [&](const std::vector<Value>& a) {
        render_sr = (size_t)to_int("set_render_samplerate", a, 0, "usize", 0, UMAX);
        dump += "set_render_samplerate(" + std::to_string(render_sr) + ")\n";
        return Value::nil();
    });
    lua.set_function("set_render_bitdepth", [&](const std::vector<Value>& a) {
        bd = (size_t)to_int("set_render_bitdepth", a, 0, "usize", 0, UMAX);
        dump += "set_render_bitdepth(" + std::to_string(bd) + ")\n";
        return Value::nil();
    });
    lua.set_function("set_output_file", [&](const std::vector<Value>& a) {
        output_file = to_str("set_output_file", a, 0);
        dump += "set_output_file(\"" + output_file + "\")\n";
        return Value::nil();
    });
    lua.set_function("load_sample", [&](const std::vector<Value>& a) {   // state.rs:112
        // registered as (String, String, String); the README calls it with two arguments
        // (README.md:100-101) -- a missing mode is accepted as "" (harmless superset)
        std::string mode = arg_at(a, 2).type == Value::NIL ? "" : to_str("load_sample", a, 2);
        new_samples.emplace_back(to_str("load_sample", a, 0), to_str("load_sample", a, 1), mode);
        dump += "load_sample(\"" + std::get<0>(new_samples.back()) + "\",\"" + std::get<1>(new_samples.back()) + "\",\"" + mode + "\")\n";
        return Value::nil();
    });
    lua.set_function("load_resource", [&](const std::vector<Value>& a) {
        new_resources.push_back({to_str("load_resource", a, 0), to_str("load_resource", a, 1)});
        dump += "load_resource(\"" + new_resources.back().first + "\",\"" + new_resources.back().second + "\")\n";
        return Value::nil();
    });
    lua.set_function("load_midi_floww", [&](const std::vector<Value>& a) {
        midis.push_back({to_str("load_midi_floww", a, 0), to_str("load_midi_floww", a, 1)});
        dump += "load_midi_floww(\"" + midis.back().first + "\",\"" + midis.back().second + "\")\n";
        return Value::nil();
    });
    lua.set_function("declare_stream", [&](const std::vector<Value>& a) {
        streams.push_back(to_str("declare_stream", a, 0));
        dump += "declare_stream(\"" + streams.back() + "\")\n";
        return Value::nil();
    });
    lua.set_function("load_lv2", [&](const std::vector<Value>& a) {
        new_lv2plugins.push_back({to_str("load_lv2", a, 0), to_str("load_lv2", a, 1)});
        dump += "load_lv2(\"" + new_lv2plugins.back().first + "\",\"" + new_lv2plugins.back().second + "\")\n";
        return Value::nil();
    });
    lua.set_function("parameter", [&](const std::vector<Value>& a) {
        dump += "parameter(\"" + to_str("parameter", a, 0) + "\",\"" + to_str("parameter", a, 1) + "\"," + fnum(to_f32("parameter", a, 2)) + ")\n";
        return Value::nil();
    });
    lua.set_function("add_sum", [&](const std::vector<Value>& a) {
        sums.push_back({to_str("add_sum", a, 0), to_f32("add_sum", a, 1), to_f32("add_sum", a, 2)});
        dump += "add_sum(\"" + sums.back().name + "\"," + fnum(sums.back().gain) + "," + fnum(sums.back().angle) + ")\n";
        return Value::nil();
    });
    lua.set_function("add_normalize", [&](const std::vector<Value>& a) {
        norms.push_back({to_str("add_normalize", a, 0), to_f32("add_normalize", a, 1), to_f32("add_normalize", a, 2)});
        dump += "add_normalize(\"" + norms.back().name + "\"," + fnum(norms.back().gain) + "," + fnum(norms.back().angle) + ")\n";
        return Value::nil();
    });
    lua.set_function("add_sampleloop", [&](const std::vector<Value>& a) {
        const char* f = "add_sampleloop";
        sampleloops.push_back({to_str(f, a, 0), to_f32(f, a, 1), to_f32(f, a, 2), to_str(f, a, 3)});
        auto& c = sampleloops.back();
        dump += std::string(f) + "(\"" + c.name + "\"," + fnum(c.gain) + "," + fnum(c.angle) + ",\"" + c.sample + "\")\n";
        return Value::nil();
    });
    lua.set_function("add_sample_multi", [&](const std::vector<Value>& a) {
        const char* f = "add_sample_multi";
        samplemultis.push_back({to_str(f, a, 0), to_f32(f, a, 1), to_f32(f, a, 2), to_str(f, a, 3), to_str(f, a, 4),
                                (int)to_int(f, a, 5, "i32", IMIN, IMAX)});
        auto& c = samplemultis.back();
        dump += std::string(f) + "(\"" + c.name + "\"," + fnum(c.gain) + "," + fnum(c.angle) + ",\"" + c.sample + "\",\"" + c.floww + "\"," + std::to_string(c.note) + ")\n";
        return Value::nil();
    });
    lua.set_function("add_sample_lerp", [&](const std::vector<Value>& a) {
        const char* f = "add_sample_lerp";
        samplelerps.push_back({to_str(f, a, 0), to_f32(f, a, 1), to_f32(f, a, 2), to_str(f, a, 3), to_str(f, a, 4),
                               (int)to_int(f, a, 5, "i32", IMIN, IMAX), (int)to_int(f, a, 6, "i32", IMIN, IMAX)});
        auto& c = samplelerps.back();
        dump += std::string(f) + "(\"" + c.name + "\"," + fnum(c.gain) + "," + fnum(c.angle) + ",\"" + c.sample + "\",\"" + c.floww + "\"," + std::to_string(c.note) + "," + std::to_string(c.lerp_len) + ")\n";
        return Value::nil();
    });
    lua.set_function("add_debug_sine", [&](const std::vector<Value>& a) {
        const char* f = "add_debug_sine";
        debugsines.push_back({to_str(f, a, 0), to_f32(f, a, 1), to_f32(f, a, 2), to_str(f, a, 3)});
        auto& c = debugsines.back();
        dump += std::string(f) + "(\"" + c.name + "\"," + fnum(c.gain) + "," + fnum(c.angle) + ",\"" + c.floww + "\")\n";
        return Value::nil();
    });
    lua.set_function("add_synth", [&](const std::vector<Value>& a) {
        const char* f = "add_synth";
        synths.push_back({to_str(f, a, 0), to_f32(f, a, 1), to_f32(f, a, 2), to_str(f, a, 3), to_f32(f, a, 4), to_f32(f, a, 5),
                          to_vecf(f, a, 6), to_f32(f, a, 7), to_f32(f, a, 8), to_vecf(f, a, 9), to_f32(f, a, 10), to_vecf(f, a, 11)});
        auto& c = synths.back();
        dump += std::string(f) + "(\"" + c.name + "\"," + fnum(c.gain) + "," + fnum(c.angle) + ",\"" + c.floww + "\"," + fnum(c.sq_vel) + "," + fnum(c.sq_z) + "," + fvec(c.sq) + "," +
                fnum(c.tf_vel) + "," + fnum(c.tf_z) + "," + fvec(c.tf) + "," + fnum(c.tr_vel) + "," + fvec(c.tr) + ")\n";
        return Value::nil();
    });
    lua.set_function("add_sampsyn", [&](const std::vector<Value>& a) {
        const char* f = "add_sampsyn";
        sampsyns.push_back({to_str(f, a, 0), to_f32(f, a, 1), to_f32(f, a, 2), to_str(f, a, 3), to_vecf(f, a, 4), to_str(f, a, 5)});
        auto& c = sampsyns.back();
        dump += std::string(f) + "(\"" + c.name + "\"," + fnum(c.gain) + "," + fnum(c.angle) + ",\"" + c.floww + "\"," + fvec(c.adsr) + ",\"" + c.resource + "\")\n";
        return Value::nil();
    });
    lua.set_function("add_lv2fx", [&](const std::vector<Value>& a) {
        const char* f = "add_lv2fx";
        lv2fxs.push_back({to_str(f, a, 0), to_f32(f, a, 1), to_f32(f, a, 2), to_f32(f, a, 3), to_str(f, a, 4)});
        auto& c = lv2fxs.back();
        dump += std::string(f) + "(\"" + c.name + "\"," + fnum(c.gain) + "," + fnum(c.angle) + "," + fnum(c.wet) + ",\"" + c.plugin + "\")\n";
        return Value::nil();
    });
    lua.set_function("add_adsr", [&](const std::vector<Value>& a) {
        const char* f = "add_adsr";
        adsrs.push_back({to_str(f, a, 0), to_f32(f, a, 1), to_f32(f, a, 2), to_f32(f, a, 3), to_str(f, a, 4), to_bool(a, 5), to_bool(a, 6),
                         (int)to_int(f, a, 7, "i32", IMIN, IMAX), to_vecf(f, a, 8)});
        auto& c = adsrs.back();
        dump += std::string(f) + "(\"" + c.name + "\"," + fnum(c.gain) + "," + fnum(c.angle) + "," + fnum(c.wet) + ",\"" + c.floww + "\"," + (c.use_off ? "true" : "false") + "," +
                (c.use_max ? "true" : "false") + "," + std::to_string(c.note) + "," + fvec(c.conf) + ")\n";
        return Value::nil();
    });
    lua.set_function("add_bandpass", [&](const std::vector<Value>& a) {
        const char* f = "add_bandpass";
        bandpasses.push_back({to_str(f, a, 0), to_f32(f, a, 1), to_f32(f, a, 2), to_f32(f, a, 3), to_f32(f, a, 4), to_f32(f, a, 5), to_bool(a, 6)});
        auto& c = bandpasses.back();
        dump += std::string(f) + "(\"" + c.name + "\"," + fnum(c.gain) + "," + fnum(c.angle) + "," + fnum(c.wet) + "," + fnum(c.lo) + "," + fnum(c.hi) + "," + (c.pass ? "true" : "false") + ")\n";
        return Value::nil();
    });
    lua.set_function("connect", [&](const std::vector<Value>& a) {
        edges.push_back({to_str("connect", a, 0), to_str("connect", a, 1)});
        dump += "connect(\"" + edges.back().first + "\",\"" + edges.back().second + "\")\n";
        return Value::nil();
    });
    lua.set_function("set_output", [&](const std::vector<Value>& a) {
        output_vertex = to_str("set_output", a, 0);
        dump += "set_output(\"" + output_vertex + "\")\n";
        return Value::nil();
    });

    std::string lerr;
    if (!lua.run(contents, &lerr)) return fail("Could not execute lua code!\n\t" + lerr);   // state.rs:160-164

    s->cs = cs;
    s->bd = bd;
    s->render_sr = render_sr;
    s->output_file = output_file;
    s->output_vertex = output_vertex;

    // samples: keep unchanged (name, file, method) triples, drop removed ones, add new ones (state.rs:202-219)
    for (auto& old : s->cur_samples)
        if (std::find(new_samples.begin(), new_samples.end(), old) == new_samples.end()) bank_remove(s->sb, std::get<0>(old));
    std::vector<std::string> to_exclude;
    std::string first_err;
    for (auto& ns : new_samples) {
        if (std::find(s->cur_samples.begin(), s->cur_samples.end(), ns) != s->cur_samples.end()) continue;
        if (!td_samplebank_add_file(s->sb, std::get<0>(ns).c_str(), join_path(s->wdir, std::get<1>(ns)).c_str(), std::get<2>(ns).c_str())) {
            if (first_err.empty()) first_err = g_error;
            to_exclude.push_back(std::get<0>(ns));
        }
    }
    if (!to_exclude.empty()) {   // do_excluding!: keep what loaded, abort the refresh
        for (auto& n : to_exclude)
            new_samples.erase(std::remove_if(new_samples.begin(), new_samples.end(), [&](const Triple& t) { return std::get<0>(t) == n; }),
                              new_samples.end());
        s->cur_samples = new_samples;
        return fail(first_err);
    }
    s->cur_samples = new_samples;

    // flowws are always reloaded (state.rs:240-250)
    td_flowwbank_reset(s->fb);
    for (auto& m : midis) {
        std::vector<td_event> ev;
        std::string err;
        if (!read_event_file(join_path(s->wdir, m.second), &ev, &err)) return fail(err);
        td_flowwbank_add_events(s->fb, m.first.c_str(), ev.data(), ev.size());
    }
    for (auto& st : streams) td_flowwbank_declare_stream(s->fb, st.c_str());

    // graph rebuild, vertices grouped by type (state.rs:327-457)
    td_graph_reset(s->g);
    auto sample_index = [&](const std::string& smp, const std::string& vname, size_t* out) {
        long i = td_samplebank_get_index(s->sb, smp.c_str());
        if (i < 0) return fail("Could not get sample index for vertex \"" + vname + "\".");
        *out = (size_t)i;
        return 1;
    };
    auto floww_index = [&](const std::string& fl, const std::string& vname, size_t* out) {
        long i = td_flowwbank_get_index(s->fb, fl.c_str());
        if (i < 0) return fail("Could not get floww index for vertex \"" + vname + "\".");
        *out = (size_t)i;
        return 1;
    };
    size_t si = 0, fi = 0;
    for (auto& c : sums) td_graph_add_sum(s->g, c.name.c_str(), c.gain, c.angle);
    for (auto& c : norms) td_graph_add_normalize(s->g, c.name.c_str(), c.gain, c.angle);
    for (auto& c : sampleloops) {
        if (!sample_index(c.sample, c.name, &si)) return 0;
        td_graph_add_sampleloop(s->g, c.name.c_str(), c.gain, c.angle, si);
    }
    for (auto& c : samplemultis) {
        if (!sample_index(c.sample, c.name, &si) || !floww_index(c.floww, c.name, &fi)) return 0;
        td_graph_add_sample_multi(s->g, c.name.c_str(), c.gain, c.angle, si, fi, c.note);
    }
    for (auto& c : samplelerps) {
        if (!sample_index(c.sample, c.name, &si) || !floww_index(c.floww, c.name, &fi)) return 0;
        td_graph_add_sample_lerp(s->g, c.name.c_str(), c.gain, c.angle, si, fi, c.note, c.lerp_len);
    }
    for (auto& c : debugsines) {
        if (!floww_index(c.floww, c.name, &fi)) return 0;
        td_graph_add_debug_sine(s->g, c.name.c_str(), c.gain, c.angle, fi);
    }
    for (auto& c : synths) {
        if (!floww_index(c.floww, c.name, &fi)) return 0;
        if (!td_graph_add_synth(s->g, c.name.c_str(), c.gain, c.angle, fi, c.sq_vel, c.sq_z, c.sq.data(), (int)c.sq.size(), c.tf_vel, c.tf_z,
                                c.tf.data(), (int)c.tf.size(), c.tr_vel, c.tr.data(), (int)c.tr.size()))
            return 0;
    }
    for (auto& c : sampsyns) {   // state.rs:406-426
        if (!floww_index(c.floww, c.name, &fi)) return 0;
        const std::string* path = nullptr;
        for (auto& r : new_resources)
            if (r.first == c.resource) path = &r.second;
        if (!path) return fail("Could not find resource named " + c.resource + "!");   // state.rs:413 panics
        std::ifstream rf(join_path(s->wdir, *path), std::ios::binary);
        if (!rf) return fail("TermDaw: BufferBank: could not open file \"" + *path + "\".");   // bufferbank.rs:26-52
        std::stringstream rs;
        rs << rf.rdbuf();
        const std::string bytes = rs.str();
        if (!td_graph_add_sampsyn(s->g, c.name.c_str(), c.gain, c.angle, fi, c.adsr.data(), (int)c.adsr.size(), bytes.data(), bytes.size()))
            return 0;
    }
    // lv2fxs: the reference only builds them with the optional `lv2` cargo feature (state.rs:427-436,
    // Cargo.toml:9-11, off by default); without it the vertices do not exist and edges naming them fail.
    (void)lv2fxs;
    for (auto& c : adsrs) {
        if (!floww_index(c.floww, c.name, &fi)) return 0;
        if (!td_graph_add_adsr(s->g, c.name.c_str(), c.gain, c.angle, c.wet, fi, c.use_off, c.use_max, c.note, c.conf.data(), (int)c.conf.size()))
            return 0;
    }
    for (auto& c : bandpasses) td_graph_add_bandpass(s->g, c.name.c_str(), c.gain, c.angle, c.wet, c.lo, c.hi, c.pass);
    for (auto& e : edges) td_graph_connect(s->g, e.first.c_str(), e.second.c_str());   // failures only warn (state.rs:459)
    td_graph_set_output(s->g, s->output_vertex.c_str());
    if (!td_graph_check(s->g)) return fail("TermDaw: graph check failed! (" + g_error + ")");
    td_graph_reset_normalize_vertices(s->g);   // state.rs:467
    s->loaded = true;
    return 1;
}

}  // namespace

extern "C" {

td_state* td_state_new(const char* wdir, size_t project_samplerate, size_t buffer_length) {
    td_state* s = new td_state();
    s->wdir = wdir ? wdir : "";
    s->psr = project_samplerate;
    s->bl = buffer_length;
    s->sb = td_samplebank_new(s->psr);
    s->fb = td_flowwbank_new(s->psr, s->bl);
    s->g = td_graph_new(s->bl, s->psr);
    // The front-end's band-pass vertices run the scan kernels UNDER THE GUARD (engine option "band_mode" 2): tolerance class,
    // <= 1e-6 RMS and +-1 LSB against the reference's serial recurrence -- the bound BASELINE's north_star sets for filter
    // paths -- with the bound checked per render: a render whose own estimate of its deviation is over 2e-7 is done again
    // with the exact kernels (engine.h tde::Guard; BASELINE config 4 renders in 0.4 ms instead of 12 ms and is never
    // redone).  td_state_set_option(s, "band_mode", 0) selects the exact kernels outright, 1 the scan without the guard.
    s->g->band_mode = 2;
    // ... and its debug_sine / synth vertices their fast forms UNDER THE SAME GUARD ("sine_mode" 2; config 3's oscillators in 0.09
    // instead of 0.32 ms): k_sine_probe measures, on a sample of every chunk's frames, how far the fast launch's output is from the
    // reference's own arithmetic (glibc's sinf, adsr.rs's divisions: what "sine_mode" 1 renders), the audit carries that to the
    // graph's output beside the band-pass estimate -- a graph can amplify the white part of a 3e-8 difference without bound (a
    // `cut` band-pass that cancels 43 dB, then a Normalize vertex) -- and over 2e-7 the render is done again in mode 1's form.
    // td_state_set_option(s, "sine_mode", 1): glibc's sinf bit for bit outright; 0: the fast forms without the guard.
    s->g->sine_mode = 2;
    return s;
}

// project.toml (config.rs:19-76): only [settings] main / buffer_length / project_samplerate matter here.
td_state* td_state_open(const char* wdir) {
    const std::string dir = wdir ? wdir : "./";
    std::ifstream f(join_path(dir, "project.toml"));
    if (!f) {
        fail("could not open " + join_path(dir, "project.toml"));
        return nullptr;
    }
    std::string line, section, main_file;
    size_t bl = 1024, psr = 44100;
    while (std::getline(f, line)) {
        size_t h = line.find('#');
        if (h != std::string::npos) line.resize(h);
        auto trim = [](std::string x) {
            size_t a = x.find_first_not_of(" \t\r"), b = x.find_last_not_of(" \t\r");
            return a == std::string::npos ? std::string() : x.substr(a, b - a + 1);
        };
        line = trim(line);
        if (line.empty()) continue;
        if (line.front() == '[') { section = trim(line.substr(1, line.find(']') - 1)); continue; }
        size_t eq = line.find('=');
        if (eq == std::string::npos) continue;
        std::string k = trim(line.substr(0, eq)), v = trim(line.substr(eq + 1));
        if (v.size() >= 2 && (v.front() == '"' || v.front() == '\'')) v = v.substr(1, v.size() - 2);
        if (section == "settings") {
            if (k == "main") main_file = v;
            else if (k == "buffer_length") bl = (size_t)strtoull(v.c_str(), nullptr, 10);
            else if (k == "project_samplerate") psr = (size_t)strtoull(v.c_str(), nullptr, 10);
        }
    }
    if (main_file.empty()) {
        fail("project.toml: [settings] main is required (config.rs:48)");
        return nullptr;
    }
    td_state* s = td_state_new(dir.c_str(), psr, bl);
    s->main_file = main_file;
    return s;
}

void td_state_free(td_state* s) {
    if (!s) return;
    td_graph_free(s->g);
    td_flowwbank_free(s->fb);
    td_samplebank_free(s->sb);
    delete s;
}

int td_state_set_option(td_state* s, const char* key, long value) { return td_graph_set_option(s->g, key, value); }

int td_state_refresh_source(td_state* s, const char* lua_source) { return do_refresh(s, lua_source ? lua_source : ""); }

int td_state_refresh(td_state* s) {
    std::ifstream f(join_path(s->wdir, s->main_file));
    if (!f) {
        s->loaded = false;
        return fail("Can't open main lua file!");
    }
    std::stringstream ss;
    ss << f.rdbuf();
    return do_refresh(s, ss.str());
}

int td_state_scan_exact(td_state* s) {
    if (!s->loaded) return fail("State not loaded!");   // check_loaded! ui_workflow.rs:101-109
    return td_graph_normalize_scan(s->g, s->sb, s->fb, s->cs);
}

static int state_render_device(td_state* s) {
    if (!s->loaded) return fail("State not loaded!");
    if (!(s->bd == 8 || s->bd == 16 || s->bd == 24 || s->bd == 32))
        return fail("Bitdepth of " + std::to_string(s->bd) + " not supported: choose bitdepth in {8, 16, 24, 32}.");
    if (s->cs == 0) {
        s->host_pcm.clear();
        s->out_frames = 0;
        return 1;
    }
    const size_t word = s->bd > 16 ? 4 : 2;
    size_t frames = s->cs * s->bl;
    if (s->psr > s->render_sr) {   // state.rs:533-561 (build-defined resampler, parity unpinned)
        frames = td_graph_render_all_resampled(s->g, s->sb, s->fb, s->cs, (int)s->bd, s->psr, s->render_sr);
        if (!frames) return 0;
    } else {
        // the sink only ever reads the integer PCM (state.rs:515-575): a Normalize output vertex keeps no f32 copy of
        // its frames for this render (engine option "output_f32" 0 -- the form bench.py times), whatever the handle's setting
        const bool keep = s->g->output_f32;
        s->g->output_f32 = false;
        const size_t n = td_graph_render_all(s->g, s->sb, s->fb, s->cs, (int)s->bd);
        s->g->output_f32 = keep;
        if (!n) return 0;
    }
    s->out_frames = frames;
    if (!s->host_pcm.resize(frames * 2 * word)) return fail("termdaw_amd: out of page-locked host memory for the PCM read-back");
    return td_graph_read_pcm(s->g, s->host_pcm.data(), s->host_pcm.size());
}

int td_state_render(td_state* s, const char* path_override) {
    if (!state_render_device(s)) return 0;
    std::string err;
    const std::string path = path_override ? path_override : s->output_file;
    // A 60 s render is 11.5 MB of words that ARE the file's bytes (16- / 32-bit): one thread copying them into the page cache
    // took 2.0 of td_state_render's 2.3 ms.  Slices written side by side (pwrite at their own offsets, wav.cpp) make the same file.
    const size_t data_bytes = s->host_pcm.size();
    const unsigned hw = std::thread::hardware_concurrency();
    const int parts = ((s->bd == 16 || s->bd == 32) && data_bytes >= ((size_t)1 << 20)) ? (int)std::max(1u, std::min(8u, hw / 2)) : 1;
    if (parts == 1) {
        if (!tdw::write_wav_int(path.c_str(), s->host_pcm.data(), s->out_frames, 2, s->render_sr, (int)s->bd, &err)) return fail(err);
        return 1;
    }
    {   // (a file of that name is rewritten from its first byte, as File::create + write would leave it)
        std::vector<std::string> errs((size_t)parts);
        std::vector<char> ok((size_t)parts, 0);
        std::vector<std::thread> pool;
        for (int part = 1; part < parts; ++part)
            pool.emplace_back([&, part] {
                ok[(size_t)part] = tdw::write_wav_int_part(path.c_str(), s->host_pcm.data(), s->out_frames, 2, s->render_sr, (int)s->bd, part, parts, &errs[(size_t)part]);
            });
        ok[0] = tdw::write_wav_int_part(path.c_str(), s->host_pcm.data(), s->out_frames, 2, s->render_sr, (int)s->bd, 0, parts, &errs[0]);
        for (auto& t : pool) t.join();
        for (int part = 0; part < parts; ++part)
            if (!ok[(size_t)part]) return fail(errs[(size_t)part]);
    }
    return 1;
}

size_t td_state_render_to_memory(td_state* s, void* out, size_t bytes) {
    if (!out) {
        const size_t word = s->bd > 16 ? 4 : 2;
        size_t frames = s->cs * s->bl;
        if (s->psr > s->render_sr && s->psr) frames = (size_t)(((unsigned __int128)frames * s->render_sr + s->psr - 1) / s->psr);
        return frames * 2 * word;
    }
    if (!state_render_device(s)) return 0;
    if (bytes < s->host_pcm.size()) {
        fail("render_to_memory: buffer too small");
        return 0;
    }
    memcpy(out, s->host_pcm.data(), s->host_pcm.size());
    return s->host_pcm.size();
}

const void* td_state_render_view(td_state* s, size_t* bytes) {
    if (bytes) *bytes = 0;
    if (!state_render_device(s)) return nullptr;
    if (bytes) *bytes = s->host_pcm.size();
    return s->host_pcm.size() ? (const void*)s->host_pcm.data() : (const void*)"";
}

size_t td_state_chunk_count(const td_state* s) { return s->cs; }
size_t td_state_buffer_length(const td_state* s) { return s->bl; }
size_t td_state_project_samplerate(const td_state* s) { return s->psr; }
size_t td_state_render_samplerate(const td_state* s) { return s->render_sr; }
size_t td_state_bitdepth(const td_state* s) { return s->bd; }
const char* td_state_output_file(const td_state* s) { return s->output_file.c_str(); }
td_graph* td_state_graph(td_state* s) { return s->g; }
td_samplebank* td_state_samplebank(td_state* s) { return s->sb; }
td_flowwbank* td_state_flowwbank(td_state* s) { return s->fb; }
const char* td_state_dump_calls(td_state* s) { return s->dump.c_str(); }

}  // extern "C"
