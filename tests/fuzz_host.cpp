// Host-side robustness fuzz (built with -fsanitize=address,undefined by tests/test_host_fuzz.py; CPU only): mutated
// Standard MIDI Files, WAV files and project scripts go through the engine's own readers -- csrc/midi.cpp,
// csrc/wav.cpp, csrc/lua_subset.cpp -- which must accept or reject them cleanly (no crash, no sanitizer report,
// no runaway allocation).  usage: fuzz_host <iterations> <work file> <seed.mid> <seed.wav> <seed.lua>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <random>
#include <string>
#include <vector>
#include "midi.h"
#include "wav.h"
#include "lua_subset.h"

static std::vector<unsigned char> slurp(const char* p) {
    std::vector<unsigned char> v; FILE* f = fopen(p, "rb"); if (!f) return v;
    unsigned char b[4096]; size_t n; while ((n = fread(b, 1, sizeof b, f)) > 0) v.insert(v.end(), b, b + n); fclose(f); return v;
}
int main(int argc, char** argv) {
    std::mt19937 rng(1234);
    std::vector<std::vector<unsigned char>> seeds;
    const int iterations = atoi(argv[1]);
    const char* work = argv[2];
    for (int i = 3; i < argc; ++i) seeds.push_back(slurp(argv[i]));
    size_t ok = 0, bad = 0;
    for (int it = 0; it < iterations; ++it) {
        std::vector<unsigned char> d = seeds[it % seeds.size()];
        const int kind = it % (int)seeds.size();
        int nm = 1 + rng() % 8;
        for (int m = 0; m < nm && !d.empty(); ++m) {
            switch (rng() % 5) {
                case 0: d[rng() % d.size()] = (unsigned char)rng(); break;
                case 1: d.resize(rng() % (d.size() + 1)); break;
                case 2: d.insert(d.begin() + rng() % (d.size() + 1), (unsigned char)rng()); break;
                case 3: { size_t a = rng() % d.size(); d[a] ^= 1u << (rng() % 8); } break;
                case 4: { size_t a = rng() % d.size(), b = rng() % d.size(); std::swap(d[a], d[b]); } break;
            }
        }
        std::string err;
        bool r = false;
        if (kind == 0) { std::vector<td_event> ev; r = tde::parse_midi(d.data(), d.size(), &ev, &err); }
        else if (kind == 1) {
            FILE* f = fopen(work, "wb"); fwrite(d.data(), 1, d.size(), f); fclose(f);
            tdw::WavData w; r = tdw::read_wav(work, &w, &err);
            tdw::WavRaw wr; std::string e2; (void)tdw::read_wav_raw(work, &wr, &e2);
        } else {
            tdl::Interp in;
            in.set_function("f", [](const std::vector<tdl::Value>& a) { return tdl::Value(); });

            r = in.run(std::string(d.begin(), d.end()), &err);
        }
        (r ? ok : bad)++;
    }
    printf("fuzz done: %zu accepted, %zu rejected cleanly\n", ok, bad);
    return 0;
}
