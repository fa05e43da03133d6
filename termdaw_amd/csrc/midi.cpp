#include "midi.h"

#include <algorithm>
#include <cstdint>
#include <fstream>
#include <iterator>

namespace tde {
namespace {

struct Reader {
    const unsigned char* p;
    size_t n, i = 0;
    bool ok = true;
    uint32_t u8() { if (i >= n) { ok = false; return 0; } return p[i++]; }
    uint32_t be16() { uint32_t a = u8(); return (a << 8) | u8(); }
    uint32_t be32() { uint32_t a = be16(); return (a << 16) | be16(); }
    uint32_t vlq() {   // variable-length quantity, at most 4 bytes
        uint32_t v = 0;
        for (int k = 0; k < 4; ++k) {
            const uint32_t b = u8();
            v = (v << 7) | (b & 0x7Fu);
            if (!(b & 0x80u)) return v;
        }
        ok = false;
        return v;
    }
    void skip(size_t k) { if (k > n - i) { ok = false; i = n; } else i += k; }
};

struct RawNote { uint64_t tick; uint32_t order; float note, vel; };
struct Tempo { uint64_t tick; uint32_t order; uint32_t us_per_quarter; };

}  // namespace

bool parse_midi(const unsigned char* bytes, size_t n, std::vector<td_event>* out, std::string* err) {
    Reader r{bytes, n};
    if (n < 14 || bytes[0] != 'M' || bytes[1] != 'T' || bytes[2] != 'h' || bytes[3] != 'd') {
        *err = "not a Standard MIDI File (no MThd)";
        return false;
    }
    r.i = 4;
    const uint32_t hlen = r.be32();
    const uint32_t format = r.be16(), ntrks = r.be16(), division = r.be16();
    if (!r.ok || hlen < 6 || format > 2) { *err = "bad MThd chunk"; return false; }
    r.skip(hlen - 6);
    if ((division & 0x8000u) == 0 && division == 0) { *err = "zero ticks per quarter note"; return false; }

    std::vector<RawNote> notes;
    std::vector<Tempo> tempi;
    uint32_t order = 0;
    uint32_t tracks_seen = 0;
    while (r.ok && r.i + 8 <= n && tracks_seen < ntrks) {
        const bool is_track = bytes[r.i] == 'M' && bytes[r.i + 1] == 'T' && bytes[r.i + 2] == 'r' && bytes[r.i + 3] == 'k';
        r.i += 4;
        const uint32_t len = r.be32();
        if (!r.ok || len > n - r.i) { *err = "truncated chunk"; return false; }
        if (!is_track) { r.skip(len); continue; }   // alien chunks are skipped, as the SMF spec asks
        ++tracks_seen;
        Reader t{bytes + r.i, len};
        r.skip(len);
        uint64_t tick = 0;
        uint32_t status = 0;
        while (t.ok && t.i < t.n) {
            tick += t.vlq();
            uint32_t b = t.u8();
            if (!t.ok) break;
            if (b == 0xFFu) {   // meta
                const uint32_t type = t.u8();
                const uint32_t mlen = t.vlq();
                if (!t.ok || mlen > t.n - t.i) { t.ok = false; break; }
                if (type == 0x51u && mlen == 3) {
                    const uint32_t us = ((uint32_t)t.p[t.i] << 16) | ((uint32_t)t.p[t.i + 1] << 8) | t.p[t.i + 2];
                    tempi.push_back({tick, order++, us});
                }
                t.skip(mlen);
                if (type == 0x2Fu) break;   // end of track
                continue;
            }
            if (b == 0xF0u || b == 0xF7u) {   // sysex
                const uint32_t slen = t.vlq();
                t.skip(slen);
                status = 0;
                continue;
            }
            uint32_t d1;
            if (b & 0x80u) { status = b; d1 = t.u8(); }
            else {   // running status
                if (!status) { t.ok = false; break; }
                d1 = b;
            }
            const uint32_t hi = status & 0xF0u;
            if (hi == 0xC0u || hi == 0xD0u) continue;   // one data byte
            if (hi < 0x80u || hi == 0xF0u) { t.ok = false; break; }
            const uint32_t d2 = t.u8();
            if (!t.ok) break;
            if (hi == 0x90u && d2 > 0) notes.push_back({tick, order++, (float)(d1 & 0x7Fu), (float)(d2 & 0x7Fu) / 127.0f});
            else if (hi == 0x80u || hi == 0x90u) notes.push_back({tick, order++, (float)(d1 & 0x7Fu), 0.0f});
        }
        if (!t.ok) { *err = "malformed track data"; return false; }
    }
    if (tracks_seen == 0) { *err = "no MTrk chunk"; return false; }

    std::stable_sort(notes.begin(), notes.end(), [](const RawNote& a, const RawNote& b) { return a.tick < b.tick; });
    std::stable_sort(tempi.begin(), tempi.end(), [](const Tempo& a, const Tempo& b) { return a.tick < b.tick; });

    out->clear();
    out->reserve(notes.size());
    if (division & 0x8000u) {
        const int fps_code = -(int)(int8_t)(division >> 8);
        const double fps = fps_code == 29 ? 30000.0 / 1001.0 : (double)fps_code;
        const double tpf = (double)(division & 0xFFu);
        if (!(fps > 0.0) || !(tpf > 0.0)) { *err = "bad SMPTE division"; return false; }
        for (const RawNote& e : notes) out->push_back({(float)((double)e.tick / (fps * tpf)), e.note, e.vel});
        return true;
    }
    const double ppq = (double)division;
    size_t ti = 0;
    uint64_t seg_tick = 0;         // start of the current tempo segment
    double seg_sec = 0.0;          // ... in seconds
    double us_per_q = 500000.0;
    for (const RawNote& e : notes) {
        while (ti < tempi.size() && tempi[ti].tick <= e.tick) {
            seg_sec += (double)(tempi[ti].tick - seg_tick) * us_per_q / ppq * 1e-6;
            seg_tick = tempi[ti].tick;
            us_per_q = (double)tempi[ti].us_per_quarter;
            ++ti;
        }
        const double sec = seg_sec + (double)(e.tick - seg_tick) * us_per_q / ppq * 1e-6;
        out->push_back({(float)sec, e.note, e.vel});
    }
    return true;
}

bool read_midi_file(const std::string& path, std::vector<td_event>* out, std::string* err) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { *err = "cannot open file"; return false; }
    std::vector<unsigned char> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    return parse_midi(bytes.data(), bytes.size(), out, err);
}

}  // namespace tde
