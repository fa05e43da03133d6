// wav.cpp -- see wav.h.  16-bit stereo PCM is the canonical 44-byte-header RIFF file; for more than
// 16 bits hound writes WAVE_FORMAT_EXTENSIBLE -- that header variant is emitted here too (the data
// chunk is what parity tests compare; the header is sanity-checked separately).
#include "wav.h"

#include <fcntl.h>
#include <unistd.h>
#include <algorithm>

#include <stdio.h>
#include <string.h>

namespace tdw {

static uint32_t rd16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
static uint32_t rd32(const uint8_t* p) { return rd16(p) | (rd16(p + 2) << 16); }

bool read_wav(const char* path, WavData* out, std::string* err) {
    FILE* f = fopen(path, "rb");
    if (!f) {
        *err = std::string("TermDaw: SampleBank: could not open file \"") + path + "\".";
        return false;
    }
    std::vector<uint8_t> buf;
    uint8_t tmp[1 << 16];
    size_t n;
    while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    fclose(f);
    if (buf.size() < 12 || memcmp(buf.data(), "RIFF", 4) != 0 || memcmp(buf.data() + 8, "WAVE", 4) != 0) {
        *err = std::string("TermDaw: SampleBank: \"") + path + "\" is not a RIFF/WAVE file.";
        return false;
    }
    size_t pos = 12;
    uint32_t fmt = 0, ch = 0, rate = 0, bits = 0, align = 0;
    bool have_fmt = false;
    while (pos + 8 <= buf.size()) {
        const uint32_t sz = rd32(&buf[pos + 4]);
        const uint8_t* body = &buf[pos + 8];
        if (memcmp(&buf[pos], "fmt ", 4) == 0 && pos + 8 + 16 <= buf.size()) {
            fmt = rd16(body);
            ch = rd16(body + 2);
            rate = rd32(body + 4);
            align = rd16(body + 12);
            bits = rd16(body + 14);
            if (fmt == 0xFFFE && sz >= 26 && pos + 8 + 26 <= buf.size()) fmt = rd16(body + 24);
            have_fmt = true;
        } else if (memcmp(&buf[pos], "data", 4) == 0) {
            if (!have_fmt || ch == 0) {
                *err = "TermDaw: SampleBank: malformed WAV (no fmt chunk).";
                return false;
            }
            size_t end = pos + 8 + (size_t)sz;
            if (end > buf.size()) end = buf.size();
            const bool enc_ok = (fmt == 3 && bits == 32) || (fmt == 1 && (bits == 8 || bits == 16 || bits == 24 || bits == 32));
            if (!enc_ok) {
                *err = "TermDaw: SampleBank: unsupported WAV encoding.";
                return false;
            }
            // sample stride: the container size from block_align when it is sane, never less than the sample itself
            size_t bps = bits / 8;
            if (align && align / ch > bps && align / ch <= 8) bps = align / ch;
            out->linear.clear();
            for (size_t o = pos + 8; o + bps <= end; o += bps) {
                const uint8_t* p = &buf[o];
                if (fmt == 3 && bits == 32) {
                    float v;
                    memcpy(&v, p, 4);
                    out->linear.push_back(v);
                } else if (fmt == 1 && bits == 8) {
                    out->linear.push_back((float)((int)p[0] - 128));
                } else if (fmt == 1 && bits == 16) {
                    out->linear.push_back((float)(int16_t)rd16(p));
                } else if (fmt == 1 && bits == 24) {
                    int32_t v = (int32_t)(rd16(p) | ((uint32_t)p[2] << 16));
                    if (v & 0x800000) v |= ~0xFFFFFF;
                    out->linear.push_back((float)v);
                } else if (fmt == 1 && bits == 32) {
                    out->linear.push_back((float)(int32_t)rd32(p));
                } else {
                    *err = "TermDaw: SampleBank: unsupported WAV encoding.";
                    return false;
                }
            }
            out->channels = (int)ch;
            out->sample_rate = rate;
            out->bits = bits;
            out->is_float = fmt == 3;
            return true;
        }
        pos += 8 + (size_t)sz + (sz & 1);
    }
    *err = "TermDaw: SampleBank: WAV without data chunk.";
    return false;
}

bool read_wav_raw(const char* path, WavRaw* out, std::string* err) {
    FILE* f = fopen(path, "rb");
    if (!f) {
        *err = std::string("TermDaw: SampleBank: could not open file \"") + path + "\".";
        return false;
    }
    std::vector<uint8_t> buf;
    uint8_t tmp[1 << 16];
    size_t n;
    while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    fclose(f);
    if (buf.size() < 12 || memcmp(buf.data(), "RIFF", 4) != 0 || memcmp(buf.data() + 8, "WAVE", 4) != 0) {
        *err = std::string("TermDaw: SampleBank: \"") + path + "\" is not a RIFF/WAVE file.";
        return false;
    }
    size_t pos = 12;
    uint32_t fmt = 0, ch = 0, rate = 0, bits = 0;
    bool have_fmt = false;
    while (pos + 8 <= buf.size()) {
        const uint32_t sz = rd32(&buf[pos + 4]);
        const uint8_t* body = &buf[pos + 8];
        if (memcmp(&buf[pos], "fmt ", 4) == 0 && pos + 8 + 16 <= buf.size()) {
            fmt = rd16(body);
            ch = rd16(body + 2);
            rate = rd32(body + 4);
            bits = rd16(body + 14);
            if (fmt == 0xFFFE && sz >= 26 && pos + 8 + 26 <= buf.size()) fmt = rd16(body + 24);
            have_fmt = true;
        } else if (memcmp(&buf[pos], "data", 4) == 0) {
            if (!have_fmt || ch == 0) {
                *err = "TermDaw: SampleBank: malformed WAV (no fmt chunk).";
                return false;
            }
            const bool ok = (fmt == 3 && bits == 32) || (fmt == 1 && (bits == 8 || bits == 16 || bits == 24 || bits == 32));
            if (!ok) {
                *err = "TermDaw: SampleBank: unsupported WAV encoding.";
                return false;
            }
            size_t end = pos + 8 + (size_t)sz;
            if (end > buf.size()) end = buf.size();
            const size_t bps = bits / 8;
            out->n_values = (end - (pos + 8)) / bps;
            out->bytes.assign(buf.begin() + (long)(pos + 8), buf.begin() + (long)(pos + 8 + out->n_values * bps));
            out->channels = (int)ch;
            out->sample_rate = rate;
            out->bits = bits;
            out->is_float = fmt == 3;
            return true;
        }
        pos += 8 + (size_t)sz + (sz & 1);
    }
    *err = "TermDaw: SampleBank: WAV without data chunk.";
    return false;
}

static void put16(std::vector<uint8_t>& b, uint32_t v) { b.push_back(v & 0xFF); b.push_back((v >> 8) & 0xFF); }
static void put32(std::vector<uint8_t>& b, uint32_t v) { put16(b, v & 0xFFFF); put16(b, v >> 16); }

static std::vector<uint8_t> wav_header(size_t frames, int channels, size_t sample_rate, int bits) {
    const size_t bps = (size_t)bits / 8;
    const size_t data_bytes = frames * (size_t)channels * bps;
    std::vector<uint8_t> h;
    const bool ext = bits > 16;
    h.insert(h.end(), {'R', 'I', 'F', 'F'});
    put32(h, (uint32_t)(4 + (8 + (ext ? 40 : 16)) + 8 + data_bytes + (data_bytes & 1)));
    h.insert(h.end(), {'W', 'A', 'V', 'E', 'f', 'm', 't', ' '});
    put32(h, ext ? 40 : 16);
    put16(h, ext ? 0xFFFE : 1);
    put16(h, (uint32_t)channels);
    put32(h, (uint32_t)sample_rate);
    put32(h, (uint32_t)(sample_rate * channels * bps));
    put16(h, (uint32_t)(channels * bps));
    put16(h, (uint32_t)bits);
    if (ext) {
        put16(h, 22);
        put16(h, (uint32_t)bits);
        put32(h, channels == 2 ? 3u : (channels == 1 ? 4u : 0u));   // speaker mask
        static const uint8_t pcm_guid[16] = {0x01, 0x00, 0x00, 0x00, 0x00, 0x00, 0x10, 0x00,
                                             0x80, 0x00, 0x00, 0xAA, 0x00, 0x38, 0x9B, 0x71};
        h.insert(h.end(), pcm_guid, pcm_guid + 16);
    }
    h.insert(h.end(), {'d', 'a', 't', 'a'});
    put32(h, (uint32_t)data_bytes);
    return h;
}

// One of `parts` equal slices of the file write_wav_int would write (bits 16 / 32 only: the words are the file's bytes), by
// pwrite at its own offset -- several threads write one file side by side; part 0 also writes the header and sets the
// file's length (a longer file of that name is cut, a slice written before that stays).
bool write_wav_int_part(const char* path, const void* words, size_t frames, int channels, size_t sample_rate, int bits,
                        int part, int parts, std::string* err) {
    if (!(bits == 16 || bits == 32) || parts < 1 || part < 0 || part >= parts) {
        *err = "write_wav_int_part: 16- and 32-bit files only";
        return false;
    }
    const std::vector<uint8_t> h = wav_header(frames, channels, sample_rate, bits);
    const size_t data_bytes = frames * (size_t)channels * ((size_t)bits / 8);
    const int fd = open(path, O_WRONLY | O_CREAT, 0644);
    if (fd < 0) {
        *err = std::string("could not create \"") + path + "\"";
        return false;
    }
    bool ok = true;
    auto put = [&](const void* p, size_t n, size_t at) {
        const uint8_t* b = (const uint8_t*)p;
        while (n && ok) {
            const ssize_t w = pwrite(fd, b, n, (off_t)at);
            if (w <= 0) { ok = false; break; }
            b += w; at += (size_t)w; n -= (size_t)w;
        }
    };
    if (part == 0) {
        ok = ftruncate(fd, (off_t)(h.size() + data_bytes + (data_bytes & 1))) == 0;   // (the pad byte of an odd chunk reads 0)
        put(h.data(), h.size(), 0);
    }
    const size_t per = ((data_bytes + (size_t)parts - 1) / (size_t)parts + 4095) & ~(size_t)4095;
    const size_t lo = std::min(data_bytes, per * (size_t)part), hi = std::min(data_bytes, lo + per);
    put((const uint8_t*)words + lo, hi - lo, h.size() + lo);
    ok = (close(fd) == 0) && ok;
    if (!ok) {
        *err = std::string("short write to \"") + path + "\"";
        (void)unlink(path);   // (no file rather than a valid header over a mix of old and new words)
    }
    return ok;
}

bool write_wav_int(const char* path, const void* words, size_t frames, int channels, size_t sample_rate, int bits,
                   std::string* err) {
    const size_t bps = (size_t)bits / 8;
    const size_t n = frames * (size_t)channels;
    const size_t data_bytes = n * bps;
    const std::vector<uint8_t> h = wav_header(frames, channels, sample_rate, bits);
    FILE* f = fopen(path, "wb");
    if (!f) {
        *err = std::string("could not create \"") + path + "\"";
        return false;
    }
    bool ok = fwrite(h.data(), 1, h.size(), f) == h.size();
    if (bits == 16 || bits == 32) {   // little-endian host: the words are the file bytes, written in place
        ok = ok && fwrite(words, 1, data_bytes, f) == data_bytes;
    } else {
        std::vector<uint8_t> body;
        body.reserve(data_bytes + 1);
        if (bits == 8) {
            const int16_t* w = (const int16_t*)words;
            for (size_t i = 0; i < n; ++i) body.push_back((uint8_t)(w[i] + 128));
        } else {
            const int32_t* w = (const int32_t*)words;
            for (size_t i = 0; i < n; ++i) { body.push_back(w[i] & 0xFF); body.push_back((w[i] >> 8) & 0xFF); body.push_back((w[i] >> 16) & 0xFF); }
        }
        ok = ok && fwrite(body.data(), 1, body.size(), f) == body.size();
    }
    if (data_bytes & 1) ok = ok && fputc(0, f) != EOF;
    ok = (fclose(f) == 0) && ok;
    if (!ok) *err = std::string("short write to \"") + path + "\"";
    return ok;
}

}  // namespace tdw
