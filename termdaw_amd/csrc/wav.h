// wav.h -- minimal RIFF/WAVE PCM reader / writer for the two ends of the render path: the decode in
// front of SampleBank::add (hound::WavReader, sample.rs:231-274) and the integer sink behind
// State::render (hound::WavWriter, state.rs:508-532).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

namespace tdw {

struct WavData {
    std::vector<float> linear;   // interleaved; integer PCM as its integer value cast to f32, float PCM as is
    int channels = 0;
    size_t sample_rate = 0;
    size_t bits = 0;
    bool is_float = false;
};

bool read_wav(const char* path, WavData* out, std::string* err);

// Header parse only: the PCM words stay raw (little-endian, as in the file) for the device-side decode.
struct WavRaw {
    std::vector<uint8_t> bytes;   // data chunk, truncated to whole samples
    size_t n_values = 0;          // bytes / bytes-per-sample
    int channels = 0;
    size_t sample_rate = 0;
    size_t bits = 0;
    bool is_float = false;
};
bool read_wav_raw(const char* path, WavRaw* out, std::string* err);

// words: frames*channels integers, int16 (bits <= 16) or int32 (bits > 16) as State::render produces them
// (write_16s / write_32s).  bits 8 -> unsigned 8-bit samples, 24 -> packed 3-byte samples.
bool write_wav_int(const char* path, const void* words, size_t frames, int channels, size_t sample_rate, int bits,
                   std::string* err);

// The same file in `parts` slices, written by pwrite at their own offsets (several threads, one file); bits 16 / 32 only.
bool write_wav_int_part(const char* path, const void* words, size_t frames, int channels, size_t sample_rate, int bits,
                        int part, int parts, std::string* err);

}  // namespace tdw
