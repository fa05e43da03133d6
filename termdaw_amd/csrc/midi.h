// midi.h -- Standard MIDI File -> event list for FlowwBank::add_floww (floww.rs:40-48).
//
// The reference delegates this to floww 0.1.10 (`read_floww_from_midi`), which is not vendored, so the
// mapping below is this engine's own, fixed here (parity unpinned vs the crate; the FlowwBank cursor logic
// that consumes the list IS the reference's, floww.rs:70-141):
//   * SMF formats 0, 1 and 2; all tracks and all channels are merged into one list, ordered by absolute
//     tick (ties: track order, then file order).
//   * ticks -> seconds through the tempo map (meta 0x51, default 500 000 us per quarter note; tempo events
//     of every track apply globally) for PPQ division, or 1 / (fps * ticks_per_frame) for SMPTE division
//     (29 means 29.97 fps); computed in f64, rounded once to f32.
//   * note-on with velocity v > 0 -> (t, note, v / 127.0f); note-on with velocity 0 and note-off ->
//     (t, note, 0.0f).  Everything else (controllers, pitch bend, sysex, other meta) is skipped.
#pragma once
#include <string>
#include <vector>

#include "../../include/termdaw_amd.h"

namespace tde {
bool parse_midi(const unsigned char* bytes, size_t n, std::vector<td_event>* out, std::string* err);
bool read_midi_file(const std::string& path, std::vector<td_event>* out, std::string* err);
}  // namespace tde
