// compile.h -- what the host compiler (compile.cpp) and the device side of the engine (engine.cpp) share.
#pragma once
#include <chrono>
#include <string>
#include <vector>

#include "engine.h"

namespace tde {

enum Family { F_LOOP, F_MULTI, F_LERP, F_SINE, F_SYNTH, F_SAMPSYN, F_ENV, F_PROBE /* k_sine_probe: behind the sine kinds' launches of its level */, F_SUM, F_SCALE, F_NORMFIX, F_ADSR, F_BAND, F_BAND_SPEC, F_BAND_FIX, F_BAND_FILL, F_BAND_SCAN, F_QUANT, F_AUDIT,
              F_SOURCES /* (no descriptors of its own: several of the families above as ONE grid, submit_chunk) */, F_COUNT };
extern const char* const kFamilyName[F_COUNT];

// ---- compile.cpp ----
void save_state(const Vertex& v, std::string& out);      // the carried host state of an event-driven vertex, as bytes
void load_state(Vertex& v, const std::string& in);
void build_plan(td_graph* g);                            // reachable vertices in topological order, levels (graph.rs:98-121: what the DFS reaches)
// Steps 1 and 2 for ONE graph, appended to `cb` (which several graphs of a batch may share): event tables, descriptors, launches
int compile_chunk(td_graph* g, const td_samplebank* sb, const td_flowwbank* fb, const std::vector<BlockCursor>& cur, uint64_t t0,
                  bool is_scan, void* pcm_dst, int qmode, float amplitude, ChunkBuild& cb);
size_t desc_size(int fam);
bool is_band_family(int fam);
double ms_between(std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b);

// ---- engine.cpp: the device memory the compiler hands out addresses of ----
int upload_tables(td_graph* g, TableCache& tc, const Staging& tmp);   // a vertex' event tables -> its device buffer (queued on the graph's stream)
int ensure_buffers(td_graph* g, size_t frames);                      // the edge-buffer pool holds buffers of >= frames frames; all of them free
float2* take_buffer(td_graph* g);                                    // one edge buffer (nullptr: out of device memory)

}  // namespace tde
