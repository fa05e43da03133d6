// engine.h -- host-side data model behind the C ABI (include/termdaw_amd.h).
//
// The reference renders block by block with a recursive DFS (graph.rs:98-121, 182-193).  Here the
// graph is *compiled* per chunk of whole blocks: a host pass replays only the cheap, inherently
// sequential bookkeeping of the reference (event cursors floww.rs:70-141, voice lists, f32 envelope
// clocks) into small tables, and every per-sample operation runs in the HIP kernels of kernels.hip,
// one batched launch per (topological level, vertex kind).
#pragma once
#include <string.h>

#include <deque>
#include <memory>
#include <utility>
#include <map>
#include <string>
#include <vector>

#include "../../include/termdaw_amd.h"
#include "kernels.h"

namespace tde {

extern thread_local std::string g_error;
int fail(const std::string& msg);   // sets g_error, returns 0

// ---- Rust cast emulation (saturating, truncating, NaN -> 0) ----
size_t f32_as_usize(float x);

struct SampleEntry {
    float2* d = nullptr;   // interleaved frames in HBM: len frames + the first 15 again (wrap frames for looping readers)
    size_t len = 0;
    // Optional packed form for samples that came from <= 16-bit integer PCM through a per-channel-scale load
    // mode: one word per frame (int16 l | int16 r << 16), the loop + its first 15 frames again (kernels.h, InTerm
    // kind 3), + the two normalisation scales.  (float)int * scale rebuilds the f32 frame bit for bit (it IS how the
    // f32 frame was made) at half the bytes.
    uint32_t* d16 = nullptr;
    float scale_l = 0.0f, scale_r = 0.0f;
};

}  // namespace tde

struct td_samplebank {
    size_t sample_rate = 0;
    std::vector<tde::SampleEntry> samples;
    std::map<std::string, size_t> names;
    size_t max_sr = 0, max_bd = 0;
    int device = 0;
    // load-pipeline scratch (decoded stream, planar l / r, scalars, raw PCM), kept from add to add: a hipMalloc /
    // hipFree pair per temporary and sample was most of what loading a project cost
    unsigned char* tmp = nullptr;
    size_t tmp_cap = 0;
    // Sample storage comes out of a few large slabs instead of one hipMalloc per sample: the tables a sum kernel
    // gathers from then sit in one or two big allocations (2 MB page fragments, a handful of TLB entries for a whole
    // project) instead of dozens of small ones.  Pool 0: f32 entries, pool 1: packed 16-bit entries.
    struct Slab { unsigned char* base = nullptr; size_t cap = 0, used = 0, live = 0; };
    std::vector<Slab> slabs[2];
    void* alloc(int pool, size_t bytes);     // nullptr on failure (g_error set)
    void release(void* p);                   // slab memory or a stand-alone hipMalloc
    void release_all();
};

struct td_flowwbank {
    size_t sr = 0, bl = 0, frame = 0;
    std::vector<std::vector<td_event>> flowws;
    std::vector<size_t> start_indices;
    std::map<std::string, size_t> names;
    std::vector<size_t> stream_list;
    std::vector<uint64_t> versions;   // per floww: process-wide unique stamp, renewed whenever its events change
    static uint64_t next_version();
    size_t frame_of(const td_event& e) const { return tde::f32_as_usize(e.t_sec * (float)sr); }
    void set_start_indices_to_frame(size_t t_frame, bool do_skip);
    void set_time(size_t t);
    void set_time_to_next_block();
};

namespace tde {

enum Kind { K_SUM, K_NORMALIZE, K_SAMPLE_LOOP, K_SAMPLE_MULTI, K_SAMPLE_LERP, K_DEBUG_SINE, K_SYNTH,
            K_SAMPSYN, K_ADSR, K_BAND_PASS, K_COUNT };

struct SineNote { float note, vel; };
struct SynthNote { float note, vel, env_t, rel_t, hz; };   // hz = 440 * 2^((note - 69) / 12), extensions.rs:503
struct Voice3 { float t, vel, rel; };

// Compiled event tables of one event-driven vertex (hit lists / interval tables), resident in a device buffer of their
// own together with the key they were compiled from.  A chunk whose key is byte-identical -- same events, same
// FlowwBank cursor, same chunk shape, same carried vertex state at the chunk start -- reuses the tables without
// replaying a single event (and the carried state jumps to the stored end state).
struct TableCache {
    std::string key;                 // empty: nothing cached
    std::string end_state;           // carried host state after the chunk
    uint8_t* d = nullptr;            // device copy of the tables
    uint8_t* h = nullptr;            // pinned host staging of the last upload
    size_t cap = 0;
    hipEvent_t copied = nullptr;
    bool inflight = false;
    // table layout: offsets into d
    size_t hits_off = 0, istart_off = 0, ivoff_off = 0, voices_off = 0, tile_first_off = 0, tile_order_off = 0;
    uint32_t n_hits = 0, n_int = 0;
    // a probed affine Synth vertex (sine_mode 2): the RAW interval table beside the affine one -- k_sine_probe evaluates the
    // reference's own form, which reads (hz, vel, env_t, rel_t) records
    size_t raw_istart_off = 0, raw_ivoff_off = 0, raw_voices_off = 0, raw_tile_first_off = 0;
    uint32_t raw_n_int = 0;
    size_t probe_v_off = 0;   // ... and k_sine_probe's per-sample voice ranges (ProbeDesc::ranges)
    float hz_max = 0.0f;      // (Synth) the tables' largest |hz|: SynthDesc::small_args
};

struct Vertex {
    Kind kind = K_SUM;
    std::string name;
    float gain = 1, angle = 0, wet = 0;
    // parameters
    size_t sample_index = 0, floww_index = 0;
    bool has_note = false;
    size_t note = 0;
    size_t lerp_len = 0;
    tdk::OscConfD square{}, topflat{}, triangle{};
    tdk::AdsrConfD conf{};
    bool use_off = false, use_max = false, pass = true;
    bool exact_sin = false;   // (debug_sine, synth) the graph's "sine_mode" option at the last set_option / add: glibc's sinf on the device
    bool probe = false;       // (debug_sine, synth; set per chunk by compile_chunk) sine_mode 2: the fast form, its deviation measured by k_sine_probe
    float lgamma = 0, hgamma = 0;
    tdk::WaveTableD wavetable{};   // K_SAMPSYN: table in HBM (owned by the graph)
    // carried host state (what the reference keeps inside VertexExt, extensions.rs:15-80)
    uint64_t loop_t = 0;
    std::deque<std::pair<int64_t, float>> ts;
    int64_t p_off = 0, g_off = 0;
    float p_vel = 0, g_vel = 0;
    uint64_t countdown = 0;
    std::vector<SineNote> sine_notes;
    std::vector<SynthNote> notes;
    Voice3 ap{0, 0, 0}, ag{0, 0, 0};
    // carried device state slot (Normalize / BandPass), index into Graph::dstate
    int state_slot = -1;
    // reset_normalization (extensions.rs:295-299) is kept on the host until the next render consumes it
    bool first_pending = false;         // band-pass: set_time since the vertex was last compiled into a submission (its descriptor's first_override)
    bool has_init_override = false;
    float init_override = 0.0f;
    // Normalize: the carried max is the result of a normalize scan (graph.rs:222-237), so a render is expected to
    // stay below it -> speculative single-pass form (SumDesc mode 3)
    bool peak_known = false;
    std::shared_ptr<TableCache> tables;   // event-driven kinds only (shared_ptr: Vertex stays copyable)
    bool has_input() const {
        return kind == K_SUM || kind == K_NORMALIZE || kind == K_ADSR || kind == K_BAND_PASS;
    }
};

// 32-byte slot: NormState or BandState
union StateSlot {
    tdk::NormState norm;
    tdk::BandState band;
};

struct KernelTime { std::string name; float ms = 0; size_t launches = 0; };

}  // namespace tde

namespace tde {
// std::allocator whose value-less construct() default-initialises: resize() of a byte vector then costs no
// zero fill (every staged byte is written right after being allocated; alignment gaps are never read).
template <class T>
struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = NoInitAlloc<U>; };
    template <class U, class... A>
    void construct(U* p, A&&... a) {
        if constexpr (sizeof...(A) == 0) ::new ((void*)p) U;
        else ::new ((void*)p) U(std::forward<A>(a)...);
    }
};
struct Staging {
    std::vector<uint8_t, NoInitAlloc<uint8_t>> b;
    size_t alloc(size_t n) {
        const size_t end = b.size();
        const size_t o = (end + 15) & ~(size_t)15;
        b.resize(o + n);
        for (size_t i = end; i < o; ++i) b[i] = 0;   // (the upload-skip compares whole arenas)
        return o;
    }
    template <class T>
    size_t put(const std::vector<T>& v) {
        size_t o = alloc(v.size() * sizeof(T) + 16);   // never zero-sized
        if (!v.empty()) memcpy(&b[o], v.data(), v.size() * sizeof(T));
        memset(&b[o + v.size() * sizeof(T)], 0, 16);
        return o;
    }
};

}  // namespace tde

namespace tde {
struct BlockCursor {   // FlowwBank state at the start of one block: frame + start index of every floww
    size_t frame;
    const size_t* start;   // [n], into the chunk's flat cursor table
    size_t n;
};

// Table / descriptor arena of one submission: pinned host mirror + device copy, device-only scratch behind the
// uploaded part.  A graph rendering alone owns one; the graphs of a td_batch share the batch's.
struct Arena {
    uint8_t* h = nullptr;
    uint8_t* d = nullptr;
    size_t cap = 0;
    hipEvent_t copied = nullptr;
    bool inflight = false;
    size_t valid = 0;          // bytes of `h` that `d` currently mirrors
    size_t device_bytes = 0;
    // Single-pass Normalize launches (SumDesc modes 4 / 5) whose k_norm_fix was NOT enqueued behind them (the vertex is the
    // last thing its submission computes): a tile whose bounded wait gives up sets *h_flag (page-locked host memory, d_flag =
    // its device address); settle_arena() looks at the word once the stream has drained and launches these then.
    uint32_t* h_flag = nullptr;
    uint32_t* d_flag = nullptr;
    struct PendingFix { size_t off; int n; uint32_t M, bl; };
    std::vector<PendingFix> pending_fix;
    size_t fix_runs = 0;       // how often that happened (td_graph_norm_fix_runs)
    // While pending_fix is non-empty the arena is on a process-wide list (engine.cpp: note_pending / unlist_arena), so that
    // whoever frees device memory such a fix would read -- a SampleBank's slabs -- can settle it first.
    hipStream_t fix_stream = nullptr;
    int fix_device = 0;
    bool listed = false;
    // Epoch-tagged tile words (ChunkBuild::esync_bytes): where the region lay in the last submission that used epochs, and
    // the last epoch handed out (>= 2 once used; 1 is the tag of zeroed words).  Same place -> everything in it is zero or a
    // word of an earlier epoch, and no memset is needed; anywhere else -> zeroed once.
    size_t esync_at = 0, esync_len = 0;
    uint32_t epoch = 1;
};

// HIP-event timing of launch families (bench hook)
struct ProfCtx {
    unsigned every = 0, count = 0;   // events around the launches of every `every`-th submission
    bool now = false;
    struct EvPair { hipEvent_t a, b; int fam; };
    std::vector<EvPair> pending;
    std::vector<hipEvent_t> free_ev;
    std::vector<KernelTime> last_times;
};

// One kernel launch of a compiled chunk: `n` descriptors of family `fam` at staging offset `off`.
struct Launch {
    int fam;
    size_t off;
    int n;
    uint32_t aux;      // F_SUM / F_ADSR: term mode | 0x100 (wide ok); band families: max segment count; F_NORMFIX: 1 = deferred
                       // (not launched: kept as the arena's pending fix)
    int level;
    uint32_t M, bl;    // frames of the chunk, reference block length
    int is_scan;
};

// What compile_chunk() appends to: the staging arena under construction plus everything the submission needs
// to patch and launch it.  Several graphs (a td_batch) compile into ONE ChunkBuild; same-family launches of
// different graphs are then merged into one grid (blockIdx.y indexes the descriptors).
struct ChunkBuild {
    Staging* st = nullptr;
    size_t scratch_bytes = 0;
    struct Fix { size_t at; size_t off; };
    std::vector<Fix> scratch_fix;   // pointer fields -> device scratch (patched once the upload size is known)
    std::vector<Fix> table_fix;     // pointer fields -> uploaded tables
    struct Zero { size_t off, bytes; };
    std::vector<Zero> zero;         // scratch ranges that need a memset before the launches
    // hand-off words of the in-launch scans (k_band_scan's granules): one region behind the scratch, zeroed by ONE
    // memset per submission
    size_t sync_bytes = 0;
    std::vector<Fix> sync_fix;
    // ... and the tile words of the stand-alone single-pass Normalize launches (SumDesc::sync of k_sum16w / k_norm1): a region
    // of its own behind the zeroed one, NOT zeroed per submission -- the words carry the submission's epoch (submit_chunk)
    size_t esync_bytes = 0;
    std::vector<Fix> esync_fix;
    std::vector<size_t> flag_fix;   // pointer fields -> the arena's host-visible word (SumDesc::host_flag)
    std::vector<Launch> launches;
    size_t n_graphs = 0;
    bool one_grid_sources = true;   // every graph compiled in has engine option "one_grid_sources" set (submit_chunk: k_sources)
    void clear() {
        st->b.clear();
        scratch_bytes = 0;
        scratch_fix.clear();
        table_fix.clear();
        zero.clear();
        sync_bytes = 0;
        sync_fix.clear();
        esync_bytes = 0;
        esync_fix.clear();
        flag_fix.clear();
        launches.clear();
        n_graphs = 0;
        one_grid_sources = true;
    }
};
}  // namespace tde

struct td_batch;
struct td_graph;


namespace tde {
// What compiling a chunk changes on the host side of a project (engine.cpp): taken before a step, put back if it fails.
struct HostSnapshot {
    size_t t = 0, fb_frame = 0;
    std::vector<size_t> fb_start;
    struct V { uint64_t loop_t; bool has_init_override, peak_known, first_pending; float init_override; std::string state; };
    std::vector<V> v;
    void take(const td_graph* g, const td_flowwbank* fb);
    void put(td_graph* g, td_flowwbank* fb) const;
};
// band_mode 2.  A render that holds guarded scan launches (k_band_chain<.., true>) ends in k_band_audit, which leaves its
// estimate of the render's RMS deviation from the reference in word[1] and raises word[0] when it is over the bound.  The
// words are looked at when the graph is drained (td_graph_sync, the read functions, td_batch_sync, or the next render that
// continues from carried state): a raised word means the render is done again, from the state it started in, with the
// exact kernels; afterwards the host side is put back to what stood when the verdict was looked at (what the caller did
// between the render and the drain -- a reset_normalization, a set_time, a FlowwBank rewind for the next render -- does not
// depend on the band-pass arithmetic and must survive).  What "the state it started in" takes: the host side in `snap` (taken before the first chunk compiles);
// the device side -- the carried Normalize / band-pass slots -- in `d_backup`, copied on the stream in front of the
// render unless every reachable vertex with a slot starts afresh anyway (reset_normalization / set_time pending: the
// pipelined fresh renders of the bench loops pay nothing).
struct Guard {
    uint32_t* h_word = nullptr;      // page-locked host memory: [0] over the bound, [1] estimate (f32 bits)
    uint32_t* d_word = nullptr;      // ... its device address
    bool armed = false;              // the last render carried an audit; its verdict has not been looked at
    bool in_redo = false;
    bool chunk_audited = false;      // the chunk compiled last holds guarded launches (an audit launch, or the in-launch verdict)
    // the render to do again
    const td_samplebank* sb = nullptr;
    td_flowwbank* fb = nullptr;
    size_t n_blocks = 0, scan_t0 = 0;
    bool is_scan = false, advance = false, want_pcm = false;
    int bits = 16;
    int post = 0;                    // (unused: the redo puts the host side back to what stood when the verdict was looked at)
    HostSnapshot snap;
    bool have_backup = false;
    void* d_backup = nullptr;        // StateSlot[backup_cap]
    size_t backup_cap = 0;
    size_t redos = 0, audits = 0;    // renders done again / renders that carried an audit
    float last_est = 0.0f, max_est = 0.0f;
};
}  // namespace tde

struct td_graph {
    // graph.rs:12-22
    std::vector<tde::Vertex> vertices;
    std::vector<std::vector<size_t>> edges;   // reverse edges: edges[b] = [a...] in connect() order
    std::map<std::string, size_t> name_map;
    long output_vertex = -1;
    size_t bl = 0, sr = 0, t = 0;

    // ---- device side ----
    int device = 0;
    hipStream_t stream = nullptr;
    bool plan_dirty = true;
    std::vector<size_t> order;                // reachable vertices, topological (inputs first)
    std::vector<int> level;                   // per vertex, -1 = unreachable
    int n_levels = 0;
    std::vector<float*> wavetables;            // device tables of K_SAMPSYN vertices
    std::vector<float2*> pool;                // every edge buffer ever allocated (cap_frames each)
    std::vector<float2*> free_bufs;
    size_t cap_frames = 0;
    std::vector<float2*> vbuf;                // per vertex, buffer for the current chunk
    // carried device state
    std::vector<tde::StateSlot> hstate;       // host mirror
    tde::StateSlot* dstate = nullptr;
    size_t dstate_cap = 0;
    bool state_host_dirty = true, state_dev_dirty = false;
    // per-chunk table arena (pinned host + device); unused while the graph renders as part of a batch
    tde::Arena arena;
    tde::ChunkBuild build;
    td_batch* batch = nullptr;                // set by td_batch_add: the graph shares the batch's stream
    size_t batch_projects = 0;                // (set per render) projects of the submission this graph compiles into; 0 / 1: alone
    bool owns_stream = true;
    // outputs of the last render
    void* d_pcm = nullptr;
    size_t pcm_cap = 0, pcm_bytes = 0;
    bool pcm_borrowed = false;                // d_pcm is a slice of the batch's PCM arena (td_batch_render_to_files), not the graph's own
    float2* d_out_f32 = nullptr;              // owned only for multi-chunk renders
    size_t out_f32_cap = 0;
    const float2* last_out_f32 = nullptr;
    size_t last_frames = 0;
    int last_bits = 16;
    float* d_scalar = nullptr;
    float2* d_resampled = nullptr;            // output of the last td_graph_render_all_resampled
    size_t device_bytes = 0;
    bool fuse_sources = true;                  // inline sample_loop sources into their consumers
    bool packed_samples = true;                // inlined sources read the packed 16-bit sample form when it exists
    bool inline_adsr = true;                   // a one-input, one-consumer Adsr vertex is evaluated by its consumer's sum
    bool output_f32 = true;                    // 0: a Normalize output vertex rendered to PCM keeps no f32 copy of its frames
    bool table_cache = true;                   // event tables: reuse across renders / across identical vertices of a chunk
    bool spec_normalize = true;                // renders after a normalize scan use the speculative single-pass normalize
    bool inline_probe = true;                  // (sine_mode 2) one probed vertex behind the graph's one guarded chain launch: the launch's tiles evaluate the probe's samples themselves (test hook "debug.inline_probe" 0: k_sine_probe as a launch of its own)
    bool one_grid_sources = true;              // a level's source launches (affine Synth, wavetable voice, SampleLerp) and the envelope launch go out as ONE grid (k_sources)
    bool fuse_normalize = true;                // band_mode 1: a Normalize vertex right behind a scan launch is evaluated by that launch (BandScanDesc::norm)
    bool single_pass_normalize = true;         // fresh renders of wide all-loop sums find the running peak inside the sum launch (SumDesc mode 4)
    int norm_debug = 0;                        // (tests) bit 0: every single-pass Normalize tile gives up its wait at once -> k_norm_fix
    bool defer_fix = true;                     // (set per render) this render is one chunk: the output vertex' k_norm_fix may wait for settle()
    float band_live_thr = 1e-9f;               // energy from before the short window / energy inside it below which it is enough
    unsigned band_depth = 100;                 // the guess chains block responses until (1 - gamma)^(256 K) <= e^-band_depth
    unsigned band_medium = 30;                 // medium warm-up = band_medium / gamma frames (guess + an alive window)
    unsigned band_quick = 12;                  // quick warm-up = band_quick / gamma frames, taken with the block-response guess (0: no guess)
    unsigned band_short = 40;                  // short warm-up = band_short / gamma frames
    unsigned band_warmup = 150;                // long warm-up = band_warmup / gamma frames (speed only, never exactness)
    bool band_serial = false;                  // (tests, debug.band_serial) every band-pass vertex on the serial kernel -- otherwise the fallback of cut-offs below 5 Hz and block pulls
    int sine_mode = 1;                         // 1 (a bare td_graph's default, like band_mode 0: the reference's bytes): debug_sine / synth evaluate glibc's sinf operation for operation and adsr.rs's own divisions; 0: the device sine of the tolerance class (<= 3.3e-7 from sinf) and the affine / one-grid Synth forms, unguarded; 2 (the front-end's default): the fast forms under the guard -- k_sine_probe measures their deviation at the vertex, the audit carries it to the output, over the bound the render is done again in mode 1's form
    int band_mode = 0;                         // 0: exact (bit-identical to the reference's serial loop), 1: blocked affine scan
                                               //    (tolerance class, <= 1e-6 RMS; one launch per band-pass vertex),
                                               // 2: the scan under the guard (tde::Guard below): every render estimates its own
                                               //    deviation from the reference and is rendered again with the exact kernels when
                                               //    the estimate is over `band_guard_ppb` -- the front-end's default
    unsigned band_guard_ppb = 200;             // band_mode 2: bound on the estimated RMS deviation, in 1e-9 of full scale (0: every guarded render is redone)
    tde::Guard guard;
    bool band_chain = true;                    // scan mode: a chain of band-pass vertices (linked by stages / Adsr vertices) is ONE launch
    int band_scan_nf = 16;                     // frames per lane of k_band_scan (8 | 16): tile = 256 x that
    int band_scan_debug = 0;                   // (tests) bit 0: every k_band_scan poll times out: predecessors are recomputed
    std::vector<size_t> band_stats_off;        // scratch offsets of the last chunk's k_band_fix counters
    const uint8_t* band_stats_base = nullptr;  // device scratch base those offsets refer to
    size_t max_chunk_frames = (size_t)1 << 24;   // edge-buffer chunk cap (16.7 M frames = 128 MiB per buffer)
    tde::ProfCtx prof;
    tde::Staging staging;               // table / descriptor arena under construction (host)
    std::vector<tde::BlockCursor> cursor;   // per-block FlowwBank cursors of the chunk being compiled
    std::vector<size_t> cursor_starts;
    double host_ms[4] = {0, 0, 0, 0};   // host time per run_chunk phase: compile, descriptors, upload, launches
    size_t host_chunks = 0;
    tde::HostSnapshot snapshot;         // host state at the start of the chunk being compiled (restored if it fails)
};

// Many independent projects rendered together (BASELINE config 5: the body of State::render's loop,
// state.rs:563-575, for every project of a GPU's share): the graphs compile into one arena and same-family
// launches of different projects share one grid.
struct td_batch {
    int device = 0;
    hipStream_t stream = nullptr;
    std::vector<td_graph*> graphs;
    std::vector<const td_samplebank*> sbs;
    std::vector<td_flowwbank*> fbs;
    tde::Arena arena;
    tde::Staging staging;
    tde::ChunkBuild build;
    tde::ProfCtx prof;
    float* d_peaks = nullptr;            // [size] per-project peak scratch + [size] source pointers behind it
    float* d_table = nullptr;            // td_batch_exchange_peaks: the job's peak table, per_rank x world floats (device)
    float* h_table = nullptr;            // ... its page-locked host mirror (a td_comm of the host kind reduces there)
    size_t table_cap = 0, table_n = 0;
    size_t peaks_cap = 0;
    std::vector<const float*> peak_src;  // the source-pointer table as the device holds it (td_batch_peak_table_device)
    double host_ms[4] = {0, 0, 0, 0};
    size_t host_steps = 0;
    // td_batch_render_to_files: the PCM of every project in page-locked host memory, filled by a copy stream while later
    // projects render; events: a group's render done / a project's copy done (+ timed pairs for the report)
    hipStream_t copy_stream = nullptr;
    uint8_t* host_pcm = nullptr;
    size_t host_pcm_cap = 0;
    uint8_t* d_pcm_arena = nullptr;      // device PCM of every project, in project order, slices laid out like host_pcm: a
    size_t d_pcm_arena_cap = 0;          // group's PCM is ONE contiguous device -> host copy
    std::vector<size_t> host_pcm_off, host_pcm_bytes;
    std::vector<hipEvent_t> ev_pool;
    hipEvent_t ev_mark[2] = {nullptr, nullptr};   // td_batch_mark
    bool mark_set[2] = {false, false};
};

namespace tde {
int resample_device(const float2* in, size_t len, size_t from, size_t to, float2** out, size_t* nout, hipStream_t st);
int graph_render_chunks(td_graph* g, const td_samplebank* sb, td_flowwbank* fb, size_t n_blocks, bool is_scan,
                        int bits, bool advance_graph_time, size_t scan_t0, bool want_pcm);
}
