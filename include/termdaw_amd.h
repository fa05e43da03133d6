/* termdaw_amd.h -- C ABI of the MI355X-native offline render engine for termdaw audio graphs.
 *
 * This is the drop-in boundary for termdaw's per-block vertex/graph render path.  The reference
 * (Rust, /root/reference/src) has no FFI of its own: its "operator API" is the in-process surface
 * State (state.rs) uses to talk to Graph (graph.rs), SampleBank (sample.rs) and FlowwBank
 * (floww.rs).  Every entry point below cites the reference item it replaces; INTEGRATION.md shows
 * the Rust `extern "C"` block a termdaw maintainer would add to bind them.
 *
 * Conventions
 *   - plain C types only; opaque handles are created/destroyed by the library
 *   - functions returning int: 1 = ok/true, 0 = failed/false (mirrors the reference's bool / Result /
 *     Option); td_last_error() returns the message of the last failure on the calling thread
 *   - audio is f32; "frames" are stereo frames; PCM out is interleaved L,R little-endian
 *   - one host thread per graph handle; handles bound to different GPUs are independent.  Handles may be
 *     freed in any order, also with asynchronous renders not yet synced; a SampleBank that gives device
 *     memory back (td_samplebank_free, an entry replaced) first completes what is queued on its GPU, so do
 *     that from the thread that renders the graphs using the bank
 *   - the library never falls back to a CPU implementation: without a usable gfx950 device every
 *     render call fails with an error
 */
#ifndef TERMDAW_AMD_H
#define TERMDAW_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct td_samplebank td_samplebank;
typedef struct td_flowwbank td_flowwbank;
typedef struct td_graph td_graph;
typedef struct td_state td_state;
typedef struct td_batch td_batch;
typedef struct td_comm td_comm;       /* the ranks of a multi-GPU job, for the one collective of the path (td_batch_exchange_peaks) */

/* One event of a floww: the (_, t, note, vel) tuple of the floww crate as used at floww.rs:74-75,
 * 105-116, 131-135 (field 0 is never read on the render path). vel <= 0.001 means note-off. */
typedef struct td_event {
    float t_sec;
    float note;
    float vel;
} td_event;

const char* td_last_error(void);
/* HIP device selection for handles created afterwards on this thread (one process per GPU). */
int td_device_count(void);
int td_set_device(int device);

/* ---- SampleBank (sample.rs:187-348) -------------------------------------------------------- */
td_samplebank* td_samplebank_new(size_t sample_rate);                    /* SampleBank::new   sample.rs:213 */
void td_samplebank_free(td_samplebank* sb);
/* SampleBank::add sample.rs:224-314 (WAV decode -> load mode -> peak normalise -> upload to HBM).
 * method: "" (stereo) | "left" | "right" | "loudest" | "normalize-seperate" | "mix-down".
 * A file whose rate differs from the bank's is resampled (Sample::resample sample.rs:150-175) with this
 * engine's own sinc resampler (rubato is un-vendored: parity unpinned, DESIGN.md "Resampler"). */
int td_samplebank_add_file(td_samplebank* sb, const char* name, const char* path, const char* method);
/* Same pipeline from an already decoded stream (what hound hands to sample.rs:262-274): `linear`
 * holds n interleaved values, integer PCM cast to f32 without scaling, or float PCM as is. */
int td_samplebank_add_decoded(td_samplebank* sb, const char* name, const float* linear, size_t n,
                              int channels, size_t sample_rate, size_t bits, const char* method);
long td_samplebank_get_index(const td_samplebank* sb, const char* name);  /* get_index sample.rs:338; -1 = None */
size_t td_samplebank_sample_len(const td_samplebank* sb, size_t index);   /* Sample::len sample.rs:79 */
/* get_sample sample.rs:342: copies the bank entry back from HBM as planar l / r. */
int td_samplebank_read(const td_samplebank* sb, size_t index, float* l, float* r);
void td_samplebank_get_max_sr_bd(const td_samplebank* sb, size_t* max_sr, size_t* max_bd); /* sample.rs:346 */

/* ---- FlowwBank (floww.rs:6-141) ------------------------------------------------------------- */
td_flowwbank* td_flowwbank_new(size_t sr, size_t bl);                     /* FlowwBank::new floww.rs:19 */
void td_flowwbank_free(td_flowwbank* fb);
void td_flowwbank_reset(td_flowwbank* fb);                                /* floww.rs:23-30 */
/* declare_floww floww.rs:32-38 with the events given directly (add_floww floww.rs:40-48 reads them
 * from MIDI through the un-vendored floww crate). Returns the floww index, or -1. */
long td_flowwbank_add_events(td_flowwbank* fb, const char* name, const td_event* events, size_t n);
/* add_floww floww.rs:40-48: a Standard MIDI File through this library's own reader (the floww crate's
 * read_floww_from_midi is un-vendored; mapping fixed in csrc/midi.h).  Returns the floww index, or -1 with
 * the reference's "Could not read midi file" message in td_last_error(). */
long td_flowwbank_add_midi(td_flowwbank* fb, const char* name, const char* path);
long td_flowwbank_declare_stream(td_flowwbank* fb, const char* name);     /* floww.rs:50-53 */
/* append_streams floww.rs:55-57 with the packets already decoded (`unpacket` is the floww crate's): appends
 * events to the named floww.  Returns its new length, or -1 if the name is unknown. */
long td_flowwbank_append_stream(td_flowwbank* fb, const char* name, const td_event* events, size_t n);
void td_flowwbank_trim_streams(td_flowwbank* fb);                         /* floww.rs:59-64 */
/* copy of floww `index` (at most cap events are written); returns its length */
size_t td_flowwbank_get_events(const td_flowwbank* fb, size_t index, td_event* out, size_t cap);
long td_flowwbank_get_index(const td_flowwbank* fb, const char* name);    /* floww.rs:66-68 */
void td_flowwbank_set_time(td_flowwbank* fb, size_t t);                   /* floww.rs:83-86 */
void td_flowwbank_set_time_to_next_block(td_flowwbank* fb);               /* floww.rs:88-91 */

/* ---- Graph (graph.rs:12-238) + vertex constructors (extensions.rs:83-194, state.rs:341-457) -- */
td_graph* td_graph_new(size_t max_buffer_len, size_t sr);                 /* Graph::new graph.rs:25 */
void td_graph_free(td_graph* g);
void td_graph_reset(td_graph* g);                                         /* graph.rs:39-47 */
/* Graph::add(Vertex::new(bl, gain, angle, wet, VertexExt::<kind>(..)), name).  Argument fix-ups are
 * those of state.rs:341-457: note < 0 -> any note; lerp_len < 0 -> 0; square z clamped to >= 1e-4;
 * adsr arrays of 0, 6 or 9 floats (anything else fails, where the reference panics). */
int td_graph_add_sum(td_graph* g, const char* name, float gain, float angle);
int td_graph_add_normalize(td_graph* g, const char* name, float gain, float angle);
int td_graph_add_sampleloop(td_graph* g, const char* name, float gain, float angle, size_t sample_index);
int td_graph_add_sample_multi(td_graph* g, const char* name, float gain, float angle, size_t sample_index,
                              size_t floww_index, int note);
int td_graph_add_sample_lerp(td_graph* g, const char* name, float gain, float angle, size_t sample_index,
                             size_t floww_index, int note, int lerp_len);
int td_graph_add_debug_sine(td_graph* g, const char* name, float gain, float angle, size_t floww_index);
int td_graph_add_synth(td_graph* g, const char* name, float gain, float angle, size_t floww_index,
                       float square_vel, float square_z, const float* square_adsr, int square_adsr_len,
                       float topflat_vel, float topflat_z, const float* topflat_adsr, int topflat_adsr_len,
                       float triangle_vel, const float* triangle_adsr, int triangle_adsr_len);
/* add_sampsyn (state.rs:406-426, extensions.rs:143-150,532-578).  The wavetable oscillator and its file
 * format live in the un-vendored sampsyn crate; this engine defines its own (DESIGN.md "Wavetable voice":
 * "TDWT" u32 version=1, u32 n_frames, u32 frame_len, f32 table_seconds, n_frames*frame_len f32 LE) --
 * parity with the reference is unpinned.  Unparseable / NULL bytes select the default table. */
int td_graph_add_sampsyn(td_graph* g, const char* name, float gain, float angle, size_t floww_index, const float* adsr,
                         int adsr_len, const void* table_bytes, size_t table_len);
int td_graph_add_adsr(td_graph* g, const char* name, float gain, float angle, float wet, size_t floww_index,
                      int use_off, int use_max, int note, const float* adsr, int adsr_len);
int td_graph_add_bandpass(td_graph* g, const char* name, float gain, float angle, float wet,
                          float cut_off_hz_low, float cut_off_hz_high, int pass);
int td_graph_connect(td_graph* g, const char* a, const char* b);          /* graph.rs:80-96 (+58-78) */
int td_graph_set_output(td_graph* g, const char* vertex);                 /* graph.rs:141-148 */
int td_graph_check(const td_graph* g);                                    /* check_graph graph.rs:150-174 */
void td_graph_set_time(td_graph* g, size_t time);                         /* graph.rs:123-128 */
size_t td_graph_change_time(td_graph* g, size_t delta, int plus);         /* graph.rs:130-135 */
size_t td_graph_get_time(const td_graph* g);                              /* graph.rs:137-139 */
void td_graph_reset_normalize_vertices(td_graph* g);                      /* graph.rs:207-211 */
float td_graph_get_normalization_value(const td_graph* g, const char* name); /* extensions.rs:301-307 */
size_t td_graph_vertex_count(const td_graph* g);

/* Graph::render graph.rs:182-193: renders ONE block at the current playhead, advances the playhead by
 * max_buffer_len, copies the output vertex' block to l / r (max_buffer_len floats each, either may be
 * NULL).  Returns 1 = Some(block), 0 = no output vertex (the reference's None), -1 = failure (td_last_error).
 * Like the reference it does not advance the FlowwBank -- the caller does (state.rs:572). */
int td_graph_render_block(td_graph* g, const td_samplebank* sb, td_flowwbank* fb, float* l, float* r);

/* Graph::true_normalize_scan graph.rs:222-237 over `chunks` blocks (whole timeline on the GPU). */
int td_graph_normalize_scan(td_graph* g, const td_samplebank* sb, td_flowwbank* fb, size_t chunks);

/* The accelerated entry: the body of State::render's loop (state.rs:562-575) for `n_blocks` blocks --
 * n_blocks x { Graph::render; quantise (x*amplitude) as i16|i32 (state.rs:515-532); fb.set_time_to_next_block }
 * then Graph::set_time(0).  The whole timeline is rendered by one kernel per vertex batch; the
 * result stays in HBM.  bits in {8,16,24,32}: <= 16 -> int16 words, otherwise int32 words, exactly the
 * integers the reference hands to hound.  Returns the number of frames rendered (0 on failure). */
size_t td_graph_render_all(td_graph* g, const td_samplebank* sb, td_flowwbank* fb, size_t n_blocks, int bits);
/* The `psr > render_sr` arm of State::render (state.rs:533-561): the same render followed by a down-sample
 * of the whole timeline from psr to render_sr and the quantise.  The reference streams the un-vendored
 * rubato resampler block by block; this engine uses its own documented sinc resampler with rubato's
 * parameter set (DESIGN.md "Resampler") -- parity with the reference is unpinned.  Returns output frames. */
size_t td_graph_render_all_resampled(td_graph* g, const td_samplebank* sb, td_flowwbank* fb, size_t n_blocks, int bits,
                                     size_t psr, size_t render_sr);
/* Device-resident results of the last td_graph_render_all (valid until the next render on g).  After the ASYNC forms the
 * contents are final only once td_graph_sync / td_batch_sync has returned: those settle what a render may have left
 * pending -- the single-pass Normalize's deferred check, the guard's verdict of "band_mode" 2 -- which a plain stream or
 * device synchronisation of the caller's own does not. */
const void* td_graph_output_pcm_device(const td_graph* g);    /* int16|int32 interleaved, frames*2 words */
const float* td_graph_output_f32_device(const td_graph* g);   /* float2 per frame, un-quantised output vertex */
/* D2H copies of the above (pcm: frames*2 words of 2 or 4 bytes; f32: frames*2 floats). */
int td_graph_read_pcm(const td_graph* g, void* out, size_t bytes);
int td_graph_read_f32(const td_graph* g, float* out, size_t n_floats);
/* Per-block absolute peak of the last render's un-quantised output (n_blocks floats) reduced on the
 * device to one float: used for the per-project peak table of the multi-GPU batch. */
float td_graph_output_peak(const td_graph* g);
/* Timing hook for bench.py: enqueue one full render on the graph's stream without the final host
 * synchronisation (td_graph_sync waits).  Same work as td_graph_render_all. */
size_t td_graph_render_all_async(td_graph* g, const td_samplebank* sb, td_flowwbank* fb, size_t n_blocks, int bits);
int td_graph_sync(td_graph* g);
/* How often a single-pass Normalize launch of this graph (or of the batch it belongs to) had to be redone by the check
 * kernel after the fact: a tile of the launch gave up its bounded wait for an earlier tile's running peak -- another
 * process or stream kept part of the grid off the device -- and td_graph_sync / td_batch_sync / the read functions then
 * ran k_norm_fix before returning.  The results are the same either way; nothing in the library traps or waits without
 * bound on work that may not be running.  0 in normal operation. */
size_t td_graph_norm_fix_runs(const td_graph* g);
/* HIP-event timing of the launches of the last render, per kernel family (ms).  names/ms are parallel
 * arrays of capacity cap; returns the number of entries. Enabled by td_graph_set_profiling(g, n): n = 1
 * times every render, n > 1 every n-th render only (the events themselves cost a few us per launch), 0 off. */
void td_graph_set_profiling(td_graph* g, int on);
size_t td_graph_last_kernel_times(const td_graph* g, const char** names, float* ms, size_t* launches, size_t cap);
/* Host-side time spent inside the renders since the last reset, by phase (ms): [0] event compile (cursor /
 * voice bookkeeping -> tables), [1] descriptor build, [2] table upload, [3] kernel launches.  Returns the
 * number of chunks accumulated. */
size_t td_graph_host_times(td_graph* g, double* ms4, int reset);
/* HBM bytes allocated for edge buffers / tables by this graph handle. */
size_t td_graph_device_bytes(const td_graph* g);
/* Device and page-locked memory given back by freed handles stays with the process and is handed out again to the next handle
 * that asks for a block of that size (no reference counterpart; budget: environment TD_ALLOC_CACHE_MB, default 2048, 0 = off;
 * DESIGN.md 3 "Memory").  td_trim_memory gives everything on the free lists back to the driver now; td_cached_memory_bytes
 * says how much is there. */
void td_trim_memory(void);
/* Diagnostic (no reference counterpart): out[i] = the engine's sine of in[i], evaluated on the device -- sine_mode 1: glibc's sinf
 * restated (what debug_sine / synth use under engine option "sine_mode" 1; tests compare it with the host's sinf on all 2^32 bit patterns), 0: the tolerance-class sine.  Host pointers, n values; 1 = done. */
int td_device_sinf(const float* in, float* out, size_t n, int sine_mode);
size_t td_cached_memory_bytes(void);
/* Engine options (no reference counterpart).  SEVEN supported keys:
 * "fuse_sources" 0|1 (default 1: sample_loop sources are gathered inside the consuming sum kernel instead of through an edge
 *   buffer -- same values, same order; 0 = SURVEY 8(d)'s edge-buffer model, `bench.py --no-fuse`);
 * "packed_samples" 0|1 (default 1: inlined sources gather the packed 16-bit form of samples that came from <= 16-bit integer
 *   PCM -- (float)int * scale is how the f32 bank entry was made, so values are identical);
 * "max_chunk_frames" n (edge-buffer chunk cap, default 2^24; smaller values force multi-chunk renders);
 * "output_f32" 0|1 (default 1; 0: a Normalize output vertex rendered to PCM keeps no f32 copy of its frames --
 *   td_graph_read_f32 then fails, the PCM is unchanged);
 * "band_mode" 0|1|2 (default 0 = exact: band_pass_gen, extensions.rs:654-689, bit-identical to the reference's serial
 *   recurrence -- the parity mode.  1 = scan: the same filter as a blocked affine scan, TOLERANCE class: another
 *   realisation of the reference's own f32 rounding noise -- measured 6.3e-8 RMS through 84 band-pass vertices in a row,
 *   +-1 LSB on the PCM; above 1e-6 of the output peak only where a band-pass vertex removes >= 30 dB of its input and a
 *   Normalize vertex brings the rest back up (8 of 18 000 random graphs, at most 3.3e-6).  A state that goes
 *   NaN / infinite stays NaN, as in the reference.  One launch per band-pass vertex, and ONE launch for a whole chain of
 *   `pass` band-pass vertices linked by single-input Sum / Adsr vertices, with the Sum vertex in front and the Normalize
 *   vertex behind.  Cut-offs below ~1.5 Hz keep the exact kernels.  A `pass` vertex' right-channel smoothers reach its
 *   output only as NaN once they are not finite (extensions.rs:685-687): the chain launch does not run them and tracks the
 *   first non-finite right input frame instead.  2 = the scan UNDER THE GUARD -- what a State defaults to: `pass` vertices take
 *   the chain launch wherever its own estimate of its deviation can be carried to the output (any static path, at most one
 *   Normalize vertex on it; no sample loop shorter than 2 048 frames upstream), every other band-pass vertex keeps the exact
 *   kernels, and a render whose estimate is over the bound is rendered again with the exact kernels when the graph is drained:
 *   td_graph_band_guard_stats);
 * "band_guard_ppb" n (default 200: the guard's bound on the estimated RMS deviation of a render, in 1e-9 of full scale -- for
 *   band_mode 2 and sine_mode 2 alike; 0: every audited render is done again);
 * "sine_mode" 0|1|2 (default 1 on a bare td_graph, 2 on a State's graph: debug_sine_gen / synth_gen, extensions.rs:450,501 `f32::sin`.
 *   1 = the oscillators evaluate glibc's sinf -- libm's, what `f32::sin` calls on Linux -- operation for operation in double
 *   precision (glibc 2.28 and later, x86-64 FMA variant; tools/sinf_restate.c agrees with the host's sinf on every finite float) and
 *   the envelopes adsr.rs's own divisions: the two kinds carry the reference's bits like every other kind (50 000 random graphs x 3
 *   renders bit for bit).  0 = the fast forms: a 14-instruction f32 sine (<= 3.3e-7 from sinf), reciprocal envelopes, the
 *   affine / one-grid Synth forms -- ~3e-8 RMS of the vertex' OWN scale (what a graph makes of that -- cancellation, then a Normalize
 *   vertex -- is not bounded), 0.09 instead of 0.32 ms for BASELINE config 3's oscillators.  2 = the fast forms UNDER THE GUARD:
 *   behind every fast launch k_sine_probe evaluates one frame in 256 the reference's way (mode 1's code) and measures the
 *   distance; the audit carries it to the output like a scan launch's estimate (static gains, the Normalize vertex' running
 *   1 / max at the sample's own block); over the bound the render is done again in mode 1's form.  A vertex whose path to the
 *   output the audit cannot follow (two Normalize vertices in a row) renders in mode 1's form from the start).
 * The defaults of td_graph_new and of a State's graph differ in exactly two keys, band_mode (0 | 2) and sine_mode (1 | 2)
 * (tests/test_host_logic.py::test_the_two_default_sets_differ_in_two_keys).
 *
 * Test hooks, "debug.<name>" -- not part of the supported surface: each selects an older or alternative form of a launch, or moves
 * a speculation parameter whose outcome the device verifies; value-neutral by construction, and each pinned by the test named:
 *   debug.norm n (bit 0: every single-pass Normalize tile gives up its wait at once -> the check kernel redoes the vertex;
 *     tests/test_gpu_bench_form.py) / debug.spec_normalize 0|1 / debug.single_pass_normalize 0|1 / debug.fuse_normalize 0|1
 *     (the two-launch Normalize forms; tests/test_gpu_spec_normalize.py, test_gpu_band_scan.py) /
 *   debug.inline_adsr 0|1 (an Adsr vertex materialised instead of read through; tests/test_gpu_parity.py) /
 *   debug.one_grid_sources 0|1 (a level's source launches one by one instead of as k_sources; tests/test_gpu_sources_grid.py) /
 *   debug.inline_probe 0|1 (sine_mode 2: k_sine_probe as a launch of its own instead of inside the guarded chain launch's tiles;
 *     tests/test_gpu_sine_guard.py) /
 *   debug.table_cache 0|1 (event tables recompiled every render; tests/test_gpu_parity.py) /
 *   debug.band_chain 0|1, debug.band_scan_nf 8|16, debug.band_scan n (scan mode: one launch per vertex, frames per lane, bit 0
 *     = every look-back poll times out and predecessors are recomputed; tests/test_gpu_band_scan.py) /
 *   debug.band_serial 0|1 (1: every band-pass vertex on the serial kernel; tests/test_gpu_parity.py) /
 *   debug.band_quick n, debug.band_medium n, debug.band_short n, debug.band_warmup n, debug.band_live_exp n, debug.band_depth n
 *     (the exact band-pass' speculative warm-up lengths in 1 / gamma frames and its liveness thresholds: speed only, the
 *     bit-wise check and repair of k_band_fix keep every result exact; tests/test_gpu_quirks.py, tools/band_*_sweep.py).
 * Gone in round 6, with the measurements that retired them in DESIGN.md 7: "branch_streams" (a HIP stream per independent
 * branch of a level: 1.64 against 1.10 ms), "graph_replay" (HIP-graph capture of an unchanged submission: GPU time unchanged,
 * short projects slower), "band_parallel" 0 (the serial band-pass kernel for everything: 58 against 0.7 ms; it remains the
 * fallback for cut-offs below 5 Hz and block pulls), "band_guess_min" / "band_scan_depth" (constants now). */
int td_graph_set_option(td_graph* g, const char* key, long value);
/* Reads any key td_graph_set_option takes (1 = found).  td_graph_option_key(n): the n-th key, NULL behind the last. */
int td_graph_get_option(const td_graph* g, const char* key, long* value);
const char* td_graph_option_key(size_t n);
/* Counters of the exact parallel band-pass for the last rendered chunk, summed over its band-pass
 * vertices: out[0] repair cascades started (segments whose entry state failed the bit-wise check, incl.
 * re-checks after optimistic repairs), out[1] segments recomputed, out[2] of those cut short by a fixed point
 * under constant input. */
int td_graph_band_stats(const td_graph* g, uint32_t out[3]);
/* The guard of "band_mode" 2 (the scan kernels with a bound that is checked, not assumed): every render that holds scan
 * launches ends in one small launch that adds up the launches' own estimates of how far their output lies from the
 * reference's f32 trajectory (band_pass_gen, extensions.rs:654-689: the rounding of the smoother's state, once per frame --
 * the one thing the scan gives up), carried to the graph's output through the gains behind them (normalize_gen's 1 / max
 * included, extensions.rs:321-329).  An estimate over "band_guard_ppb" x 1e-9 RMS (default 200 = 2e-7; the class's bar is
 * 1e-6) raises a word in page-locked memory; whoever drains the graph next (td_graph_sync, the read functions,
 * td_batch_sync, a render that continues from carried state) then renders the same thing again from the state the render
 * started in, with the exact kernels.  The banks handed to a render must therefore stay alive until the graph has been
 * synced.  out[0] renders that carried an audit, out[1] renders done again, out[2] the last estimate (RMS, full scale 1),
 * out[3] the largest one seen. */
int td_graph_band_guard_stats(const td_graph* g, double out[4]);

/* ---- Batch of independent projects (BASELINE config 5) ---------------------------------------
 * The reference renders one project per process: State::render's loop `for _ in 0..cs { g.render(..); write;
 * fb.set_time_to_next_block() }` (state.rs:563-575).  A batch runs that loop for many independent projects at
 * once on one GPU: every project keeps its own Graph / SampleBank / FlowwBank handles (added with td_batch_add,
 * still usable on their own afterwards), the batch compiles all of them into one table arena and merges
 * same-kind launches of different projects into one grid.  Results are exactly those of td_graph_render_all per
 * project.  Projects shard across GPUs as one batch per process; the only cross-GPU exchange is the per-project
 * peak table below (one all-reduce(max), RCCL). */
td_batch* td_batch_new(void);                              /* on the device selected by td_set_device */
void td_batch_free(td_batch* b);                           /* the projects' handles stay valid */
/* Adds one project; returns its index in the batch or -1.  The graph's device work moves to the batch's stream. */
long td_batch_add(td_batch* b, td_graph* g, const td_samplebank* sb, td_flowwbank* fb);
size_t td_batch_size(const td_batch* b);
/* For every project: Graph::reset_normalize_vertices (state.rs:467) + FlowwBank::set_time(0) -- the state
 * right after State::refresh, from which a fresh render starts. */
void td_batch_rewind(td_batch* b);
/* td_graph_render_all[_async] for every project (each project's PCM / f32 stays readable through its own
 * td_graph_read_pcm / td_graph_output_pcm_device).  Returns n_blocks x the FIRST project's block length (projects of a
 * batch may differ in block length: project i rendered n_blocks x its own), 0 on failure.  A step that fails while its
 * projects are being compiled (an out-of-range sample index, an impossible release note ...) leaves the HOST side of every
 * project where the step found it -- FlowwBank cursor, playhead, loop cursors, carried voices, a pending reset_normalization --
 * so the call can be repeated; what the device has already run cannot be taken back. */
size_t td_batch_render_all(td_batch* b, size_t n_blocks, int bits);
size_t td_batch_render_all_async(td_batch* b, size_t n_blocks, int bits);
int td_batch_sync(td_batch* b);
int td_batch_normalize_scan(td_batch* b, size_t chunks);    /* State::scan_exact (state.rs:473-475) per project */
/* State::render (state.rs:477-577) for every project END TO END -- render, PCM to the host, WAV file (hound::WavWriter,
 * state.rs:508-575: the same header and words td_state_render writes) -- as a pipeline: the projects render in groups of
 * `group` (<= 0: 8), one submission each, queued back to back, into ONE device arena (a project's PCM then lives in its slice
 * of it: td_graph_read_pcm keeps working); a copy stream moves each group's PCM -- one contiguous transfer -- into page-locked
 * host memory as soon as the group has rendered, while later groups render; `writers` host threads write paths[i] as soon as
 * project i's group has landed.  paths NULL (or writers 0): no files, the PCM stays readable through td_batch_host_pcm.
 * Returns 1 when everything is written.  times (may be NULL): 8 doubles -- [0] wall ms of the call, [1] of which buffer /
 * event setup (first call), [2] GPU ms first render start -> last render end, [3] copy-stream ms first copy start -> last
 * copy end, [4] sum of the copies' own ms, [5] PCM bytes, [6] host ms first file opened -> last file closed, [7] host ms
 * spent enqueueing. */
int td_batch_render_to_files(td_batch* b, size_t n_blocks, int bits, size_t render_sr, const char* const* paths, int group,
                             int writers, double* times);
/* Project i's PCM of the last td_batch_render_to_files in the library's page-locked buffer (valid until the next such call
 * or td_batch_free). */
const void* td_batch_host_pcm(const td_batch* b, size_t i, size_t* bytes);
/* Per-project peak after the last render: the output Normalize vertex' running peak (`max`, extensions.rs:323,
 * i.e. the project's pre-normalisation peak) or, for other output kinds, the absolute peak of the output.
 * td_batch_peaks: host copy, one float per project in td_batch_add order.  td_batch_peak_table_device: fills a
 * table of n_total floats in DEVICE memory (caller-owned, e.g. the tensor handed to the all-reduce): project i
 * goes to entry first + i * stride, all other entries are zeroed. */
int td_batch_peaks(td_batch* b, float* out);
int td_batch_peak_table_device(td_batch* b, float* d_table, size_t n_total, size_t first, size_t stride);
/* ---- the job's one collective, behind the C ABI (round 6).  BASELINE config 5: 512 independent projects over the 8 GPUs of a
 * node, one process per GPU, "RCCL over xGMI only for the final peak all-reduce".  The reference renders one project per process
 * (State::render's loop, state.rs:563-575); a batch driver running that loop on every GPU ends with this exchange.
 * td_comm_unique_id: rank 0 makes the 128-byte id (ncclGetUniqueId) and hands it to the other ranks by whatever the host has
 *   (a file, a socket, MPI); td_comm_init: every rank, on its own device (td_set_device first), joins (ncclCommInitRank -- it
 *   returns when all `world` ranks have called it).  RCCL is dlopen'ed (librccl.so.1 as the process already holds it, else from the
 *   ROCm install, else $TD_RCCL_LIB): without it only td_comm_init fails.  td_comm_init_host: the same job over the HOST's own
 *   all-reduce -- `allreduce_max(ctx, table, n)` replaces table[0 .. n) (host memory) by its element-wise maximum over the ranks and
 *   returns 1 -- for hosts that bring MPI, and for tests that put two ranks on one GPU (RCCL refuses that).
 * td_batch_exchange_peaks: the table of per_rank * world floats -- this rank's project i (td_batch_add order) at entry
 *   rank + i * world, zeros elsewhere (td_batch_peak_table_device) -- then ONE ncclAllReduce(ncclMax, ncclFloat32), in place, on
 *   the batch's own stream right behind the renders: no host synchronisation between the last render and the collective.  Returns
 *   when it is enqueued (RCCL kind); td_batch_sync waits for it.  c NULL = a job of one rank (no collective).  Every rank passes
 *   the same per_rank (>= its own project count).  td_batch_peak_table: the table in device memory (valid after td_batch_sync,
 *   until the next exchange); td_batch_read_peak_table: synchronises and copies n entries out.
 * td_comm_backend: "rccl-native" | "host-callback"; td_comm_library: the RCCL library dlopen resolved ("" if none yet). */
int td_comm_unique_id(void* out, size_t bytes);
td_comm* td_comm_init(const void* unique_id, size_t bytes, int rank, int world);
typedef int (*td_allreduce_max_fn)(void* ctx, float* table, size_t n);
td_comm* td_comm_init_host(td_allreduce_max_fn allreduce_max, void* ctx, int rank, int world);
void td_comm_free(td_comm* c);
const char* td_comm_backend(const td_comm* c);
const char* td_comm_library(void);
int td_batch_exchange_peaks(td_batch* b, td_comm* c, size_t per_rank);
const float* td_batch_peak_table(const td_batch* b, size_t* n);
int td_batch_read_peak_table(td_batch* b, float* out, size_t n);
/* bench hooks, as for a graph */
void td_batch_set_profiling(td_batch* b, int on);
size_t td_batch_last_kernel_times(td_batch* b, const char** names, float* ms, size_t* launches, size_t cap);
size_t td_batch_host_times(td_batch* b, double* ms4, int reset);
/* ... and two marks on the batch's stream: td_batch_mark(b, 0) before a run of submissions, td_batch_mark(b, 1) behind it;
 * td_batch_marked_ms: the time between them on the device (HIP events; waits for the second; < 0: not both set). */
int td_batch_mark(td_batch* b, int which);
double td_batch_marked_ms(td_batch* b);

/* ---- Project front-end: State (state.rs:27-578) -------------------------------------------- */
/* State{..} as constructed at main.rs:75-98 (render_sr 48000, bd 16, output "outp.wav"). */
td_state* td_state_new(const char* wdir, size_t project_samplerate, size_t buffer_length);
/* Reads <wdir>/project.toml ([settings] main, buffer_length=1024, project_samplerate=44100; config.rs:19-76). */
td_state* td_state_open(const char* wdir);
void td_state_free(td_state* s);
/* Engine options of the State's graph (the keys of td_graph_set_option; they survive td_state_refresh).  TWO defaults differ
 * from a bare td_graph's -- a State trades the reference's bytes for speed inside the bound BASELINE's north_star sets for float
 * synth / filter paths (1e-6 RMS), a bare graph does not:
 *   "band_mode" 2: band-pass vertices (band_pass_gen, extensions.rs:654-689) as a blocked affine scan UNDER THE GUARD -- every
 *     render estimates its own deviation from the reference's serial recurrence and is rendered again with the exact kernels when
 *     the estimate is over 2e-7 (td_graph_band_guard_stats).  BASELINE config 4 (84 band-pass vertices): 0.39 ms instead of 12.2 ms;
 *     270 000 random-graph renders: none above 1e-6 by the filter arithmetic.  td_state_set_option(s, "band_mode", 0): the exact
 *     kernels outright, bit-identical to the reference's recurrence;
 *   "sine_mode" 0: debug_sine / synth with the tolerance-class device sine (<= 3.3e-7 from libm's sinf per oscillator; <= 1e-6 RMS of
 *     the vertex' scale) and the affine / one-grid Synth forms.  td_state_set_option(s, "sine_mode", 1): glibc's sinf operation for
 *     operation, the reference's bytes.
 * A project without band-pass, debug_sine and synth vertices renders the same bytes either way. */
int td_state_set_option(td_state* s, const char* key, long value);
/* State::refresh state.rs:50-471 on the given Lua source / on <wdir>/<main>. 1 = loaded. */
int td_state_refresh_source(td_state* s, const char* lua_source);
int td_state_refresh(td_state* s);
int td_state_scan_exact(td_state* s);                                     /* state.rs:473-475 */
/* State::render state.rs:477-577: renders cs blocks and writes the integer WAV to output_file
 * (relative to the process working directory, like hound::WavWriter::create at state.rs:514). path_override may be NULL. */
int td_state_render(td_state* s, const char* path_override);
/* Same render, PCM left in memory: copies frames*2 words to out (may be NULL to query the size). */
size_t td_state_render_to_memory(td_state* s, void* out, size_t bytes);
/* The same render, returned as a view of the library's own page-locked read-back buffer (interleaved PCM,
 * *bytes long; valid until the next render or td_state_free): no second copy.  NULL on failure. */
const void* td_state_render_view(td_state* s, size_t* bytes);
size_t td_state_chunk_count(const td_state* s);                            /* cs, state.rs:104 */
size_t td_state_buffer_length(const td_state* s);                          /* config.rs:58-60 */
size_t td_state_project_samplerate(const td_state* s);                     /* config.rs:62-64 */
size_t td_state_render_samplerate(const td_state* s);
size_t td_state_bitdepth(const td_state* s);
const char* td_state_output_file(const td_state* s);
td_graph* td_state_graph(td_state* s);
td_samplebank* td_state_samplebank(td_state* s);
td_flowwbank* td_state_flowwbank(td_state* s);
/* The recorded script calls in call order, one per line, canonical text (host-logic tests). */
const char* td_state_dump_calls(td_state* s);

#ifdef __cplusplus
}
#endif
#endif /* TERMDAW_AMD_H */
