// kernels.h -- descriptor structs and launch wrappers of the gfx950 vertex kernels.
//
// Layout contract (DESIGN.md "Data layout"): every edge buffer is `frames` interleaved stereo frames
// (float2 = {l, r}) in HBM; a kernel launch covers a contiguous run of whole reference blocks
// ("chunk") and is batched over same-kind vertices of one topological level through blockIdx.y, each
// vertex described by one descriptor in a device-resident table.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "adsr_math.h"

namespace tdk {

constexpr int kThreads = 256;       // 4 wave64 per workgroup
constexpr int kTileFrames = 1024;
constexpr uint32_t kBandMaxSegs = 131072;   // k_band_fix keeps its mismatch bitmap in LDS
constexpr uint32_t kBandMaxS = 1024;        // ... so chunks beyond 33 M frames take longer segments (any multiple of 256 works: compile.cpp plan_band picks longer ones for batches)   // frames per workgroup tile: 2 x float4 (2 frames each) per thread

// Vertex epilogue: Sample::apply_angle then Sample::apply_gain (sample.rs:97-114, order fixed at
// extensions.rs:262-263).  Amplitudes are computed on the host with libm; flags carry the skip
// thresholds (|angle| < 0.001, |gain-1| < 0.001).
struct PanGain {
    float l_amp, r_amp, gain;
    uint32_t flags;  // bit0: apply pan, bit1: apply gain
};

// One input term of a summing vertex, in connect() order.
//   kind 0: an edge buffer in HBM.
//   kind 1: an inlined sample_loop source (extensions.rs:331-341 + its own pan/gain epilogue): the consumer
//           gathers sample[(t0 + m) % len] itself, so the source's edge buffer never exists (DESIGN.md
//           "source inlining").  32-bit form: t0 + m < 2^32, modulo by Barrett reduction with
//           magic = floor(2^32 / len).
//   kind 2: the same with 64-bit cursor / length (generic modulo).
//   kind 3: kind 1 over the sample's packed 16-bit form: one 32-bit word per frame (int16 l | int16 r << 16), the
//           loop followed by its own first 15 frames, so that up to 16 consecutive loop frames are dword-aligned
//           16-byte loads behind ONE modulo; the f32 frame is rebuilt as (float)l * scale_l, (float)r * scale_r -- the
//           very expression the load pipeline used to produce the f32 bank entry (sample.rs:270-273 `as f32`,
//           sample.rs:121-129 `* (1.0 / max)`), so the values are bit-identical at half the gather bytes.
//   kind 4: an edge buffer read THROUGH a Sum vertex that has this one input only (a gain / pan stage): the
//           consumer computes `0.0 + x`, pan, gain itself (extensions.rs:310-319, sample.rs:97-114) and the
//           stage is never materialised.
//   kind 5: an edge buffer read through an Adsr vertex that has this one input and one consumer (adsr_gen,
//           extensions.rs:593-651, evaluated per frame by the consumer), optionally followed by a kind-4 stage
//           (magic != 0, `pg` is then the stage's): `len` = device address of the vertex' AdsrVDesc (its input term, its
//           own pan / gain, and `env`: the vertex' per-frame gain `lerp(1.0, adsr_vel, wet)` for this chunk, made once
//           per distinct envelope by k_adsr_env -- the gain depends on the frame and the event tables only, so the 84
//           envelope stages of a deep chain share one).  The consumer's only term (TERMS_ADSR1) or one of several
//           (TERMS_WITH_ADSR).
struct InTerm {
    const float2* p;   // edge buffer (kind 0) or sample frames (kind 1, 2)
    uint64_t len;      // sample length
    uint64_t t0;       // loop cursor at the chunk's first frame
    PanGain pg;        // the source vertex' epilogue
    uint32_t kind;
    uint32_t magic;
    float scale_l, scale_r;   // kind 3
};
enum TermMode : uint32_t { TERMS_MIXED = 0, TERMS_ALL_EDGE = 1, TERMS_ALL_LOOP32 = 2, TERMS_ALL_LOOP16 = 3,
                           TERMS_EDGE_FEW = 4,     // all edge buffers, fewer than 8: no deep prefetch pipeline (and its registers)
                           TERMS_ADSR1 = 5,        // exactly one term, of kind 5
                           TERMS_WITH_ADSR = 6 };  // several terms, at least one of kind 5: summed one term at a time

// Running-peak bookkeeping of normalize_gen (extensions.rs:321-329), carried across chunks / passes.
// `violated` / `ticket` belong to the speculative single-pass normalize (SumDesc mode 3): both are 0 between launches.
struct NormState { float max, scan_max; uint32_t violated, ticket; };

// Block responses of a band-pass vertex' two smoothers, made by the k_sum launch that materialises its input (mode 2)
// and used by k_band_spec as the starting point of its speculative warm-up: for every aligned 256-frame block b and
// chain c (low L, low R, high L, high R)   resp[4 b + c] = sum_i gamma (1 - gamma)^(255 - i) x_c[256 b + i],
// i.e. what the block alone adds to the smoother's state at its end -- in double, so that a few dozen of them chained by
// y <- (1 - gamma)^256 y + resp reproduce the exact-arithmetic state to far below an f32 ulp.
struct BandRespParam {
    double ql[8], qh[8];   // (1 - gamma)^(2^j), j = 0..7, low / high smoother
    double gl, gh;         // the two gammas; 0 = no responses for that smoother (it is constant, or so fast -- gamma >= 0.05,
                           // (1 - gamma)^400 < 2^-30 -- that any starting value has converged long before the window ends)
    double* resp;          // [ceil(frames / 256)][4]
};

// sum_inputs (extensions.rs:310-319), optionally + per-reference-block absolute peak
// (normalize_gen's scan_max, extensions.rs:322 / sample.rs:116-118).
struct SumDesc {
    const InTerm* ins;         // k input terms, in connect() order
    float2* out;
    float* peaks;              // [n_blocks] (mode 1)
    // mode 1: workgroup 0 snapshots the carried normalize state into scratch {max, scan_max} so that pass
    // B can update the state in place without racing its own readers
    const NormState* state;
    float* init_copy;
    float init_max;            // used instead of state->max when use_init (reset_normalization, extensions.rs:295-299)
    uint32_t use_init;
    uint32_t k;
    uint32_t mode;             // 0: Sum vertex (epilogue applied), 1: Normalize pass A (raw sum + peaks),
                               // 2: band-pass input (raw sum + peak of every 256-frame block -> peaks)
                               // 3: Normalize in ONE pass, speculating that no block peak exceeds the carried max (true
                               //    after a normalize scan, graph.rs:222-237, unless carried vertex state makes this
                               //    render louder than the scan was): sum, scale by 1 / max, epilogue, optional quantise
                               //    straight out of registers; block peaks still go to `peaks`, a peak above max sets
                               //    state->violated and k_norm_fix (same descriptors) redoes the vertex the two-pass way
                               // 4: Normalize in ONE pass with the running peak (fresh renders), see `sync` below
                               // 5: the same, for a grid the host expects to be resident at once (k_sum16w, k_norm1, the epilogue
                               //    of k_band_chain).  Same kernel code as mode 4: the wait for earlier tiles is BOUNDED, a tile
                               //    that gives up raises state->violated (and *host_flag) and k_norm_fix redoes the vertex
    uint32_t term_mode;        // TermMode: lets the kernel pick a loop specialised for the term kinds
    uint32_t debug;            // (tests) bit 0: every wait for an earlier tile's granule gives up at once -> violated -> k_norm_fix;
                               // bit 1 (k_sum16w, timing experiments only -- WRONG results): the earlier tiles' words are not read at all
    PanGain pg;
    // mode 2: a second copy of the raw sum, planar within every aligned 4-frame block --
    // {l0 l1 l2 l3}{r0 r1 r2 r3} instead of {l0 r0 l1 r1}{l2 r2 l3 r3} -- the form k_band_spec's warm-up
    // walks (one register = one chain's next four inputs).  Same size and word addresses as `out`.
    float2* out_q4;
    // mode 3: fused quantise when the vertex is the output (as in ScaleDesc); `state` is written by k_norm_fix
    void* pcm;
    float amplitude;
    uint32_t qmode;
    const BandRespParam* rp;   // mode 2: block responses wanted (nullptr: not)
    // mode 4 (k_sum16w only): Normalize in ONE pass with the RUNNING peak -- every tile publishes its maximum as a granule
    // in `sync` ([tiles] words, zeroed before the launch) and reads the earlier tiles'; k_norm_fix behind it as in mode 3
    unsigned long long* sync;
    // modes 4 / 5: a word in page-locked HOST memory (device address), set to 1 by a tile whose bounded wait gave up.  Where
    // the vertex is the last thing its submission computes, the engine does not enqueue k_norm_fix behind the launch: it
    // looks at this word once the stream has drained and launches the fix only then (engine.cpp settle()).  nullptr: none.
    uint32_t* host_flag;
};

// Normalize pass B: running max over the block peaks (`*max = buf_max.max(*max)`), buf.scale(len, 1.0 / max)
// (extensions.rs:323-328), epilogue, optional fused quantise.  Every workgroup derives the running max of
// its own blocks from the peak table (a few KB from L2); the workgroup of the last tile stores the carried
// state.  Scan passes scale by the STALE max and accumulate scan_max instead (quirk Q3).
struct ScaleDesc {
    float2* buf;            // in place
    const float* peaks;     // [n_blocks]
    const float* init_copy; // {max, scan_max} at chunk start
    NormState* state;       // updated by the last tile's workgroup
    void* pcm;              // optional int16/int32 interleaved output
    float amplitude;
    uint32_t qmode;         // 0 none, 1 int16, 2 int32
    PanGain pg;
    uint32_t pcm_only;      // the scaled f32 frames are not written back (nobody reads them: engine option "output_f32" 0)
    uint32_t pad[3];
};

struct QuantDesc {
    const float2* in;
    void* pcm;
    float amplitude;
    uint32_t qmode;
};

// sample_loop_gen (extensions.rs:331-341): out[m] = sample[(t0 + m) % len]
struct LoopDesc {
    const float2* sample;
    float2* out;
    uint64_t len;
    uint64_t t0;
    uint32_t magic;   // floor(2^32 / len) when the 32-bit form applies (len, t0 + frames < 2^32), else 0
    uint32_t pad[3];
    PanGain pg;
};

// sample_multi_gen (extensions.rs:344-381): hits in onset order; a voice with origin o sounds on frames
// [o, o + len) reading sample[m - o].  Origins are chunk-relative and may be negative (carried voices).
struct MultiHit { int64_t origin; float vel; float pad; };
struct MultiDesc {
    const float2* sample;
    float2* out;
    const MultiHit* hits;
    uint64_t len;
    uint32_t n_hits;
    uint32_t pad;
    PanGain pg;
    const uint32_t* tile_first;   // [tiles]: first hit that can still sound at the tile's first frame (origin > frame - len)
};

// sample_lerp_gen (extensions.rs:384-421).  key = frame from which the entry is `primary` (INT64_MIN for
// the two carried entries), origin = frame of sample position 0, fade = frame at which countdown was set
// to lerp_len.
struct LerpHit { int64_t key, origin, fade; float vel; float pad; };
struct LerpDesc {
    const float2* sample;
    float2* out;
    const LerpHit* hits;   // >= 2 entries; [0] = initial ghost, [1] = initial primary
    uint64_t len;
    uint32_t n_hits;
    uint32_t lerp_len;
    PanGain pg;
    const uint32_t* tile_first;   // [tiles]: number of entries with key <= the tile's first frame
};

// Interval tables for the voice-list vertices: the chunk is cut at block starts and event frames into
// intervals inside which the voice list is constant.  istart[i] = first frame (chunk-relative),
// ivoff[i]..ivoff[i+1] = voice records.
struct IntervalTab {
    const uint32_t* istart;
    const uint32_t* ivoff;
    const float4* voices;
    const uint32_t* tile_first;   // [tiles] index of the interval holding each 1024-frame tile's first frame
    uint32_t n_int;
    uint32_t pad;
    // [tiles] a permutation of the tiles, the costliest first (compile.cpp put_intervals: a tile costs a pass of the voice loop per
    // interval it holds frames of, each as long as the interval has voices): the expensive ones start at the head of the launch,
    // the grid's last workgroups are the cheap ones.  (Round 3: the tiles with an interval start inside them first, k_synth
    // 0.198 -> 0.14 ms on BASELINE config 3; round 6: by cost.)  nullptr: identity.
    const uint32_t* tile_order;
};

struct OscConfD { float volume, param; AdsrConfD adsr; };

// debug_sine_gen (extensions.rs:423-457): voice = (hz, vel, -, -)
struct SineDesc {
    IntervalTab tab;
    float2* out;
    uint64_t t0;   // graph time of the chunk's first frame
    uint32_t sr;
    uint32_t exact_sin;   // 1: glibc's sinf, operation for operation (SynthDesc::exact_sin)
    PanGain pg;
};

// synth_gen (extensions.rs:460-529): voice = (hz, vel, env_t at block start, rel_t)
struct SynthDesc {
    IntervalTab tab;
    float2* out;
    uint64_t t0;
    uint32_t sr, bl;
    OscConfD square, topflat, triangle;
    float osc_amp_multiplier;
    // envelope sharing: 1 / 2 = this oscillator's AdsrConf is bit-identical to the square's / topflat's (and that
    // one is enabled), so its envelope value -- a pure function of the conf and the voice's clocks -- is reused
    uint32_t tf_env_src, tr_env_src;
    PanGain pg;
    // 1: the interval table is cut at every envelope breakpoint and a voice is FOUR float4 -- (hz, env_t, 0, 0) and, per
    // oscillator (square, top-flat, triangle), (s1, s2, A, B): envelope x velocity x volume x osc_amp_multiplier x shape scale
    // = A + B ((t - s1) - s2), t = env_t + in-block offset (engine.cpp synth_refine_affine); 0: one float4 (hz, vel, env_t,
    // rel_t) per voice, envelopes evaluated per frame (confs that can reach the `res <= -1.0` escape, zero-length pieces)
    uint32_t affine;
    uint32_t exact_sin;   // 1: the oscillators' sine is glibc's sinf, operation for operation (kernels.hip sin_glibc; engine option "sine_mode"); affine is 0 then    // 1 (host: compile.cpp synth_desc_of): every sine argument of the chunk, (t0 + m) / sr * hz * 2 pi over the tables' largest hz,
    // stays below 2e6 rad -- the affine form's sine then rounds to half turns by adding 1.5 * 2^23 and takes a degree-9 polynomial
    // fitted to the reduced range that bound leaves (sin_small2); beyond: sin_any2
    uint32_t small_args, pad_sa;
};

// sampsyn_gen (extensions.rs:532-578) with this engine's own wavetable oscillator (the sampsyn crate is
// un-vendored; DESIGN.md "Wavetable voice"): voice = (hz, vel, env_t at block start, rel_t)
struct WaveTableD {
    const float4* quads;    // [n_frames][frame_len] of {w[f][i], w[f][i+1 wrapped], w[f+1 clamped][i], w[f+1 clamped][i+1 wrapped]}
    uint32_t n_frames, frame_len;
    float table_seconds;
    uint32_t pad;
};
struct SampsynDesc {
    IntervalTab tab;
    float2* out;
    WaveTableD wt;
    uint32_t sr, bl;
    AdsrConfD adsr;
    float amp_multiplier;
    PanGain pg;
};

// adsr_gen (extensions.rs:593-651): two voice records per interval: primary, ghost = (t_off, vel,
// release_val, skip) -- skip != 0 on the primary record marks a frame the reference leaves untouched
// (the `continue` at extensions.rs:632-635).
struct AdsrVDesc {
    IntervalTab tab;
    const InTerm* ins;
    float2* out;
    uint32_t k;
    uint32_t sr, bl;
    uint32_t use_off, use_max, term_mode;
    float wet;
    AdsrConfD conf;
    PanGain pg;
    // for the run form of the envelope (adsr_run, kernels.hip): 1.0 / (double)x of attack_sec, decay_sec, sustain_sec,
    // release_sec and of (float)sr -- (float)((double)n * rcp) is the IEEE f32 quotient n / x -- and whether every level an
    // attack / decay / sustain piece can return stays above the `res <= -1.0` escape of adsr.rs:62-69,75-86
    double rcp[5];
    uint32_t tame, pad2;
    // the vertex' gain per frame of the chunk (>= frames + 1 floats): written by k_adsr_env, read by consumers of a kind-5 term
    float* env;
    // ... and its mean square over every 512 frames ([ceil(frames / 512)], behind the gains in the same buffer): what the
    // guarded chain launch multiplies its estimate by where the vertex is a link (two scalar loads per stage)
    float* env_tile;
};
void adsr_fill_run_consts(AdsrVDesc* d);   // host: rcp[], tame from conf / sr
void launch_adsr_env(const AdsrVDesc* d, int n_desc, uint32_t frames, hipStream_t s);

// band_pass_gen (extensions.rs:654-689), exact sequential form.
struct BandState { float lprevl, lprevr, hprevl, hprevr; uint32_t first; uint32_t pad[3]; };
struct BandDesc {
    const InTerm* ins;
    float2* out;
    BandState* state;   // carried across chunks
    uint32_t k;
    uint32_t term_mode;
    uint32_t pass;
    float lgamma, hgamma;
    PanGain pg;
    // set_time since the vertex last ran (extensions.rs:196-204 sets BandPass.first): the kernel treats the state's `first`
    // word as set whatever the device copy says -- a descriptor bit instead of a fill kernel per set_time
    uint32_t first_override, pad_fo;
};

// band_pass_gen, exact AND parallel: speculative segments.
//   The recurrence y += gamma * (x - y) is a contraction: two trajectories fed the same input collapse onto
//   each other and, once bit-identical, stay identical.  The chunk is cut into segments of S frames; a
//   quad of lanes (low L, low R, high L, high R) runs each segment after a warm-up of W frames started from
//   a guess, records the state it entered the segment with and the state it left with, and writes the
//   segment's output.  k_band_fix then checks, bit for bit, that every segment entered with exactly the
//   state its predecessor left with -- by induction from the exactly-known first segment this proves the
//   whole output exact -- and recomputes (only) segments where the check fails, keeping seg_start honest
//   (= the entry state the stored output was computed from) so that the check can simply be repeated.  The usual failures are
//   constant or silent input stretches, where f32 trajectories park on different sticky points: there the
//   true state is a fixed point, so k_band_fix skips the whole stretch in one step and leaves its output
//   (state fixed, input known) to the parallel fill at the end of the launch (band_fill_tiles).
struct BandJob { uint32_t begin, end; float y[4]; uint32_t pad[2]; };   // frames [begin, end) with parked state y
struct BandSpecDesc {
    const float2* x;        // summed input of the vertex, materialised (sum_inputs, no epilogue)
    const float2* xq4;      // the same frames, planar within 4-frame blocks (SumDesc::out_q4)
    float2* out;
    BandState* state;       // carried across chunks
    float* seg_start;       // [nseg][4] state on entry (after warm-up)
    float* seg_final;       // [nseg][4] state on exit
    const float* blk_peaks; // [ceil(frames / 256)] per 256-frame block: input peak, or -1 if bit-constant (k_sum mode 2)
    uint32_t* seg_flags;    // [nseg] bit0: input bit-identical over the whole segment, bit1: input all (+-)0
    float2* seg_x0;         // [nseg] first input frame of the segment
    BandJob* jobs;          // [nseg] parked stretches found by k_band_fix's cascades, executed by its fill phase
    uint32_t* seg_job;      // [nseg] index of the job whose stretch covers (part of) the segment, or ~0
    uint32_t* stats;        // [8]: cascades repaired, segments recomputed, segments parked, jobs, ticket (zeroed by k_band_spec)
    uint32_t nseg, S, W;
    uint32_t Ws;            // short warm-up (k_band_spec picks W or Ws per segment from blk_peaks)
    float live_thr;         // (energy left from before the short window) / (energy fed in inside it) below which Ws is enough
    float gmin;             // the smaller non-zero gamma of the two smoothers
    float decay1, decay4;   // e^(-gamma_min * 256), and its 4th power: decay of a state over one / four 256-frame blocks
    uint32_t post_blocks;   // blocks right before the segment that must not be held constants (~20 / gamma frames)
    uint32_t pass;
    uint32_t first_override;   // see BandDesc
    float lgamma, hgamma;
    PanGain pg;
    // warm-up guess from the block responses (BandRespParam): state at a block boundary = Horner over the last K blocks
    const double* resp;     // nullptr: no guess (the warm-up starts from the input frame itself and takes Ws)
    double Al, Ah;          // (1 - gamma)^256 of the low / high smoother
    uint32_t Kl, Kh;        // blocks after which A^K is negligible
    uint32_t Wq;            // quick warm-up, taken when the energy from before it is below quick_thr x the energy inside it
    uint32_t Wq2;           // medium warm-up, taken when the window is merely alive (no parked stretch); then Ws, then W
    float quick_thr;
    uint32_t pad2[3];
};

// band_pass_gen, TOLERANCE class (engine option "band_mode" 1; <= 1e-6 RMS against the exact forms above): the four
// one-pole recurrences  y <- y + gamma (x - y)  =  (1 - gamma) y + gamma x  as a blocked affine scan, ONE launch per
// vertex -- or per CHAIN of band-pass vertices -- that also evaluates the first vertex' input terms (no materialised
// input sum).
//   A workgroup owns a tile of NF * 256 consecutive frames, a lane NF consecutive frames.  Per lane the zero-state
//   response b of its run (double), a wave / workgroup scan of the (a^NF, b) pairs gives the tile's response B, which
//   is published as eight 8-byte {tag, value} granules (agent-scope atomic stores).  The state entering the tile is
//   the Horner chain  C = sum_j a_tile^(j-1) B_(tile-j)  over the K preceding tiles (a_tile^K <= e^-depth: what is
//   dropped is below the f32 denormal floor), read back with agent-scope atomic loads; then every lane starts from its
//   exact-arithmetic entry state rounded to f32 and runs the REFERENCE's expression over its NF frames, so what
//   differs from the exact kernels is only the entry state's last bits (the f32 trajectory's own accumulated rounding).
//   Tiles are numbered by a ticket drawn at start: a workgroup only ever waits for LOWER tickets, whose holders are
//   running -- nothing depends on the order workgroups are dispatched in or on how many are resident at once.
//   One vertex (n_stages 1): the look-back's wait is bounded all the same -- a predecessor that has not published in
//   time is recomputed by the waiting workgroup itself, same arithmetic, same values.
//   A chain (n_stages > 1: `pass` band-pass vertices linked by single-input, single-consumer gain / pan stages and Adsr
//   vertices -- the shape of BASELINE config 4's 252 effect stages; k_band_chain): the frames stay in registers from
//   stage to stage, the links (`0.0 + x`, envelope gain, pan / gain: BandPost) are applied in between, and every stage
//   has its own granules.  Nobody can recompute a predecessor's stage s > 0, so tiles are numbered by a ticket drawn at
//   start (one only ever waits for lower tickets, whose holders are running), and waits are unbounded.
constexpr uint32_t kScanMaxK = 128;   // look-back depth limit (tiles); slower smoothers take the exact kernels
constexpr uint32_t kScanMaxStages = 128;   // band-pass vertices per launch (the engine cuts longer chains)
struct BandPost {               // one link between two band-pass vertices of a chain
    const float* env;           // an Adsr vertex (its gain buffer, AdsrVDesc::env), or nullptr: a single-input Sum (gain / pan stage)
    PanGain pg;                 // the link vertex' own pan / gain
    uint32_t pad[2];
};
struct BandStageDesc {          // one band-pass vertex
    BandState* state;           // carried across chunks (same slot the exact kernels use)
    unsigned long long* sync;   // [n_tiles][8] granules, zeroed before the launch: chain c's B as (lo, hi) halves of the double
    const double* pw;           // [2][64]: (1 - gamma)^(NF * lane), low / high smoother
    double ap[2][6];            // (1 - gamma)^(NF * 2^s), s = 0..5: the wave scan's step factors
    double aw[2];               // (1 - gamma)^(NF * 64): one wave
    double at[2];               // (1 - gamma)^(NF * 256): one tile
    float lgamma, hgamma;
    uint32_t pass;
    uint32_t K;                 // look-back depth in tiles (1 .. kScanMaxK)
    PanGain pg;                 // the vertex' own epilogue
    // what lies between this vertex' output and the next stage's input, in graph order (unused by the last stage);
    // each link first does its own sum_inputs `0.0 + x`, and so does the next band-pass vertex
    uint32_t n_post, first_override;   // first_override: see BandDesc
    BandPost post[3];
    // k_band_chain (a chain's launch; `pass` vertices only)
    float pn[16][2];            // (1 - gamma)^(n + 1), n = 0 .. NF - 1, {low, high} smoother: what an entry state still weighs after n + 1 frames
    const double* pk;           // [2][kScanMaxK]: (1 - gamma)^(NF * 256 * j): the weight of the tile j + 1 tiles back
    uint32_t Kw, pad3;          // (= K)
    // k_band_chain, guarded form (engine option "band_mode" 2, BandScanDesc::noise): what this vertex adds to the launch's
    // estimate of its own deviation from the reference's f32 trajectory (DESIGN.md 3e "The guard"), {low, high} smoother, every
    // coefficient already multiplied by the STATIC gain G between this vertex' recurrence and the launch's last stage (own pan /
    // gain, the links', every later vertex' -- largest channel amplitude; envelope links multiply in per lane in the kernel) --
    // nzv: G^2 0.25 K0 / (gamma (2 - gamma)), the variance a unit-level state picks up from independent roundings of the
    // recurrence's sum, seen through the 0.5 of `cut`; nzs: G 0.5 2^-24 / gamma, the offset at which an f32 state of unit
    // level parks short of a (nearly) constant input; nzk: 4 * 2^-23 / gamma -- the state counts as parked while
    // |x - y| < nzk |y| (|gamma (x - y)| below 4 ulp(y)); nzk[1] = 0: the faster smoother's test is skipped (both see the same
    // input: when the slower one is parked so is the faster, at a level equal and an offset smaller by gamma_low / gamma_high)
    float nzv[2], nzs[2];
    float nzk[2];
    const float* envt;          // the envelope link behind this vertex, if any: AdsrVDesc::env_tile (at most one Adsr vertex per hop)
};
struct BandScanDesc {
    const InTerm* ins;          // the (first) vertex' input terms, in connect() order
    float2* out;                // the (last) vertex' output
    const BandStageDesc* stages;
    uint32_t* ticket;           // {tile counter, "tile 0 has read the carried states"}, zeroed before the launch
    uint32_t n_stages;
    uint32_t k, term_mode, n_tiles;
    uint32_t flags;             // bit 0: (tests) every poll times out at once -> all predecessors recomputed (n_stages 1)
    uint32_t pad;
    // k_band_chain: a Normalize vertex whose one input is the chain's last vertex (through that stage's `post` links) is
    // evaluated by the launch itself, fresh-render form (SumDesc mode 5 fields: state, init, peaks, init_copy, sync with one
    // granule per tile, pg, out / pcm / qmode / amplitude); the host has checked block length == 1 024 frames = one wave-tile
    const SumDesc* norm;
    // k_band_chain: the first vertex' one input is a Sum vertex evaluated right here -- `ins` / `k` / `term_mode` are THAT
    // vertex' terms, `pre` its pan / gain, `pre2` those of a gain / pan stage between it and the band-pass vertex (flags 0:
    // nothing to apply -- no such vertex, or one without pan and gain)
    PanGain pre, pre2;
    // [n_tiles] granules, zeroed before the launch -- the first stage at which a tile's state went non-finite
    // (n_stages: never).  The reference's state stays NaN for good once it is; the look-back forgets a tile after K tiles,
    // so every tile learns at the end of the chain whether ANY earlier tile was poisoned and turns NaN itself if so
    unsigned long long* poison;
    // k_band_chain: [n_tiles] granules, zeroed before the launch -- the first frame of the chunk (or 0xFFFFFFFF: none) at which
    // the chain's RIGHT input is not finite (tile 0: frame 0 when a carried right smoother state is not).  A `pass` vertex'
    // right output is cutr * 0 + (r - cutl) * 1 (extensions.rs:682-687): its right smoothers, which this kernel does not
    // run, reach the output in exactly one way -- once non-finite (an infinite or NaN right input frame makes them so, for
    // good) they turn it NaN.  Every tile publishes this at its start and reads all earlier tiles' at its end.
    unsigned long long* rpoison;
    // Guarded form (band_mode 2) -- k_band_chain: [ceil(frames / 1024)], per wave-tile; k_band_scan (a single vertex, `cut`
    // vertices included): [n_tiles], per workgroup tile -- the estimated ENERGY (sum over its frames of variance + offset^2) of
    // the launch's deviation from the reference at the launch's output (behind the fused Normalize vertex where there is
    // one); nullptr: not guarded (band_mode 1).  Read by k_band_audit at the end of the submission.
    float* noise;
    float nz_end;               // the static gain behind the last stage that the kernel applies itself (fused Normalize vertex' pan / gain)
    // ... or, where this launch is the graph's ONLY guarded one and carries the Normalize vertex itself, the verdict right
    // here, without k_band_audit: every tile leaves its energy as one granule, the tile with the last ticket gathers them,
    // compares  sum x nz_scale (= (static gain to the output)^2 / frames)  with nz_thr2 and writes the host words
    float nz_scale;
    unsigned long long* nz_sync;   // [n_tiles] granules, zeroed before the launch; nullptr: leave the wave-tiles' energies in `noise`
    uint32_t* nz_host;          // AuditHead::host_word
    float nz_thr2, pad5;
    // ... and with it what the graph's probed sine vertices measured (ProbeDesc::noise; engine option "sine_mode" 2), where every
    // one of them reaches the output through this launch's Normalize vertex: per sample frame the energy of the vertex' deviation
    // at its own output, nz_xcnt (<= 64) samples per wave-tile; nz_xg2 = (its static gain to the Normalize vertex' INPUT)^2.
    // The tile adds its own samples through its own 1 / max, like its own estimate.  nullptr: none.
    const float* nz_extra[2];
    float nz_xg2[2];
    uint32_t nz_xcnt, pad6;
    // ... measured BY THIS LAUNCH (round 6): with one probed vertex and a sample every 256 frames a tile's 4 096 frames hold sixteen
    // sample frames -- one workgroup of k_sine_probe's -- and the tile evaluates them itself, right behind its ticket (the probed
    // vertex' output is complete: it was written by an earlier launch), keeps the sixteen energies in LDS and adds them to its
    // verdict: no k_sine_probe launch (10 us of BASELINE config 3's 114).  nz_extra[0] is nullptr then; nz_xg2[0] as above.
    const struct ProbeDesc* nz_probe;
};
// The guard's verdict (engine option "band_mode" 2): one workgroup per graph adds up what its scan launches estimated, carried to
// the graph's output -- a static gain per launch (the host walks the graph: pan / gain of everything downstream) and, where the
// path runs through ONE Normalize vertex that the launch has not evaluated itself, that vertex' running maximum per block
// (its peak table and carried max, as k_scale / k_norm_fix read them) -- and raises a word in page-locked host memory when
// the estimated RMS deviation exceeds the bound: the engine then renders the graph again with the exact kernels.
struct AuditDesc {
    const float* noise;         // a launch's per-wave-tile energies
    const float* peaks;         // the Normalize vertex on the way to the output: [nb] block peaks (nullptr: none, or evaluated by the launch)
    const float* init_copy;     // ... its carried max at the chunk start
    uint32_t n_wt, nb, bl;      // entries of `noise`; the Normalize vertex' blocks and block length
    uint32_t tile_frames;       // frames per entry of `noise`: 1 024 (k_band_chain: a wave-tile) or NF x 256 (k_band_scan: a workgroup tile)
    float gain;                 // static gain from the launch's output to the graph's
    uint32_t sampled;           // 1: the entries are k_sine_probe's samples (ProbeDesc::noise) -- entry w was measured AT frame probe_frame(w, log2 tile_frames), whose block's running max it goes through
    uint32_t pad2[2];
};
struct AuditHead {              // one graph of the submission
    const AuditDesc* descs;
    uint32_t n, frames;
    float thr2;                 // (bound on the RMS)^2
    uint32_t pad;
    uint32_t* host_word;        // page-locked host memory (device address): [0] raised when over the bound, [1] the estimate (f32 bits)
};
void launch_band_audit(const AuditHead* heads, int n_heads, hipStream_t s);

// The sine kinds' guard (engine option "sine_mode" 2 -- the front-end's default).  The fast forms of debug_sine /
// synth (sin_any, affine envelopes, folded products) differ from the reference's own arithmetic by a white, signal-sized
// rounding noise that a graph can amplify without bound (43 dB of cancellation in a `cut` band-pass, then a Normalize vertex).
// Nothing models that noise: k_sine_probe MEASURES it.  Behind the vertex' launch, on a sample of the chunk's frames -- one in
// every 1 << stride_log2, at an offset that walks all residues -- it evaluates the frame the reference's way (synth_frame /
// sine_frame with exact_sin: glibc's sinf, adsr.rs's divisions, the reference's order of products -- what "sine_mode" 1 renders)
// and takes the squared distance to what the fast launch left in the vertex' buffer (pan / gain applied on both sides; the
// larger channel).  noise[i] = that x the frames the sample stands for: an energy in the audit's own units (AuditDesc::noise
// with tile_frames = the stride), carried to the output by k_band_audit -- or by the chain launch's own verdict -- like a scan
// launch's estimate: static gain, the Normalize vertex' running 1 / max AT THE SAMPLE'S OWN BLOCK.  A NaN on one side only
// counts as an infinite energy.
// sample i of a chunk probed with stride 1 << lg: one frame inside [i << lg, (i + 1) << lg), at an offset that walks all residues
// (block starts, in-block envelope clocks and event frames are not favoured)
#if defined(__HIPCC__)
__host__ __device__
#endif
inline uint32_t probe_frame(uint32_t i, uint32_t lg) { return (i << lg) + (((i & 63u) * 37u + (i >> 6) * 11u) & ((1u << lg) - 1u)); }
struct ProbeDesc {
    SynthDesc syn;              // kind 1: the vertex' descriptor with exact_sin = 1, affine = 0 and the RAW interval tables
    SineDesc sine;              // kind 0: ... with exact_sin = 1
    uint32_t kind;
    uint32_t stride_log2;       // one sample per 1 << stride_log2 frames (4 .. 8: short chunks are sampled densely)
    uint32_t n_groups, pad;     // workgroups: 16 samples each
    float* noise;               // [ceil(frames >> stride_log2)], one entry per sample
    const uint32_t* ranges;     // [2 x samples] host-made: the voice records {first, one past the last} of the sample frame's interval
};
void launch_sine_probe(const ProbeDesc* d, int n_desc, uint32_t frames, uint32_t n_groups, hipStream_t s);
void launch_band_scan(const BandScanDesc* d, int n_desc, uint32_t frames, uint32_t term_mode, int nf, hipStream_t s);
int band_scan_resident_capacity(int nf);   // workgroups of k_band_scan resident at once (0: unknown)
void launch_band_chain(const BandScanDesc* d, int n_desc, uint32_t frames, uint32_t term_mode, bool guarded, hipStream_t s);   // NF 16; guarded: every descriptor has `noise`
inline uint32_t band_scan_tile_frames(int nf) { return (uint32_t)nf * (uint32_t)kThreads; }

// ---- build-defined sinc resampler (stands in for the un-vendored rubato crate; DESIGN.md "Resampler") ----
// Shaped like the streaming SincFixedIn the reference drives block by block (state.rs:545-560): output j sits at input
// position j * from / to - sinc_len / 2 (the resampler's documented output delay: the filter only ever looks at frames it
// has already been handed, history before the first frame is zeros), it is complete -- and in a block-by-block run
// emitted -- as soon as input frame floor(j * from / to) has arrived, and nothing is flushed at the end, like the
// reference: ceil(len * to / from) outputs.  One launch over the whole timeline computes exactly what the block-by-block
// run with carried history would (an FIR has no other state).
constexpr int kSincLen = 256, kSincOver = 256;
struct ResampleDesc {
    const float2* in;
    float2* out;
    const float* table;   // [(kSincOver + 1) * kSincLen], built on the host in f64 and rounded once
    uint64_t len, nout, from, to;
};
void launch_resample(const ResampleDesc& d, hipStream_t s);

// ---- sample load pipeline (SampleBank::add, sample.rs:262-303) on the device ----
// raw PCM words -> f32 exactly like hound + `as f32` (sample.rs:264-273): ints are NOT scaled.
enum PcmFormat : uint32_t { PCM_F32 = 0, PCM_U8 = 1, PCM_S16 = 2, PCM_S24 = 3, PCM_S32 = 4 };
void launch_pcm_decode(const uint8_t* raw, float* linear, uint32_t n_values, uint32_t format, hipStream_t s);
// de-interleave by load mode into planar l / r (lengths nl, nr): src_l / src_r pick the source channel of
// the interleaved stream (channel index, stride = channels), per Sample::from (sample.rs:38-77)
void launch_sample_split(const float* linear, uint32_t channels, uint32_t src_l, uint32_t src_r, float* l, float* r,
                         uint32_t nl, uint32_t nr, hipStream_t s);
void launch_absmax(const float* v, uint32_t n, float* out, hipStream_t s);             // absmax (sample.rs:8-14)
void launch_abs_sum_serial(const float* v, uint32_t n, float* out, hipStream_t s);     // mean_energy's f32 sum, in order
void launch_add_planar(const float* a, const float* b, float* out, uint32_t n, hipStream_t s);   // mix_down sum
// frames[i] = {l[i] * scale_l, r[i] * scale_r} with scale = 1.0f / *max (normalize / normalize_seperate / mix_down)
void launch_sample_pack(const float* l, const float* r, const float* max_l, const float* max_r, float2* frames,
                        uint32_t n, hipStream_t s);
// packed 16-bit form of the same sample (un-scaled integer PCM values); *not_int16 is set when a value is
// not an integer in [-32768, 32767] (then the packed form is not usable)
void launch_sample_pack16(const float* l, const float* r, uint32_t* packed, uint32_t n, uint32_t* not_int16, hipStream_t s);

// per-project peak table of a batch: table[first + i * stride] = *src[i] for i < n_own, every other entry 0
void launch_peak_table(const float* const* src, float* table, uint32_t n_total, uint32_t n_own, uint32_t first, uint32_t stride,
                       hipStream_t s);

void launch_band_spec(const BandSpecDesc* d, int n_desc, uint32_t frames, uint32_t max_nseg, hipStream_t s);
void launch_band_fix(const BandSpecDesc* d, int n_desc, uint32_t frames, uint32_t max_nseg, hipStream_t s);
// every descriptor of one launch_sum call has the same term_mode (the engine groups them)
// k_norm1 (SumDesc mode 5 outside the wide all-loop sums): tiles per workgroup (1 | 2 | 4) with which the whole grid of a
// `frames`-long chunk is resident at once, 0 if none; and its launch
int norm1_tiles_per_workgroup(uint32_t term_mode, uint32_t frames);
void launch_norm1(const SumDesc* d, int n_desc, uint32_t frames, uint32_t term_mode, int tpw, uint32_t tag, hipStream_t s);   // tag: see launch_sum
int sum16w_resident_capacity(int nq, bool packed);   // workgroups of k_sum16w<nq, packed> the device holds at once (0: unknown)
// must_wide: the descriptors hold a mode-4 / mode-5 Normalize, which only the k_sum16w forms implement (the engine sets it where they would run anyway)
// tag: what a tile word of a single-pass Normalize (SumDesc::sync, modes 4 / 5) carries beside its value -- 1 when the engine has
// zeroed the words before the launch, otherwise the submission's epoch (engine.cpp, submit_chunk)
void launch_sum(const SumDesc* d, int n_desc, uint32_t frames, uint32_t bl, uint32_t term_mode, bool wide_ok, bool must_wide, uint32_t tag, hipStream_t s);
void launch_scale(const ScaleDesc* d, int n_desc, uint32_t frames, uint32_t bl, int is_scan, hipStream_t s);
// second half of the speculative single-pass normalize: a no-op unless a block peak exceeded the carried max
void launch_norm_fix(const SumDesc* d, int n_desc, uint32_t frames, uint32_t bl, hipStream_t s);
void launch_quantise(const QuantDesc* d, int n_desc, uint32_t frames, hipStream_t s);
void launch_sinf(const float* in, float* out, uint32_t n, int exact, hipStream_t s);   // out[i] = sin_glibc(in[i]) (exact) or sin_any(in[i])
void launch_debug_verify(const uint32_t* p, uint32_t n_words, const uint32_t* seg_sums, uint32_t* report, hipStream_t s);   // (TD_DEBUG_SYNC & 16)
void launch_sample_loop(const LoopDesc* d, int n_desc, uint32_t frames, hipStream_t s);
void launch_sample_multi(const MultiDesc* d, int n_desc, uint32_t frames, hipStream_t s);
void launch_sample_lerp(const LerpDesc* d, int n_desc, uint32_t frames, hipStream_t s);
void launch_debug_sine(const SineDesc* d, int n_desc, uint32_t frames, uint32_t bl, hipStream_t s);
void launch_synth(const SynthDesc* d, int n_desc, uint32_t frames, bool affine, hipStream_t s);   // every descriptor: SynthDesc::affine == affine
void launch_sampsyn(const SampsynDesc* d, int n_desc, uint32_t frames, hipStream_t s);
// k_sources: the source launches of a level and the envelope launch as the parts of ONE grid (kernels.hip); the engine lists
// the parts longest-running family first, launch_sources fills in the grid geometry
enum SourceKind : uint32_t { SRC_SYNTH_AFFINE = 0, SRC_SAMPSYN = 1, SRC_LERP = 2, SRC_ENV = 3 };
struct SourceParts {   // the launches that become parts (n_* 0: none of that kind)
    const SynthDesc* synth = nullptr; int n_synth = 0;       // every descriptor: SynthDesc::affine set
    const SampsynDesc* sampsyn = nullptr; int n_sampsyn = 0;
    const LerpDesc* lerp = nullptr; int n_lerp = 0;
    const AdsrVDesc* env = nullptr; int n_env = 0;
};
struct SourceGrid {
    uint32_t gx[4], end[4];   // per SourceKind: workgroups per descriptor; the first workgroup BEHIND the part
    // The submission's zeroed hand-off words (ChunkBuild::sync_bytes: tickets and granules of the scan launches BEHIND this one
    // on the stream), cleared by this grid's threads instead of by a fill kernel of their own (16-byte words; 0: nothing)
    uint4* zero;
    uint32_t zero_n16, pad;
};
int launch_sources(const SourceParts& P, uint32_t frames, void* zero, size_t zero_bytes, hipStream_t s);   // zero_bytes: a multiple of 16
void launch_adsr(const AdsrVDesc* d, int n_desc, uint32_t frames, uint32_t term_mode, hipStream_t s);
void launch_band_pass(const BandDesc* d, int n_desc, uint32_t frames, hipStream_t s);

}  // namespace tdk
