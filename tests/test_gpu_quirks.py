"""One small project per quirk of SURVEY.md appendix A (Q1..Q18): the HIP engine must reproduce each of
them exactly like the oracle (which restates the reference line by line).  Every project is bit-exact class
(no sinf), so the comparison is bitwise on the f32 output and on the PCM."""
import numpy as np
import pytest

from termdaw_amd import workloads as W
from test_gpu_parity import assert_bit_exact, assert_close

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def both(p, gpu_api, oracle, scan=False, renders=1):
    gb, ob = p.build(gpu_api), p.build(oracle)
    out = None
    for _ in range(renders):
        gp, gf = p.render(gpu_api, built=gb, scan=scan)
        op, of = p.render(oracle, built=ob, scan=scan)
        assert np.array_equal(_bits(gf), _bits(of)), "f32 differs"
        assert np.array_equal(gp, op), "pcm differs"
        out = (gp, gf)
    return out


def base(bl=256, seconds=0.25):
    p = W.ProjectScript(48000, bl)
    p.set_length(seconds)
    p.assets["n"] = W.Asset(W.noise_int16(3, 700))
    p.assets["k"] = W.Asset(W.kick_int16(4, 3000))
    p.load_sample("n", "n", "")
    p.load_sample("k", "k", "")
    return p


def events(p, name, ev):
    p.event_files[name] = np.array(ev, np.float32).reshape(-1, 3)
    p.load_midi_floww(name, name)


@pytest.mark.parametrize("angle", [0.0, 0.00099, -0.00099, 0.001, 0.0011, -0.0011, 50.0, 20.0, 90.0, -90.0, 500.0])
def test_q1_pan_skip_threshold_and_clamp(gpu_api, oracle, angle):
    """Q1: |angle| < 0.001 skips panning (unity), 0.0011 already applies ~0.7071; angles clamp to +-90."""
    p = base()
    p.add_sampleloop("a", 1.0, angle, "n")
    p.add_sum("o", 1.0, 0.0)
    p.connect("a", "o")
    p.set_output("o")
    _, f = both(p, gpu_api, oracle)
    if abs(angle) < 0.001:
        assert np.abs(f).max() == 1.0   # untouched peak-normalised sample


@pytest.mark.parametrize("gain", [1.0, 1.00099, 0.99901, 1.001, 0.999, 0.0, -2.0])
def test_gain_skip_threshold(gpu_api, oracle, gain):
    p = base()
    p.add_sampleloop("a", gain, 0.0, "n")
    p.add_sum("o", gain, 10.0)
    p.connect("a", "o")
    p.set_output("o")
    both(p, gpu_api, oracle)


def test_q2_q3_running_peak_and_nested_normalize_scan(gpu_api, oracle):
    """Q2: un-scanned normalize divides by the running peak starting at 1e-6.  Q3: during a scan the buffer
    is scaled by the STALE max, so a normalize downstream of another records a wrong scan_max."""
    p = base(seconds=0.4)
    p.add_sampleloop("a", 0.3, 0.0, "k")
    p.add_normalize("inner", 0.5, 0.0)
    p.add_sum("mid", 2.0, 0.0)
    p.add_normalize("outer", 1.0, 0.0)
    p.connect("a", "inner")
    p.connect("inner", "mid")
    p.connect("mid", "outer")
    p.set_output("outer")
    both(p, gpu_api, oracle, scan=False, renders=2)
    both(p, gpu_api, oracle, scan=True, renders=2)
    gb, ob = p.build(gpu_api), p.build(oracle)
    gb[2].true_normalize_scan(gb[0], gb[1], p.cs)
    ob[2].true_normalize_scan(ob[0], ob[1], p.cs)
    for v in ("inner", "outer"):
        assert gb[2].get_normalization_value(v) == ob[2].get_normalization_value(v)
    assert gb[2].get_normalization_value("outer") > 1000.0   # the stale 1e-6 inflated what the outer one saw


def test_q4_q14_scan_and_rerender_keep_partial_state(gpu_api, oracle):
    """Q4 / Q14: set_time rewinds only some vertex state and render does not rewind the FlowwBank first --
    a scanned render and a second render differ from a fresh one, identically on both sides."""
    p = base(seconds=0.3)
    events(p, "h", [(0.01 + 0.04 * i, 60.0, 0.5 + 0.05 * (i % 5)) for i in range(8)])
    p.add_sample_multi("m", 1.0, 0.0, "k", "h", -1)
    p.add_sample_lerp("l", 1.0, 0.0, "k", "h", -1, 32)
    p.add_adsr("e", 1.0, 0.0, 1.0, "h", False, True, -1, [0.01, 0.02, 0.6, 0.05, 0.3, 0.01])
    p.add_normalize("o", 1.0, 0.0)
    p.connect("m", "e")
    p.connect("l", "e")
    p.connect("e", "o")
    p.set_output("o")
    first = both(p, gpu_api, oracle, scan=True, renders=1)
    fresh = both(p, gpu_api, oracle, scan=False, renders=1)
    assert not np.array_equal(first[1], fresh[1])
    both(p, gpu_api, oracle, scan=False, renders=3)


def test_q9_adsr_nonmatching_hit_leaves_frame_untouched(gpu_api, oracle):
    """Q9: use_off=false with a note filter: a hit of another note `continue`s the sample loop."""
    p = base()
    events(p, "h", [(0.01, 60.0, 0.9), (0.02, 61.0, 0.9), (0.02 + 1 / 48000.0, 61.0, 0.9), (0.03, 60.0, 0.4),
                    ((256 * 3 - 1) / 48000.0, 61.0, 0.7)])   # ... and one on the last frame of a block
    p.add_sampleloop("a", 1.0, 0.0, "n")
    p.add_adsr("e", 1.0, 0.0, 1.0, "h", False, True, 60, [0.001, 0.01, 0.5, 0.01, 0.2, 0.02])
    p.connect("a", "e")
    p.set_output("e")
    _, f = both(p, gpu_api, oracle)
    m = int(np.float32(0.02) * np.float32(48000))
    loop = W.noise_int16(3, 700).astype(np.float32)
    loop = loop / np.abs(loop).max()
    assert np.array_equal(f[m], loop[m % 700])   # that one frame passes un-enveloped


def test_q10_second_hit_on_the_same_frame_is_lost(gpu_api, oracle):
    """Q10: the drum pull returns only the first on-event of a frame; the note filter runs afterwards, so a
    matching hit hidden behind another note's hit on the same frame never sounds."""
    ev = [(0.01, 61.0, 0.9), (0.01, 60.0, 0.8), (0.05, 60.0, 0.7), (0.05, 61.0, 0.6)]
    first_sound = {}
    for note in (60, 61):
        p = base()
        events(p, "h", ev)
        p.add_sample_multi("m", 1.0, 0.0, "k", "h", note)
        p.add_sample_lerp("l", 1.0, 0.0, "k", "h", note, 16)
        p.add_sum("o", 1.0, 0.0)
        p.connect("m", "o")
        p.connect("l", "o")
        p.set_output("o")
        _, f = both(p, gpu_api, oracle)
        first_sound[note] = int(np.nonzero(np.abs(f).max(axis=1) > 0)[0][0])
    f001, f005 = int(np.float32(0.01) * np.float32(48000)), int(np.float32(0.05) * np.float32(48000))
    assert f001 <= first_sound[61] <= f001 + 2      # 61 is first at 0.01 s ...
    assert f005 <= first_sound[60] <= f005 + 2      # ... so 60's hit there is lost; 60 first sounds at 0.05 s


def test_q11_stale_start_index_after_last_event(gpu_api, oracle):
    """Q11: once no later event exists start_indices stay where they were; a simple-pull vertex whose cursor
    is behind the block (stale event) sees nothing more, a drum-pull one skips ahead."""
    p = base(seconds=0.5)
    events(p, "h", [(0.01, 60.0, 0.9), (0.011, 60.0, 0.0), (0.02, 62.0, 0.8)])
    p.add_sample_multi("m", 1.0, 0.0, "k", "h", -1)
    p.add_adsr("e", 1.0, 0.0, 1.0, "h", True, False, -1, [0.005, 0.01, 0.5, 0.02, 0.3, 0.05, 0.2, 0.05, 0.0])
    p.connect("m", "e")
    p.set_output("e")
    both(p, gpu_api, oracle, renders=2)


def test_q12_unreachable_vertices_never_run(gpu_api, oracle):
    """Q12: a vertex that does not reach the output keeps its initial state (its sampleloop cursor does not
    advance); connecting it later in a second project shows the difference on both sides."""
    p = base()
    p.add_sampleloop("dead", 1.0, 0.0, "k")
    p.add_sampleloop("live", 1.0, 0.0, "n")
    p.add_sum("island", 1.0, 0.0)
    p.add_sum("o", 1.0, 0.0)
    p.connect("dead", "island")
    p.connect("live", "o")
    p.set_output("o")
    both(p, gpu_api, oracle, renders=2)


def test_q5_same_frame_on_and_off_sticks_the_wavetable_voice(gpu_api, oracle):
    """Q5 on the IEEE-only voice (sampsyn): note-off on the note-on frame gives rel_t == 0.0 -> never released."""
    p = W.ProjectScript(48000, 256)
    p.set_length(0.3)
    p.resources["wt"] = W.wavetable_bytes(5, 8, 64)
    p.load_resource("wt", "wt")
    events(p, "h", [(0.01, 57.0, 0.8), (0.01, 57.0, 0.0), (0.05, 64.0, 0.6), (0.08, 64.0, 0.0)])
    p.add_sampsyn("v", 1.0, 0.0, "h", [0.01, 0.02, 0.7, 0.05, 0.4, 0.03], "wt")
    p.add_sum("o", 1.0, 0.0)
    p.connect("v", "o")
    p.set_output("o")
    _, f = both(p, gpu_api, oracle)
    assert np.abs(f[-64:]).max() > 0.0   # the stuck voice still sounds at the end


def test_q6_zero_length_adsr_segments(gpu_api, oracle):
    """Q6: a zero-length segment gives 0/0 = NaN at its boundary and (t/0).min(1.0) = 1.0 in the release."""
    p = base()
    events(p, "h", [(0.0, 60.0, 0.9), (0.02, 60.0, 0.5)])
    p.add_sampleloop("a", 1.0, 0.0, "n")
    p.add_adsr("e", 1.0, 0.0, 1.0, "h", False, True, -1, [0.0, 0.0, 0.5, 0.01, 0.2, 0.0])
    p.connect("a", "e")
    p.set_output("e")
    gb, ob = p.build(gpu_api), p.build(oracle)
    gp, gf = p.render(gpu_api, built=gb)
    op, of = p.render(oracle, built=ob)
    assert np.array_equal(np.isnan(gf), np.isnan(of))
    ok = ~np.isnan(of)
    assert np.array_equal(_bits(gf)[ok], _bits(of)[ok]) and np.array_equal(gp, op)   # NaN -> 0 in the PCM cast


def test_q7_q8_band_pass_right_channel_uses_left_cut_and_wet_only_gates(gpu_api, oracle):
    p = base(seconds=0.3)
    p.add_sampleloop("a", 1.0, -60.0, "n")     # L != R
    p.add_bandpass("wet", 1.0, 0.0, 0.5, 300.0, 3000.0, True)       # wet 0.5: processed fully (Q8)
    p.add_bandpass("dry", 1.0, 0.0, 0.00005, 300.0, 3000.0, True)   # wet < 1e-4: bypass
    p.add_sum("o", 1.0, 0.0)
    p.connect("a", "wet")
    p.connect("a", "dry")
    p.connect("wet", "o")
    p.connect("dry", "o")
    p.set_output("o")
    both(p, gpu_api, oracle, renders=2)


def test_q13_render_rate_above_project_rate_is_not_resampled(gpu_api, oracle, tmp_path):
    """Q13: render_sr > project sr writes the un-resampled frames under the render_sr header."""
    import struct
    p = W.config1(seconds=0.1)
    p.set_render_samplerate(96000)
    want, _ = p.render(oracle)
    lua = p.to_lua(str(tmp_path / "a"))
    s = gpu_api.State("", 48000, 1024)
    assert s.refresh(lua), gpu_api.last_error()
    out = str(tmp_path / "o.wav")
    s.render(out)
    raw = open(out, "rb").read()
    assert struct.unpack("<I", raw[24:28])[0] == 96000
    assert np.array_equal(np.frombuffer(raw[44:], np.int16).reshape(-1, 2), want)


def test_q15_q17_whole_blocks_and_peak_normalised_ints(gpu_api, oracle):
    """Q15: 3 s at 48 kHz / 1024 renders 141 whole blocks (144 384 frames).  Q17: int PCM is peak-normalised at
    load, never divided by 2^15."""
    p = W.config1(seconds=3.0)
    assert p.cs == 141
    gp, gf = both(p, gpu_api, oracle)
    assert gp.shape[0] == 144384
    sb, _, _ = p.build(gpu_api)
    l, r = sb.get_sample(sb.get_index("snare"))
    assert max(np.abs(l).max(), np.abs(r).max()) == 1.0


def test_q16_sample_lerp_holds_the_last_frame(gpu_api, oracle):
    p = base(seconds=0.3)
    p.assets["dc"] = W.Asset(W.kick_int16(9, 400) + np.int16(200))
    p.load_sample("dc", "dc", "")
    events(p, "h", [(0.01, 60.0, 0.5)])
    p.add_sample_lerp("l", 1.0, 0.0, "dc", "h", -1, 0)
    p.add_sum("o", 1.0, 0.0)
    p.connect("l", "o")
    p.set_output("o")
    _, f = both(p, gpu_api, oracle)
    assert f[-1, 0] != 0.0 and np.all(f[-100:] == f[-1])


def test_q18_vertices_built_by_type_edges_in_script_order(gpu_api, oracle):
    """Q18: creation order is by type, sum order is connect() order (f32 addition does not commute here)."""
    p = base()
    p.add_sum("o", 1.0, 0.0)
    p.add_sampleloop("b", 1e-4, 0.0, "n")
    p.add_sampleloop("a", 1.0, 0.0, "k")
    p.add_sampleloop("c", -1.0, 0.0, "k")
    p.connect("c", "o")
    p.connect("b", "o")
    p.connect("a", "o")
    p.connect("b", "o")
    p.set_output("o")
    both(p, gpu_api, oracle)


@pytest.mark.parametrize("project", ["drum", "synth", "config4"])
def test_event_table_cache_is_value_neutral(gpu_api, oracle, project):
    """Compiled event tables are reused across renders (same events, cursor and carried state) and across identical
    vertices of one chunk; with the cache off every vertex replays its events every time.  Same bytes either way,
    also across renders whose carried state differs (quirk Q4) and after set_time."""
    mk = {"drum": lambda: W.drum_project(seconds=1.5), "synth": lambda: W.synth_project(seconds=1.5),
          "config4": lambda: W.config4(seconds=1.0, depth=30)}[project]
    outs = []
    for cache in (1, 0):
        p = mk()
        sb, fb, g = p.build(gpu_api)
        g.set_option("debug.table_cache", cache)
        seq = []
        for rep in range(3):
            seq.append(g.render_all(sb, fb, p.cs, 16))
        g.set_time(5 * p.bl)                      # playhead moved: FlowwBank cursor and loop cursors change the key
        fb.set_time(5 * p.bl)
        seq.append(g.render_all(sb, fb, 7, 16))
        fb.set_time(0)
        g.true_normalize_scan(sb, fb, p.cs)
        seq.append(g.render_all(sb, fb, p.cs, 16))
        outs.append(seq)
    for a, b in zip(*outs):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
    p = mk()
    ob = p.build(oracle)
    ref = ob[2].render_all(ob[0], ob[1], p.cs, 16)
    (assert_close if project == "synth" else assert_bit_exact)(outs[0][0], ref)


@pytest.mark.parametrize("consumer", ["normalize", "sum_out", "band", "band_serial", "band_thru", "two_inputs", "three_terms", "two_consumers", "adsr"])
@pytest.mark.parametrize("stage", [False, True])
@pytest.mark.parametrize("chunk", [0, 5000])
@pytest.mark.parametrize("inp", ["sum", "loop", "stage"])
def test_adsr_vertex_evaluated_by_its_consumer(gpu_api, oracle, consumer, stage, chunk, inp):
    """An Adsr vertex with one materialised input and one consumer of the summing family (directly or through a gain / pan
    stage) is evaluated inside the consumer's summing kernel (term kind 5, option inline_adsr): every consumer kind, with
    and without the stage, as the only term and among others (a second input; two envelopes and an inlined loop source),
    -0.0 inputs, skipped hits (quirk Q9), the shapes that must NOT be inlined (a second consumer, an Adsr consumer),
    single- and multi-chunk, scanned and unscanned -- bit-exact against the oracle and against the materialised form."""
    p = W.ProjectScript(48000, 1024)
    p.set_length(0.6)
    pcm = W.noise_int16(31, 4001)
    pcm[100:140] = 0
    p.assets["a"] = W.Asset(pcm)
    p.load_sample("a", "a", "")
    p.add_sampleloop("l", -0.7, 10.0, "a")      # negative gain: zeros of the asset become -0.0
    p.add_sum("src", 1.0, 0.0)                  # a materialised vertex (two inputs: no gain / pan stage)
    p.connect("l", "src")
    p.connect("l", "src")
    # the envelope's own input: that edge buffer, an inlined loop source, or a gain / pan stage over the buffer
    env_in = "src"
    if inp == "loop":
        p.add_sampleloop("l2", 0.6, -25.0, "a")
        env_in = "l2"
    elif inp == "stage":
        p.add_sum("pre", 0.8, 33.0)
        p.connect("src", "pre")
        env_in = "pre"
    p.event_files["h"] = np.array([(0.03, 60.0, 0.9), (0.11, 61.0, 0.5), (0.13, 60.0, 0.0), (0.3, 60.0, 0.7), (0.45, 60.0, 0.0)], np.float32)
    p.load_midi_floww("h", "h")
    p.add_adsr("env", 0.8, -20.0, 0.7, "h", True, False, 60, [0.01, 0.02, 0.6, 0.03, 0.2, 0.05])
    p.connect(env_in, "env")
    prev = "env"
    if stage:
        p.add_sum("st", 1.3, 15.0)
        p.connect(prev, "st")
        prev = "st"
    out = "c"
    if consumer == "normalize":
        p.add_normalize("c", 0.9, 3.0)
    elif consumer == "sum_out":
        p.add_sum("c", 0.9, 0.0)                # a single-input Sum that IS the output: materialised, reads through
    elif consumer in ("band", "band_serial"):
        p.add_bandpass("c", 1.0, 0.0, 0.9, 300.0, 5000.0, True)
    elif consumer == "band_thru":
        p.add_bandpass("c", 1.1, 0.0, 0.0, 300.0, 5000.0, True)   # wet 0: the summed input passes through
    elif consumer == "two_inputs":
        p.add_normalize("c", 1.0, 0.0)
        p.connect("src", "c")
    elif consumer == "three_terms":
        p.add_adsr("env2", 1.2, 10.0, 1.0, "h", False, True, -1, [0.002, 0.03, 0.5, 0.02, 0.1, 0.04])
        p.connect("src", "env2")
        p.add_sum("c", 0.9, -5.0)
        p.add_sum("tail", 1.0, 0.0)             # (keeps "c" from being the output: a materialised inner Sum)
        p.connect("env2", "c")
        p.connect("l", "c")                     # an inlined loop source between the two envelopes
        p.connect(prev, "c")
        p.connect("src", "tail")
        p.connect("c", "tail")
    elif consumer == "two_consumers":
        p.add_normalize("c", 1.0, 0.0)
        p.add_sum("side", 0.5, 0.0)
        p.connect(prev, "side")
        p.connect("side", "c")
    else:
        p.add_adsr("c", 1.0, 0.0, 0.5, "h", False, True, -1, [0.01, 0.02, 0.6, 0.03, 0.2, 0.05])
    if consumer != "three_terms":
        p.connect(prev, "c")
    if consumer == "three_terms":
        out = "tail"
    if consumer in ("band", "band_serial", "band_thru", "adsr"):
        p.add_normalize("o", 1.0, 0.0)
        p.connect("c", "o")
        out = "o"
    p.set_output(out)
    outs = []
    for inline in (1, 0):
        built = p.build(gpu_api)
        obuilt = p.build(oracle)
        built[2].set_option("debug.inline_adsr", inline)
        if chunk:
            built[2].set_option("max_chunk_frames", chunk)
        if consumer == "band_serial":
            built[2].set_option("debug.band_serial", 1)
        seq = [p.render(gpu_api, built=built)]
        assert_bit_exact(seq[0], p.render(oracle, built=obuilt))
        built[2].true_normalize_scan(built[0], built[1], p.cs)
        obuilt[2].true_normalize_scan(obuilt[0], obuilt[1], p.cs)
        seq.append(p.render(gpu_api, built=built))
        assert_bit_exact(seq[1], p.render(oracle, built=obuilt))
        outs.append(seq)
    for a, b in zip(*outs):
        assert_bit_exact(a, b)


@pytest.mark.parametrize("band_mode", [0, 1])
def test_q4_band_first_travels_in_the_descriptors(gpu_api, oracle, band_mode):
    """set_time re-arms BandPass.first (extensions.rs:196-204, quirk Q4) without device work: the next submission's
    descriptors say so (first_override).  The sequences that could lose it: renders WITHOUT a set_time in between (the
    state continues), two set_time calls in a row, a set_time followed by a graph edit (the host copy of the states is
    pulled and pushed back: it must carry the re-armed word), a band-pass vertex that does not reach the output while
    another does (quirk Q12: its word stays armed until it runs), and block pulls."""
    def mk():
        p = W.ProjectScript(48000, 1024)
        p.set_length(0.9)
        p.assets["a"] = W.Asset(W.noise_int16(77, 30001))
        p.load_sample("a", "a", "")
        p.add_sampleloop("src", 1.0, 0.0, "a")
        p.add_bandpass("b1", 1.0, 0.0, 1.0, 300.0, 5000.0, True)
        p.add_bandpass("b2", 1.2, 10.0, 1.0, 80.0, 0.0, False)
        p.add_bandpass("side", 1.0, 0.0, 1.0, 500.0, 2000.0, True)      # (reaches the output only after the edit below)
        p.add_sum("out", 1.0, 0.0)
        p.connect("src", "b1")
        p.connect("b1", "b2")
        p.connect("src", "side")
        p.connect("b2", "out")
        p.set_output("out")
        return p
    p = mk()
    built = []
    for be in (gpu_api, oracle):
        sb, fb, g = p.build(be)
        if be is gpu_api:
            g.set_option("band_mode", band_mode)
        built.append((sb, fb, g))
    cmp = assert_close if band_mode else assert_bit_exact

    def both(fn):
        outs = [fn(*b) for b in built]
        if outs[0] is not None:
            cmp(outs[0], outs[1])

    cs = p.cs
    both(lambda sb, fb, g: g.render_all(sb, fb, cs, 16))                       # first render: states seeded from buf[0]
    both(lambda sb, fb, g: (fb.set_time(0), g.render_all(sb, fb, cs, 16))[1])   # render_all rewound the graph: re-armed
    def no_rewind(sb, fb, g):                                                  # block pulls: no set_time in between
        outs = []
        for _ in range(3):
            l, r = g.render(sb, fb)
            fb.set_time_to_next_block()
            outs.append(np.stack([l, r], 1))
        return np.concatenate(outs)
    got, ref = [no_rewind(*b) for b in built]
    if band_mode:
        assert float(np.sqrt(np.mean((got.astype(np.float64) - ref) ** 2))) <= 1e-6 * max(1.0, float(np.abs(ref).max()))
    else:
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    def twice_then_edit(sb, fb, g):
        g.set_time(0); g.set_time(0); fb.set_time(0)
        g.connect("side", "out")                                                # the edit: host copy of the states pulled, pushed back
        return g.render_all(sb, fb, cs, 16)
    both(twice_then_edit)
    both(lambda sb, fb, g: (fb.set_time(0), g.render_all(sb, fb, cs, 16))[1])
