"""GPU parity: the HIP engine (through the C ABI) against the CPU oracle on identical projects.

Bar: bit-exact int PCM *and* bit-exact f32 edge values for the integer/index/IEEE-only vertex kinds
(sum, normalize, sampleloop, sample_multi, sample_lerp, adsr vertex, bandpass); for the kinds that
evaluate sinf (debug_sine, synth) the tolerance is 1e-6 RMS on the f32 output and +-1 LSB on PCM.
"""
import numpy as np
import pytest

from termdaw_amd import workloads as W

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def assert_bit_exact(got, ref):
    gp, gf = got
    rp, rf = ref
    assert gp.shape == rp.shape
    assert np.array_equal(np.isnan(gf), np.isnan(rf))   # same NaN positions; NaN payloads are not compared
    bad = np.nonzero((_bits(gf) != _bits(rf)) & ~np.isnan(rf))
    assert bad[0].size == 0, "f32 mismatch at frames %s: got %s ref %s" % (bad[0][:5], gf[bad][:5], rf[bad][:5])
    assert np.array_equal(gp, rp)


def assert_close(got, ref, rms_tol=1e-6):
    gp, gf = got
    rp, rf = ref
    assert gp.shape == rp.shape
    assert not np.isnan(gf).any()
    rms = float(np.sqrt(np.mean((gf.astype(np.float64) - rf.astype(np.float64)) ** 2)))
    assert rms <= rms_tol, "rms %g" % rms
    assert np.abs(gp.astype(np.int64) - rp.astype(np.int64)).max() <= 1


@pytest.mark.parametrize("scan", [False, True])
def test_config1_readme_project(gpu_api, oracle, scan):
    p = W.config1()
    assert p.cs == 141
    assert_bit_exact(p.render(gpu_api, scan=scan), p.render(oracle, scan=scan))


@pytest.mark.parametrize("scan", [False, True])
def test_config2_small(gpu_api, oracle, scan):
    p = W.config2(seconds=2.0, n_src=64)
    assert_bit_exact(p.render(gpu_api, scan=scan), p.render(oracle, scan=scan))


def test_config2_odd_fanin_and_duplicate_edges(gpu_api, oracle):
    p = W.config2(seconds=0.5, n_src=7)
    p.connect("vs03", "sum")   # duplicate connect -> summed twice (graph.rs:76)
    p.connect("sum", "sum")    # self edge rejected
    p.connect("nope", "sum")   # unknown vertex rejected
    assert_bit_exact(p.render(gpu_api), p.render(oracle))


@pytest.mark.parametrize("scan", [False, True])
def test_drum_project(gpu_api, oracle, scan):
    p = W.drum_project()
    assert_bit_exact(p.render(gpu_api, scan=scan), p.render(oracle, scan=scan))


@pytest.mark.parametrize("bl", [64, 1000, 1024, 2048, 333])
def test_drum_project_block_lengths(gpu_api, oracle, bl):
    p = W.drum_project(seconds=1.0, bl=bl)
    assert_bit_exact(p.render(gpu_api), p.render(oracle))


@pytest.mark.parametrize("scan", [False, True])
def test_synth_project(gpu_api, oracle, scan):
    p = W.synth_project()
    assert_close(p.render(gpu_api, scan=scan), p.render(oracle, scan=scan))


@pytest.mark.parametrize("bl,blocks,on_frame", [(1000, 3, 2300), (1000, 3, 2420), (1000, 2, 1300), (1000, 3, 2700), (1024, 3, 2900)])
def test_synth_envelope_breakpoint_in_the_last_partial_tile(gpu_api, oracle, bl, blocks, on_frame):
    """k_synth finds the piece of a held voice's envelope once per wave, from the wave's first and last envelope time.
    With a block length that does not divide the 1024-frame tile the chunk ends inside a tile: the lanes beyond it shadow
    the last frame, and an attack -> decay breakpoint a few frames before the end (5 ms after the note-on) must still send
    the wave down the per-frame path -- every frame against the oracle, and the frames around the breakpoint one by one.
    (The first three cases put both frame pairs of the tile's last wave into the note's interval with the breakpoint
    between the wave's first frame and the chunk's last.)"""
    p = W.ProjectScript(48000, bl)
    p.set_length(blocks * bl / 48000.0 - 1e-4)
    assert p.cs == blocks
    p.event_files["n"] = np.array([(0.001, 57.0, 0.7), ((on_frame + 0.5) / 48000.0, 60.0, 0.9)], np.float32)
    p.load_midi_floww("n", "n")
    conf = [0.005, 0.05, 0.3, 0.2, 0.1, 0.1]           # attack 240 frames: the breakpoint sits at on_frame + 240
    p.add_synth("syn", 1.0, 0.0, "n", 0.4, 0.3, conf, 1.0, 0.8, conf, 0.5, [0.004, 0.05, 0.5, 0.2, 0.2, 0.1])
    p.add_sum("out", 1.0, 0.0)
    p.connect("syn", "out")
    p.set_output("out")
    got, ref = p.render(gpu_api), p.render(oracle)
    assert_close(got, ref)
    assert np.abs(got[1].astype(np.float64) - ref[1].astype(np.float64)).max() < 2e-6


def test_config3_short(gpu_api, oracle):
    p = W.config3(seconds=4.0)
    assert_close(p.render(gpu_api), p.render(oracle))


def test_render_twice_carries_state(gpu_api, oracle):
    """A second render without refresh continues from the carried vertex state (quirks Q4/Q14)."""
    p = W.drum_project(seconds=1.5)
    gb = p.build(gpu_api)
    ob = p.build(oracle)
    for _ in range(2):
        assert_bit_exact(p.render(gpu_api, built=gb), p.render(oracle, built=ob))


def test_block_pull_matches_reference_render(gpu_api, oracle):
    """td_graph_render_block == Graph::render, block by block, incl. the caller-driven FlowwBank."""
    p = W.drum_project(seconds=0.3)
    gsb, gfb, gg = p.build(gpu_api)
    osb, ofb, og = p.build(oracle)
    for b in range(p.cs):
        gl, gr = gg.render(gsb, gfb)
        ol, orr = og.render(osb, ofb)
        assert np.array_equal(_bits(gl), _bits(ol)) and np.array_equal(_bits(gr), _bits(orr)), "block %d" % b
        gfb.set_time_to_next_block()
        ofb.set_time_to_next_block()
    assert gg.get_time() == og.get_time() == p.cs * p.bl


def test_normalization_value_after_scan(gpu_api, oracle):
    p = W.config1(seconds=1.0)
    gsb, gfb, gg = p.build(gpu_api)
    osb, ofb, og = p.build(oracle)
    assert gg.get_normalization_value("sum") == og.get_normalization_value("sum") == np.float32(0.000001)
    gg.true_normalize_scan(gsb, gfb, p.cs)
    og.true_normalize_scan(osb, ofb, p.cs)
    assert gg.get_normalization_value("sum") == og.get_normalization_value("sum")
    assert gg.get_normalization_value("one") == -1.0


def test_config2_full_size_bit_exact_and_properties(gpu_api, oracle):
    """BASELINE config 2 at full size (2,880,512 frames): bit-exact PCM vs the oracle (the oracle needs ~1 s),
    plus size-independent properties of a running-peak normalise."""
    p = W.config2()
    assert p.cs == 2813
    built = p.build(gpu_api)
    obuilt = p.build(oracle)
    gp, gf = p.render(gpu_api, built=built)
    rp, _ = p.render(oracle, built=obuilt, want_f32=False)
    assert np.array_equal(gp, rp)
    assert built[2].get_normalization_value("sum") == obuilt[2].get_normalization_value("sum")
    peak = np.abs(gf).reshape(p.cs, -1).max(axis=1)
    assert peak[0] == 1.0 and np.all(peak <= 1.0)            # every block is divided by a peak that includes it
    # scanned render: normalised by the global peak -> max |x| == 1 exactly, PCM full scale reached
    built[2].true_normalize_scan(built[0], built[1], p.cs)
    gp2, gf2 = p.render(gpu_api, built=built)
    assert np.abs(gf2).max() == 1.0 and np.abs(gp2.astype(np.int32)).max() == 32767
    # the two renders differ only by the per-block scale: ratio constant inside each block
    blk = 17
    a, b = gf[blk * 1024:(blk + 1) * 1024, 0].astype(np.float64), gf2[blk * 1024:(blk + 1) * 1024, 0].astype(np.float64)
    nz = np.abs(b) > 1e-3
    assert np.ptp(a[nz] / b[nz]) < 1e-5

def test_empty_and_degenerate_renders(gpu_api, oracle):
    p = W.config1(seconds=0.0)      # cs == 0: nothing rendered, no error
    assert p.cs == 0
    gp, gf = p.render(gpu_api)
    assert gp.shape == (0, 2)
    p = W.ProjectScript(48000, 1024)   # all-silent input: 0 * (1/1e-6) stays 0, then scan -> max 0 -> NaN -> PCM 0
    p.set_length(0.1)
    p.assets["z"] = W.Asset(np.zeros((100, 2), np.int16) + np.array([[1, 0]], np.int16))
    p.load_sample("z", "z", "")
    p.add_sampleloop("s", 0.0, 0.0, "z")   # gain 0 -> exact zeros
    p.add_normalize("n", 1.0, 0.0)
    p.connect("s", "n")
    p.set_output("n")
    for scan in (False, True):
        g, o = p.render(gpu_api, scan=scan), p.render(oracle, scan=scan)
        assert np.array_equal(g[0], o[0])
        assert np.array_equal(np.isnan(g[1]), np.isnan(o[1]))


@pytest.mark.parametrize("bits", [8, 24, 32])
def test_other_bit_depths(gpu_api, oracle, bits):
    p = W.config1(seconds=0.5)
    p.set_render_bitdepth(bits)
    gp, _ = p.render(gpu_api)
    op, _ = p.render(oracle)
    assert gp.dtype == op.dtype and np.array_equal(gp, op)


@pytest.mark.parametrize("fuse", [0, 1])
@pytest.mark.parametrize("project", ["config2", "drum"])
def test_source_inlining_is_value_identical(gpu_api, oracle, fuse, project):
    """fuse_sources only changes WHERE a sample_loop source is evaluated, never a value or the sum order."""
    p = W.config2(seconds=1.0, n_src=9) if project == "config2" else W.drum_project(seconds=1.0)
    if project == "config2":
        p.add_sum("side", 0.5, 10.0)          # vs03 feeds two consumers; one source stays un-inlined as the output elsewhere
        p.connect("vs03", "side")
        p.connect("side", "sum")
    built = p.build(gpu_api)
    built[2].set_option("fuse_sources", fuse)
    assert_bit_exact(p.render(gpu_api, built=built), p.render(oracle))


def test_sampleloop_as_output_vertex(gpu_api, oracle):
    p = W.config2(seconds=0.2, n_src=2)
    p.set_output("vs01")                       # a source may be the output (never inlined)
    assert_bit_exact(p.render(gpu_api), p.render(oracle))


@pytest.mark.parametrize("chunk_frames", [1024, 5000, 40000])
@pytest.mark.parametrize("project", ["drum", "synth"])
def test_multi_chunk_render_matches_single_chunk(gpu_api, oracle, chunk_frames, project):
    """Timelines longer than the edge-buffer chunk cap render chunk after chunk with carried state
    (running normalize peak, band-pass state, voice lists, envelope clocks)."""
    p = W.drum_project(seconds=1.5) if project == "drum" else W.synth_project(seconds=1.5)
    built = p.build(gpu_api)
    built[2].set_option("max_chunk_frames", chunk_frames)
    got = p.render(gpu_api, built=built)
    (assert_bit_exact if project == "drum" else assert_close)(got, p.render(oracle))
    built = p.build(gpu_api)
    built[2].set_option("max_chunk_frames", chunk_frames)
    got = p.render(gpu_api, built=built, scan=True)
    (assert_bit_exact if project == "drum" else assert_close)(got, p.render(oracle, scan=True))


def _gappy_project(lo, hi, pass_, seconds=2.0):
    """sample_multi with sparse hits (exact-zero gaps) and a sample_lerp holding its last frame (constant
    non-zero stretches) into one band-pass: the inputs on which speculative segments must be repaired."""
    p = W.ProjectScript(48000, 1024)
    p.set_length(seconds)
    p.assets["pluck"] = W.Asset(W.tone_int16(21, 2500))
    p.assets["kick"] = W.Asset(W.kick_int16(22, 6000) + np.int16(3))     # last frame != 0 -> held DC offset
    p.load_sample("pluck", "pluck", "")
    p.load_sample("kick", "kick", "")
    p.event_files["a"] = np.array([(0.3 * i + 0.01, 60.0, 0.8) for i in range(int(seconds / 0.3))], np.float32)
    p.event_files["b"] = np.array([(0.7 * i + 0.2, 36.0, 1.0) for i in range(int(seconds / 0.7) + 1)], np.float32)
    p.load_midi_floww("a", "a")
    p.load_midi_floww("b", "b")
    p.add_sample_multi("m", 1.0, 0.0, "pluck", "a", -1)
    p.add_sample_lerp("l", 0.5, 30.0, "kick", "b", -1, 64)
    p.add_bandpass("bp", 1.2, -20.0, 1.0, lo, hi, pass_)
    p.add_bandpass("bp2", 1.0, 0.0, 1.0, hi, lo, not pass_)
    p.add_normalize("out", 1.0, 0.0)
    p.connect("m", "bp")
    p.connect("l", "bp")
    p.connect("bp", "bp2")
    p.connect("bp2", "out")
    p.set_output("out")
    return p


@pytest.mark.parametrize("lo,hi,pass_", [(200.0, 4000.0, True), (20.0, 18000.0, False), (0.0, 50.0, True),
                                         (1000.0, 0.0, True), (5.0, 9000.0, True)])
@pytest.mark.parametrize("parallel", [1, 0])
def test_band_pass_parallel_is_exact(gpu_api, oracle, lo, hi, pass_, parallel):
    """The speculative-segment band-pass must reproduce the serial recurrence bit for bit, on inputs with
    silent and constant stretches, for cut-offs from 5 Hz (serial fallback) to 18 kHz."""
    p = _gappy_project(lo, hi, pass_)
    built = p.build(gpu_api)
    built[2].set_option("debug.band_serial", 0 if parallel else 1)   # (1: the serial kernel for every vertex -- the fallback of cut-offs below 5 Hz and of block pulls)
    got = p.render(gpu_api, built=built)
    assert_bit_exact(got, p.render(oracle))
    st = built[2].band_stats()
    if parallel and lo == 200.0:
        assert st["recomputed"] >= st["parked"] >= 0
    if not parallel:
        assert st == {"mismatched": 0, "recomputed": 0, "parked": 0}


def test_band_pass_parallel_chunked_state_carry(gpu_api, oracle):
    p = _gappy_project(300.0, 3000.0, True, seconds=3.0)
    built = p.build(gpu_api)
    built[2].set_option("max_chunk_frames", 30000)
    assert_bit_exact(p.render(gpu_api, built=built, scan=True), p.render(oracle, scan=True))


def test_config4_deep_chain_short(gpu_api, oracle):
    """256-vertex chain (sum / band-pass 20 Hz..18 kHz / adsr alternating) behind the wavetable voice and a
    resampled 44.1 kHz asset: every operation is IEEE-only -> bit-exact vs the oracle."""
    p = W.config4(seconds=1.5)
    assert sum(len(p.calls[k]) for k in p.calls if k.startswith("add_")) == 256
    assert_bit_exact(p.render(gpu_api), p.render(oracle))


def test_wavetable_voice(gpu_api, oracle):
    """Build-defined wavetable voice (stand-in for the un-vendored sampsyn crate): voice bookkeeping per
    extensions.rs:532-578, table lookups bilinear -- bit-exact HIP vs oracle, with and without a parsed table."""
    for table in (W.wavetable_bytes(3, 5, 64, 0.3), b"garbage -> default table"):
        p = W.synth_project(seconds=1.5)
        p.resources["t"] = table
        p.load_resource("t", "t")
        p.add_sampsyn("wt", 0.7, 25.0, "notes", W.NOTE_ADSR, "t")
        p.connect("wt", "mix")
        for scan in (False, True):
            assert_close(p.render(gpu_api, scan=scan), p.render(oracle, scan=scan))   # debug_sine/synth share the mix
        q = W.ProjectScript(48000, 512)
        q.set_length(1.0)
        q.event_files["n"] = p.event_files["notes"]
        q.load_midi_floww("n", "n")
        q.resources["t"] = table
        q.load_resource("t", "t")
        q.add_sampsyn("wt", 0.7, 25.0, "n", W.STD_ADSR, "t")
        q.add_normalize("out", 1.0, 0.0)
        q.connect("wt", "out")
        q.set_output("out")
        gb, ob = q.build(gpu_api), q.build(oracle)
        for _ in range(2):   # second render continues the voices: SampSyn notes are not cleared by set_time (Q4)
            assert_bit_exact(q.render(gpu_api, built=gb), q.render(oracle, built=ob))


def test_long_timeline_crosses_the_chunk_cap(gpu_api, oracle):
    """6 minutes @ 48 kHz = 17.3 M frames > the 2^24-frame edge-buffer chunk cap: the engine renders two chunks
    with carried state (loop cursors, running peak, band-pass state); the PCM must equal the oracle's."""
    p = W.ProjectScript(48000, 1024)
    p.set_length(360.0)
    p.assets["a"] = W.Asset(W.kick_int16(31, 30011))
    p.assets["b"] = W.Asset(W.noise_int16(32, 77777))
    p.load_sample("a", "a", "")
    p.load_sample("b", "b", "")
    p.add_sampleloop("la", 0.9, -20.0, "a")
    p.add_sampleloop("lb", 0.2, 35.0, "b")
    p.add_bandpass("bp", 1.0, 0.0, 1.0, 150.0, 6000.0, True)
    p.add_normalize("out", 1.0, 0.0)
    p.connect("la", "bp")
    p.connect("lb", "bp")
    p.connect("bp", "out")
    p.set_output("out")
    assert p.cs * 1024 > (1 << 24)
    gp, _ = p.render(gpu_api, want_f32=False)
    op, _ = p.render(oracle, want_f32=False)
    assert gp.shape == op.shape == (p.cs * 1024, 2)
    assert np.array_equal(gp, op)


@pytest.mark.parametrize("packed", [0, 1])
@pytest.mark.parametrize("lens", [(5, 7, 1001), (48000, 4099, 6), (4, 13, 64)])
def test_packed_16bit_sample_form_is_value_identical(gpu_api, oracle, packed, lens):
    """Inlined sources read the packed int16 form of the samples (half the gather bytes); (float)int * scale is
    how the f32 bank entry was made, so the render must not change by a bit -- including loop lengths that are
    not multiples of 4, shorter than a lane's four frames, and every per-channel-scale load mode."""
    p = W.ProjectScript(48000, 1024)
    p.set_length(0.4)
    modes = ["", "normalize-seperate", "left", "loudest"]
    for i, n in enumerate(lens):
        p.assets["a%d" % i] = W.Asset(W.noise_int16(40 + i, n))
        p.load_sample("a%d" % i, "a%d" % i, modes[i % len(modes)])
        p.add_sampleloop("l%d" % i, 0.3 + 0.4 * i, -60.0 + 50.0 * i, "a%d" % i)
    p.assets["mix"] = W.Asset(W.noise_int16(50, 333))
    p.load_sample("mix", "mix", "mix-down")            # no packed form: exercises the mixed-term path
    p.add_sampleloop("lm", 0.5, 0.0, "mix")
    p.add_normalize("out", 1.0, 0.0)
    p.add_sum("side", 1.0, 0.0)
    for i in range(len(lens)):
        p.connect("l%d" % i, "out")
        p.connect("l%d" % i, "side")
    p.connect("lm", "side")
    p.connect("side", "out")
    p.set_output("out")
    built = p.build(gpu_api)
    built[2].set_option("packed_samples", packed)
    for scan in (False, True):
        assert_bit_exact(p.render(gpu_api, built=built, scan=scan), p.render(oracle, scan=scan))


def test_stream_workflow_block_pull(gpu_api, oracle):
    """stream_workflow.rs:62-101 without the audio device: events arrive in packets between block pulls
    (trim_streams, append, set_time(graph time)), blocks are pulled one at a time.  Drum-driven vertices
    (sample_multi -> adsr vertex) stay bit-exact against the oracle doing the same."""
    rng = np.random.default_rng(21)
    banks = []
    for be in (gpu_api, oracle):
        sb = be.SampleBank(48000)
        sb.add_decoded("kick", W.kick_int16(12, 9000).astype(np.float32).reshape(-1), 2, 48000, 16, "")
        fb = be.FlowwBank(48000, 512)
        fb.declare_stream("live")
        g = be.Graph(512, 48000)
        g.add_sample_multi("hits", 0.9, 10.0, sb.get_index("kick"), fb.get_index("live"), -1)
        g.add_adsr("env", 1.0, 0.0, 1.0, fb.get_index("live"), False, True, -1, [0.01, 0.05, 0.7, 0.05, 0.2, 0.02])
        g.add_sum("out", 1.0, 0.0)
        assert g.connect("hits", "env") and g.connect("env", "out") and g.set_output("out") and g.check_graph()
        banks.append((sb, fb, g))
    (gsb, gfb, gg), (osb, ofb, og) = banks
    clock = 0.0
    for b in range(60):
        if b % 3 == 0:   # a packet: a few hits scheduled a little ahead of the playhead
            now = b * 512 / 48000.0
            batch = []
            clock = max(clock, now)
            for _ in range(int(rng.integers(1, 4))):
                clock += float(rng.uniform(0.001, 0.02))
                batch.append((clock, 36.0, float(rng.uniform(0.3, 1.0))))
            for fb, g in ((gfb, gg), (ofb, og)):
                fb.trim_streams()
                fb.append_stream("live", batch)
                fb.set_time(g.get_time())
        for fb, g in ((gfb, gg), (ofb, og)):
            fb.set_time(g.get_time())   # stream_workflow.rs:91-92
        gl, gr = gg.render(gsb, gfb)
        ol, orr = og.render(osb, ofb)
        assert np.array_equal(_bits(gl), _bits(ol)) and np.array_equal(_bits(gr), _bits(orr)), "block %d" % b
        gfb.set_time_to_next_block()
        ofb.set_time_to_next_block()
    assert np.abs(gl).max() >= 0.0 and gg.get_time() == og.get_time() == 60 * 512


def test_project_with_midi_file(gpu_api, oracle, tmp_path):
    """load_midi_floww with a real .mid through the project front-end: the library parses the SMF itself;
    the oracle is handed the event list the file stands for (W.midi_bytes' quantised times)."""
    ev = [(0.1 * i + 0.003, 60 + (i % 5), 0.4 + 0.1 * (i % 4)) for i in range(12)]
    data, quant = W.midi_bytes(ev, ppq=480, us_per_quarter=500000)
    p = W.drum_project(seconds=1.5)
    name = sorted(p.event_files)[0]
    p.event_files[name] = quant
    want = p.render(oracle)
    d = tmp_path / "proj"
    lua = p.to_lua(str(d))
    old = [l for l in lua.splitlines() if l.startswith("load_midi_floww(\"%s\"" % name)][0]
    (d / "song.mid").write_bytes(data)
    lua = lua.replace(old, 'load_midi_floww("%s", "%s");' % (name, d / "song.mid"))
    s = gpu_api.State("", p.psr, p.bl)
    s.set_option("band_mode", 0)   # (the front-end's default is scan mode: this test compares bytes)
    assert s.refresh(lua), gpu_api.last_error()
    pcm = s.render_to_memory()
    assert np.array_equal(pcm, want[0])


def _stutter_project(seconds, spacing, lo, hi, seed):
    """Thousands of very short sounds separated by exact silences and, through a sample_lerp, by held DC
    levels: every gap is an event for the band-pass repair (k_band_fix), far more than one round or one
    workgroup slice handles."""
    rng = np.random.default_rng(seed)
    p = W.ProjectScript(48000, 1024)
    p.set_length(seconds)
    p.assets["tick"] = W.Asset(W.tone_int16(31, 96))
    p.assets["blip"] = W.Asset(W.kick_int16(32, 200) + np.int16(5))
    p.load_sample("tick", "tick", "")
    p.load_sample("blip", "blip", "")
    t, a = 0.001, []
    while t < seconds:
        a.append((t, 60.0, float(rng.uniform(0.2, 1.0))))
        t += spacing * float(rng.uniform(0.6, 1.6))
    b = [(float(x), 36.0, float(rng.uniform(0.2, 1.0))) for x in np.sort(rng.uniform(0.0, seconds, int(seconds / (4 * spacing))))]
    p.event_files["a"] = np.array(a, np.float32)
    p.event_files["b"] = np.array(b, np.float32)
    p.load_midi_floww("a", "a")
    p.load_midi_floww("b", "b")
    p.add_sample_multi("m", 1.0, 0.0, "tick", "a", -1)
    p.add_sample_lerp("l", 0.5, 30.0, "blip", "b", -1, 16)
    p.add_bandpass("bp", 1.1, 15.0, 1.0, lo, hi, True)
    p.add_bandpass("bq", 1.0, 0.0, 1.0, lo, hi, False)
    p.add_sum("mix", 1.0, 0.0)
    p.add_normalize("out", 1.0, 0.0)
    p.connect("m", "bp")
    p.connect("l", "bq")
    p.connect("m", "bq")
    p.connect("bp", "mix")
    p.connect("bq", "mix")
    p.connect("mix", "out")
    p.set_output("out")
    return p


@pytest.mark.parametrize("seconds,spacing,lo,hi", [(30.0, 0.02, 3000.0, 9000.0), (12.0, 0.03, 1000.0, 0.0),
                                                   (20.0, 0.05, 1500.0, 12000.0)])
def test_band_pass_repair_storm(gpu_api, oracle, seconds, spacing, lo, hi):
    """Many more repair events than k_band_fix handles per round (2048) or per slice: the optimistic parallel
    repair has to converge to the serial result through several rounds and across slice borders."""
    p = _stutter_project(seconds, spacing, lo, hi, seed=int(seconds))
    built = p.build(gpu_api)
    obuilt = p.build(oracle)
    got = p.render(gpu_api, built=built)
    st = built[2].band_stats()
    assert_bit_exact(got, p.render(oracle, built=obuilt))
    assert st["mismatched"] > 100, st
    # second render of the same handles (carried lerp / band state, quirk Q4): same answer again
    assert_bit_exact(p.render(gpu_api, built=built), p.render(oracle, built=obuilt))


def _soak_case(seed):
    """One random stutter project: cut-offs from 0 (smoother off) over 30 Hz (block-response guess) to 18 kHz, gap spacings
    from 4 ms to 0.3 s, lengths 3 - 20 s, sometimes a chunk cap."""
    rng = np.random.default_rng(1000 + seed)
    seconds = float(rng.choice([3.0, 8.0, 20.0]))
    spacing = float(rng.choice([0.004, 0.01, 0.03, 0.08, 0.3]))
    lo = float(rng.choice([0.0, 30.0, 200.0, 1000.0, 4000.0, 9000.0]))
    hi = float(rng.choice([0.0, 60.0, 500.0, 3000.0, 12000.0, 18000.0]))
    if lo == 0.0 and hi == 0.0:
        hi = 700.0
    chunk = int(rng.choice([65536, 300000, 1 << 20])) if rng.random() < 0.3 else 0
    return _stutter_project(seconds, spacing, lo, hi, seed), chunk


@pytest.mark.parametrize("seed", range(12))
def test_band_pass_random_soak(gpu_api, oracle, seed):
    """Randomised soak of the speculative band-pass (k_band_spec / k_band_fix and its fill phase) against the oracle, bit for
    bit, fresh and scanned (tools/band_soak.py runs the same cases over any seed range)."""
    p, chunk = _soak_case(seed)
    gb, ob = p.build(gpu_api), p.build(oracle)
    if chunk:
        gb[2].set_option("max_chunk_frames", chunk)
    for scan in (False, True):
        assert_bit_exact(p.render(gpu_api, built=gb, scan=scan), p.render(oracle, built=ob, scan=scan))


@pytest.mark.parametrize("bits,seconds,bl", [(16, 40.0, 1000), (24, 40.0, 1000), (16, 58.0, 1024), (24, 58.0, 1024)])
def test_wide_loop_sums_on_long_timelines(gpu_api, oracle, bits, seconds, bl):
    """Timelines of >= 1800 / >= 2600 tiles switch the all-loop sums to 8 / 16 consecutive frames per lane
    (k_sum16w): plain Sum vertices (any block length, partial last tile) and Normalize pass A (bl = 1024), packed
    16-bit sources and f32 ones (24-bit assets have no packed form)."""
    p = W.ProjectScript(48000, bl)
    p.set_length(seconds)
    for k in range(5):
        pcm = W.noise_int16(700 + k, 3001 + 517 * k)
        if bits == 24:
            pcm = pcm.astype(np.int32) * 256 + (k + 1)
        p.assets["s%d" % k] = W.Asset(pcm, bits=bits)
        p.load_sample("s%d" % k, "s%d" % k, "")
        p.add_sampleloop("v%d" % k, 0.3 + 0.2 * k, -60.0 + 30.0 * k, "s%d" % k)
    p.add_sum("mix", 0.9, 12.0)            # mode 0 over loop sources
    p.add_normalize("out", 1.0, 0.0)       # pass A over two loop sources + (edge) mix -> mixed mode ...
    p.add_normalize("loops", 1.0, 0.0)     # ... and one over loop sources only
    for k in range(5):
        p.connect("v%d" % k, "mix")
    for k in range(3):
        p.connect("v%d" % k, "loops")
    p.add_sum("final", 1.0, 0.0)
    p.connect("mix", "out")
    p.connect("out", "final")
    p.connect("loops", "final")
    p.set_output("final")
    gb, ob = p.build(gpu_api), p.build(oracle)
    for scan in (False, True):
        assert_bit_exact(p.render(gpu_api, built=gb, scan=scan), p.render(oracle, built=ob, scan=scan))


@pytest.mark.parametrize("lens", [(1, 2, 3, 255, 256, 257, 1000), (64, 128, 511, 512, 513, 40001)])
def test_wide_loop_sums_over_loops_shorter_than_a_quad_step(gpu_api, oracle, lens):
    """k_sum16w<4> (timelines of >= 2 600 tiles) gives a lane four quads 256 frames apart; the index of a quad is the one
    before + 256 mod len, wrapped once (kernels.hip step256) -- loops of 1 .. 257 frames, where 256 mod len is not 256, an odd
    and an even number of sources (the four-source batches' tail), into a Normalize (mode 5: the fused single-pass form, the
    PCM straight out of the registers) and a plain Sum."""
    p = W.ProjectScript(48000, 1024)
    p.set_length(58.0)
    for k, n in enumerate(lens):
        p.assets["s%d" % k] = W.Asset(W.noise_int16(4100 + k, n))
        p.load_sample("s%d" % k, "s%d" % k, "")
        p.add_sampleloop("v%d" % k, 0.4 + 0.15 * k, -70.0 + 25.0 * k, "s%d" % k)
    p.add_normalize("out", 1.0, 0.0)
    p.add_sum("mix", 0.7, 20.0)
    for k in range(len(lens)):
        p.connect("v%d" % k, "out")
        p.connect("v%d" % k, "mix")
    p.add_sum("final", 1.0, 0.0)
    p.connect("out", "final")
    p.connect("mix", "final")
    p.set_output("final")
    gb, ob = p.build(gpu_api), p.build(oracle)
    for scan in (False, True):
        assert_bit_exact(p.render(gpu_api, built=gb, scan=scan), p.render(oracle, built=ob, scan=scan))
    # ... and as the output vertex itself: the launch's own quantiser
    p2 = W.ProjectScript(48000, 1024)
    p2.set_length(58.0)
    for k, n in enumerate(lens):
        p2.assets["s%d" % k] = W.Asset(W.noise_int16(4100 + k, n))
        p2.load_sample("s%d" % k, "s%d" % k, "")
        p2.add_sampleloop("v%d" % k, 0.4 + 0.15 * k, -70.0 + 25.0 * k, "s%d" % k)
    p2.add_normalize("out", 1.0, 0.0)
    for k in range(len(lens)):
        p2.connect("v%d" % k, "out")
    p2.set_output("out")
    assert_bit_exact(p2.render(gpu_api), p2.render(oracle))
