"""GPU parity of the TOLERANCE-class band-pass (`band_mode` 1, k_band_scan): band_pass_gen
(/root/reference/src/extensions.rs:654-689) evaluated as a blocked affine scan, one launch per vertex.

Bar (BASELINE.json north_star, filter paths): <= 1e-6 RMS on the f32 output and +-1 LSB on the PCM against the
oracle's serial f32 recurrence.  The exact kernels (`band_mode` 0, the default) stay the bit-exact parity mode and are
tested in test_gpu_parity.py / test_gpu_full_size.py; this file runs the same projects through the scan.
"""
import numpy as np
import pytest

from termdaw_amd import workloads as W
from test_gpu_parity import _bits, _gappy_project, _soak_case, _stutter_project, assert_bit_exact, assert_close

pytestmark = pytest.mark.gpu


def _scan_build(p, api, **opts):
    built = p.build(api)
    built[2].set_option("band_mode", 1)
    for k, v in opts.items():   # (keyword names other than the supported options stand for the header's "debug." test hooks)
        built[2].set_option(k if k in ("max_chunk_frames", "band_guard_ppb", "output_f32") else "debug." + k, v)
    return built


def _rms(a, b):
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


@pytest.mark.parametrize("nf", [16, 8])
def test_config3_short_scan(gpu_api, oracle, nf):
    p = W.config3(seconds=4.0)
    got = p.render(gpu_api, built=_scan_build(p, gpu_api, band_scan_nf=nf))
    assert_close(got, p.render(oracle))


@pytest.mark.parametrize("lo,hi,pass_", [(200.0, 4000.0, True), (20.0, 18000.0, False), (0.0, 50.0, True),
                                         (1000.0, 0.0, True), (5.0, 9000.0, True), (1.0, 300.0, True)])
@pytest.mark.parametrize("scan", [False, True])
def test_gappy_inputs_scan(gpu_api, oracle, lo, hi, pass_, scan):
    """Silent and held-constant stretches (where the exact kernels park), cut-offs from 0 (constant chain) and 1 Hz
    (beyond the scan's look-back depth: the vertex then takes the exact kernels) to 18 kHz; two band-pass vertices in a
    row, fresh and after a normalize scan (which carries the filter state into the render, quirk Q4)."""
    p = _gappy_project(lo, hi, pass_)
    built = _scan_build(p, gpu_api)
    assert_close(p.render(gpu_api, built=built, scan=scan), p.render(oracle, scan=scan))


def test_scan_matches_exact_mode_closely(gpu_api):
    """The two modes of one engine on one project: the scan differs from the exact kernels by the f32 trajectory's own
    rounding only (a few 1e-8 relative)."""
    p = _gappy_project(200.0, 4000.0, True, seconds=3.0)
    exact = p.render(gpu_api)
    scan = p.render(gpu_api, built=_scan_build(p, gpu_api))
    assert _rms(exact[1], scan[1]) <= 3e-7
    assert np.abs(exact[0].astype(np.int64) - scan[0].astype(np.int64)).max() <= 1


@pytest.mark.parametrize("nf", [16, 8])
def test_recomputed_predecessors_give_the_same_bits(gpu_api, nf):
    """`band_scan_debug`: every poll of a predecessor's granules times out at once, so every tile recomputes the
    responses of its look-back window itself (the bounded-wait fallback).  Same arithmetic -> the same bits as the
    published values give."""
    p = _gappy_project(20.0, 9000.0, True, seconds=3.0)
    normal = p.render(gpu_api, built=_scan_build(p, gpu_api, band_scan_nf=nf))
    forced = p.render(gpu_api, built=_scan_build(p, gpu_api, band_scan_nf=nf, band_scan=1))
    assert np.array_equal(_bits(normal[1]), _bits(forced[1]))
    assert np.array_equal(normal[0], forced[0])


@pytest.mark.parametrize("chunk", [1024, 5000, 30000])
def test_chunked_scan_carries_the_filter_state(gpu_api, oracle, chunk):
    """Chunks of one block (a block pull's size), of a few blocks and of a few tiles: the state crosses chunk borders in
    the same f32 slot the exact kernels use."""
    p = _gappy_project(300.0, 3000.0, True, seconds=3.0)
    built = _scan_build(p, gpu_api, max_chunk_frames=chunk)
    assert_close(p.render(gpu_api, built=built, scan=True), p.render(oracle, scan=True))


def test_render_twice_and_mode_switch(gpu_api, oracle):
    """Second render of the same handles (carried state), then the same handles switched back to the exact kernels:
    the state slot is shared, so the exact render continues from the scan's state -- still within tolerance."""
    p = _stutter_project(6.0, 0.05, 1500.0, 12000.0, seed=3)
    gb, ob = _scan_build(p, gpu_api), p.build(oracle)
    for _ in range(2):
        assert_close(p.render(gpu_api, built=gb), p.render(oracle, built=ob))
    gb[2].set_option("band_mode", 0)
    assert_close(p.render(gpu_api, built=gb), p.render(oracle, built=ob))


def test_block_pulls_in_scan_mode(gpu_api, oracle):
    """td_graph_render_block (Graph::render, graph.rs:182-193) with a band-pass in scan mode: one-tile launches."""
    p = _gappy_project(250.0, 5000.0, True, seconds=0.4)
    gsb, gfb, gg = _scan_build(p, gpu_api)
    osb, ofb, og = p.build(oracle)
    for b in range(p.cs):
        gl, gr = gg.render(gsb, gfb)
        ol, orr = og.render(osb, ofb)
        # (relative to the block's own level, down to -120 dB: in a pure decay after the input has stopped, the state
        # carried from pull to pull keeps whatever relative rounding difference it has, at 1e-20 of full scale)
        scale = max(1e-6, float(np.abs(ol).max()), float(np.abs(orr).max()))
        assert _rms(gl, ol) <= 1e-6 * scale and _rms(gr, orr) <= 1e-6 * scale, "block %d" % b
        gfb.set_time_to_next_block()
        ofb.set_time_to_next_block()


@pytest.mark.parametrize("seed", range(6))
def test_random_soak_scan(gpu_api, oracle, seed):
    p, chunk = _soak_case(seed)
    gb, ob = _scan_build(p, gpu_api), p.build(oracle)
    if chunk:
        gb[2].set_option("max_chunk_frames", chunk)
    for scan in (False, True):
        assert_close(p.render(gpu_api, built=gb, scan=scan), p.render(oracle, built=ob, scan=scan))


def test_config4_deep_chain_short_scan(gpu_api, oracle):
    """The 256-vertex chain: 84 scan launches, each evaluating `sum stage -> adsr` of its input as terms."""
    p = W.config4(seconds=3.0)
    built = _scan_build(p, gpu_api)
    assert_close(p.render(gpu_api, built=built), p.render(oracle))


@pytest.mark.parametrize("depth", [12, 84, 252])
def test_rms_per_chain_depth(gpu_api, oracle, depth):
    """Error growth along the chain (reported by tools/band_scan_rms.py for DESIGN.md): within budget at every depth of
    config 4."""
    p = W.config4(seconds=4.0, depth=depth)
    got, ref = p.render(gpu_api, built=_scan_build(p, gpu_api)), p.render(oracle)
    assert_close(got, ref)


def test_config3_full_60s_scan(gpu_api, oracle):
    p = W.config3()
    assert p.cs == 2813
    assert_close(p.render(gpu_api, built=_scan_build(p, gpu_api)), p.render(oracle))


def test_config4_full_60s_scan(gpu_api, oracle):
    p = W.config4()
    assert p.cs == 2813
    assert_close(p.render(gpu_api, built=_scan_build(p, gpu_api)), p.render(oracle))


def test_scan_mode_leaves_bit_exact_kinds_alone(gpu_api, oracle):
    """`band_mode` only changes band-pass vertices: a project without one is bit-exact in either mode."""
    p = W.drum_project(seconds=1.0)
    has_band = len(p.calls.get("add_bandpass", [])) > 0
    got = p.render(gpu_api, built=_scan_build(p, gpu_api))
    (assert_close if has_band else assert_bit_exact)(got, p.render(oracle))


def _chain_project(shapes, seconds=2.0, lo=60.0, hi=9000.0, pass_=True):
    """noise loop -> band-pass vertices linked as `shapes` says: "" (direct), "s" (gain / pan stage), "a" (Adsr vertex),
    "sa", "as", "sas" -- single-input, single-consumer links, what k_band_chain runs in one launch."""
    p = W.ProjectScript(48000, 1024)
    p.set_length(seconds)
    p.assets["n"] = W.Asset(W.noise_int16(9, 50021))
    p.load_sample("n", "n", "")
    p.event_files["g"] = np.array([(0.21 * i + 0.03, 60.0, 0.8) for i in range(int(seconds / 0.21) + 1)], np.float32)
    p.load_midi_floww("g", "g")
    p.add_sampleloop("src", 0.7, 10.0, "n")
    p.add_bandpass("b0", 1.0, 0.0, 1.0, lo, hi, pass_)
    p.connect("src", "b0")
    prev = "b0"
    for i, shape in enumerate(shapes):
        for j, ch in enumerate(shape):
            name = "l%d_%d" % (i, j)
            if ch == "s":
                p.add_sum(name, 1.3, -12.0 if i % 2 else 7.0)
            else:
                p.add_adsr(name, 1.0, 0.0, 0.6, "g", False, True, -1, [0.01, 0.05, 0.7, 0.05, 0.3, 0.02])
            p.connect(prev, name)
            prev = name
        name = "b%d" % (i + 1)
        p.add_bandpass(name, 1.1 if i % 3 == 0 else 1.0, 3.0 if i % 3 == 1 else 0.0, 1.0, lo * (1 + i % 4), hi, pass_)
        p.connect(prev, name)
        prev = name
    p.add_normalize("out", 1.0, 0.0)
    p.connect(prev, "out")
    p.set_output("out")
    return p


@pytest.mark.parametrize("chunk", [0, 7000])
def test_chain_link_shapes(gpu_api, oracle, chunk):
    """Every link shape between two band-pass vertices of a chain, fresh, chunked (state carried per stage) and rendered
    twice; the same project with chains switched off (one launch per vertex) agrees with it to rounding."""
    p = _chain_project(["", "s", "a", "sa", "as", "sas", "", "a"])
    gb, ob = _scan_build(p, gpu_api), p.build(oracle)
    if chunk:
        gb[2].set_option("max_chunk_frames", chunk)
    for _ in range(2):
        got, ref = p.render(gpu_api, built=gb), p.render(oracle, built=ob)
        assert_close(got, ref)
    single = p.render(gpu_api, built=_scan_build(p, gpu_api, band_chain=0))
    assert_close(single, p.render(oracle))
    assert _rms(single[1], p.render(gpu_api, built=_scan_build(p, gpu_api))[1]) <= 2e-7


def test_chain_longer_than_one_launch(gpu_api, oracle):
    """140 band-pass vertices in a row: the chain is cut at kScanMaxStages (128), the cut vertex is materialised."""
    p = _chain_project([""] * 139, seconds=0.5, lo=300.0, hi=12000.0)
    assert_close(p.render(gpu_api, built=_scan_build(p, gpu_api)), p.render(oracle))


def test_cut_vertex_in_a_chain_is_its_own_launch(gpu_api, oracle):
    """`pass` false vertices (whose right output needs the right-channel smoothers) do not join chains."""
    p = _chain_project(["s", "a", ""], pass_=False)
    assert_close(p.render(gpu_api, built=_scan_build(p, gpu_api)), p.render(oracle))


def test_chain_after_set_time_reseeds_every_stage(gpu_api, oracle):
    """set_time raises every band-pass vertex' `first` flag (extensions.rs:196-204): the next render seeds each stage's
    smoothers from its own first input frame (extensions.rs:664-670), in a chain too."""
    p = _chain_project(["s", "a", "sa"], seconds=1.0)
    gb, ob = _scan_build(p, gpu_api), p.build(oracle)
    assert_close(p.render(gpu_api, built=gb), p.render(oracle, built=ob))
    for b in (gb, ob):
        b[1].set_time(0)
        b[2].set_time(0)
    assert_close(p.render(gpu_api, built=gb), p.render(oracle, built=ob))


@pytest.mark.parametrize("which", ["config3", "config4", "chunked", "twice"])
def test_normalize_behind_a_scan_launch_is_value_identical(gpu_api, oracle, which):
    """A Normalize vertex right behind a scan launch is evaluated by that launch's epilogue (BandScanDesc::norm, engine option
    `fuse_normalize`, default 1): the same operations on the same values as the launch of its own it replaces -- behind a chain
    bit for bit the unfused result, and always within the mode's bar of the oracle; the carried max survives chunk boundaries and renders."""
    if which == "config3":
        p, chunk = W.config3(seconds=6.0), 0
    elif which == "config4":
        p, chunk = W.config4(seconds=5.0, depth=24), 0
    else:
        p, chunk = W.config4(seconds=7.0, depth=12), (20 * 1024 if which == "chunked" else 0)
    outs = []
    for fuse in (1, 0):
        built = _scan_build(p, gpu_api, fuse_normalize=fuse)
        if chunk:
            built[2].set_option("max_chunk_frames", chunk)
        built[2].set_profiling(1)
        got = p.render(gpu_api, built=built)
        if which == "twice":   # (a second render without a reset: one-shot state and the running max carry over, quirk Q4)
            got = p.render(gpu_api, built=built)
        fam = set(built[2].kernel_times())
        built[2].set_profiling(0)
        if which == "config3":   # (synth -> adsr -> band-pass -> normalize: the Normalize vertex' own launch is the only k_sum)
            assert ("k_sum" in fam) == (fuse == 0), fam
        outs.append(got)
    if which == "config3":   # (unfused, a single band-pass vertex takes k_band_scan, whose zero-state runs are in double: other roundings)
        assert _rms(outs[0][1], outs[1][1]) <= 1e-7 and np.abs(outs[0][0].astype(np.int64) - outs[1][0].astype(np.int64)).max() <= 1
    else:
        assert np.array_equal(outs[0][0], outs[1][0])
        assert np.array_equal(_bits(outs[0][1]), _bits(outs[1][1]))
    if which != "twice":
        assert_close(outs[0], p.render(oracle))


def _nan_burst_project(shape, bl, seconds=3.0):
    """Noise, plus ONE drum hit at 0.5 s amplified beyond the f32 range (inf on the hit's loud frames): in front of band-pass
    vertices.  The smoothers' state goes inf, then inf - inf = NaN on the next frame, and stays NaN: finite output before the
    hit, NaN from there to the end -- although the input is finite again a few hundred frames later."""
    p = W.ProjectScript(48000, bl)
    p.set_length(seconds)
    p.assets["n"] = W.Asset(W.noise_int16(3, 700))
    p.assets["k"] = W.Asset(W.kick_int16(4, 3000))
    p.load_sample("n", "n", "")
    p.load_sample("k", "k", "")
    p.event_files["h"] = np.array([(0.5, 60.0, 0.9)], np.float32).reshape(-1, 3)
    p.load_midi_floww("h", "h")
    p.add_sampleloop("a", 0.5, 0.0, "n")
    p.add_sample_multi("hit", 3.0e38, 0.0, "k", "h", -1)
    p.add_sum("boost", 1.0e6, 0.0)
    p.connect("hit", "boost")
    p.add_sum("mix", 1.0, 0.0)
    p.connect("a", "mix")
    p.connect("boost", "mix")
    prev = "mix"
    for i, pass_ in enumerate(shape):
        p.add_bandpass("b%d" % i, 1.0, 0.0, 1.0, 300.0, 6000.0, pass_)
        p.connect(prev, "b%d" % i)
        prev = "b%d" % i
    p.add_normalize("out", 1.0, 0.0)
    p.connect(prev, "out")
    p.set_output("out")
    return p


@pytest.mark.parametrize("shape", [(True,), (False,), (True, True, True), (True, False)])
@pytest.mark.parametrize("bl,chunk", [(1024, 0), (256, 0), (1024, 40 * 1024)])
def test_a_non_finite_state_stays_non_finite(gpu_api, oracle, shape, bl, chunk):
    """The reference's smoother never recovers from a NaN (y + gamma (x - y) of a NaN is a NaN): every output frame after the
    first poisoned one is NaN.  The scan's look-back forgets a tile after K tiles (2 here), so the launches end with a
    gather over ALL earlier tiles (BandScanDesc::poison): chain kernel (`pass` vertices, alone or in a chain, with and
    without the Normalize vertex in the launch) and k_band_scan (not `pass`), one chunk and several (the carried states)."""
    p = _nan_burst_project(shape, bl)
    built = _scan_build(p, gpu_api)
    if chunk:
        built[2].set_option("max_chunk_frames", chunk)
    gp, gf = p.render(gpu_api, built=built)
    op, of = p.render(oracle)
    assert np.isnan(of).sum() > of.size // 2 and np.isfinite(of).sum() > of.size // 8   # (the case is what it claims to be)
    assert np.abs(of[np.isfinite(of)]).max() > 0.1
    assert np.array_equal(np.isnan(gf), np.isnan(of))
    ok = np.isfinite(of)
    assert _rms(gf[ok], of[ok]) <= 1e-6 and np.abs(gp.astype(np.int64) - op.astype(np.int64)).max() <= 1


@pytest.mark.parametrize("silent", [0, 1])
@pytest.mark.parametrize("shape", [(False,), (True,), (False, False), (True, False)])
def test_a_non_finite_state_of_one_channel_stays_in_that_channel(gpu_api, oracle, silent, shape):
    """The burst that overflows the smoothers lives in ONE channel (the other channel of the asset is silent).  The
    reference's outputs are  cutl cut_mul + (l - cutl) pass_mul  and  cutr cut_mul + (r - cutl) pass_mul  with one factor 0
    and the other 1 (extensions.rs:682-687): a NaN times 0 is a NaN, so a non-finite LEFT pair of smoothers turns both
    outputs NaN whatever the vertex passes, a non-finite RIGHT pair the right output only -- also of a `pass` vertex, whose
    right smoothers reach its output in no other way.  NaN masks equal the oracle's per channel, the rest stays in class."""
    p = _nan_burst_project(shape, 1024)
    k = p.assets["k"].pcm.copy()
    k[:, silent] = 0
    p.assets["k"] = W.Asset(k)
    built = _scan_build(p, gpu_api)
    obuilt = p.build(oracle)
    for rep in range(2):   # (the second render starts from the carried -- partly NaN -- filter states)
        gp, gf = p.render(gpu_api, built=built)
        op, of = p.render(oracle, built=obuilt)
        assert np.array_equal(np.isnan(gf), np.isnan(of)), "render %d" % rep
        if rep == 0:   # (the case is what it claims to be)
            assert np.isnan(of[:, 1]).any() and np.isnan(of[:, 0]).any() == (silent == 1)
        ok = np.isfinite(of)
        assert _rms(gf[ok], of[ok]) <= 1e-6 and np.abs(gp.astype(np.int64) - op.astype(np.int64)).max() <= 1


def test_a_grid_longer_than_the_device_holds_stays_in_scan_mode(gpu_api, oracle):
    """k_band_scan and k_band_chain both number their tiles by a ticket drawn at start: a workgroup only waits for lower
    tickets, whose holders are running, so neither kernel depends on its grid being resident at once -- a 140 s chunk (1 641
    tiles of 4 096 frames, several times what the device holds) of a vertex that is not `pass` (k_band_scan: look-back with
    bounded waits, the all-earlier-tiles gather at the end) and of a `pass` vertex (chain kernel) both stay in scan mode,
    inside the tolerance class."""
    for pass_ in (False, True):
        p = W.ProjectScript(48000, 1024)
        p.set_length(140.0)
        p.assets["n"] = W.Asset(W.noise_int16(9, 30011))
        p.load_sample("n", "n", "")
        p.add_sampleloop("a", 0.5, 10.0, "n")
        p.add_bandpass("bp", 1.2, -20.0, 1.0, 200.0, 4000.0, pass_)
        p.add_normalize("out", 1.0, 0.0)
        p.connect("a", "bp")
        p.connect("bp", "out")
        p.set_output("out")
        built = _scan_build(p, gpu_api)
        built[2].set_profiling(1)
        got = p.render(gpu_api, built=built)
        fam = set(built[2].kernel_times())
        built[2].set_profiling(0)
        want = p.render(oracle)
        assert "k_band_spec" not in fam and "k_band_scan" in fam, fam
        assert_close(got, want)
