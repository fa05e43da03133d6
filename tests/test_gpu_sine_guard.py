"""The sine kinds under the guard: engine option "sine_mode" 2, the front-end's default (include/termdaw_amd.h td_graph_set_option;
kernels.h ProbeDesc).

debug_sine_gen / synth_gen evaluate `f32::sin` per voice and frame (/root/reference/src/extensions.rs:450-452,501-520,
synth.rs:21-29).  Their fast device forms (sin_any, affine envelopes, folded products) differ from the reference's own arithmetic
by a white rounding noise of ~3e-8 of the vertex' level, which a graph can make as large as it likes: round 5's soak held one
render at 2.6e-6 RMS (seed 123475: 43 dB of cancellation in a `cut` band-pass, then a Normalize vertex).  Mode 2 MEASURES that
noise -- k_sine_probe evaluates a sample of every chunk's frames the reference's way and compares with what the fast launch
left -- carries it to the output like the band-pass guard's estimate (static gains, the Normalize vertex' running 1 / max), and
over 2e-7 renders again with glibc's sinf on the device (mode 1's form: the oracle's bits).  Here: the probe's figure against the
true deviation at the vertex; the BASELINE config passes untouched; the round-5 outlier is inside the bound through the front-end
with nothing set; a forced verdict gives the oracle's bits in every calling pattern."""
import numpy as np
import pytest

from termdaw_amd import workloads as W
import test_gpu_fuzz as F

pytestmark = pytest.mark.gpu


def _rms(a, b):
    ok = np.isfinite(b)
    if not ok.any():
        return 0.0
    return float(np.sqrt(np.mean((a[ok].astype(np.float64) - b[ok].astype(np.float64)) ** 2)))


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _front_end_modes(p, api, **opts):
    built = p.build(api)
    built[2].set_option("band_mode", 2)
    built[2].set_option("sine_mode", 2)
    for k, v in opts.items():
        built[2].set_option(k, v)
    return built


@pytest.mark.parametrize("seed", [37, 19, 67, 29, 123475, 7, 11])
def test_the_probe_measures_the_deviation_at_the_vertex(gpu_api, oracle, seed):
    """Every debug_sine / synth vertex of a random project made the output vertex in turn -- no gain, no Normalize vertex behind
    it: the audit's estimate is then the probe's own figure, the RMS over its sample frames of the larger channel's distance
    between the fast launch's frames and the reference's.  Against the same quantity over ALL frames (the oracle's frames are
    mode 1's, bit for bit: tests/test_gpu_sine_exact.py): within the sampling error of a few hundred samples."""
    p = F.random_project(seed, allow_sinf=True)
    names = [c[0] for k in ("add_debug_sine", "add_synth") for c in p.calls.get(k, [])]
    assert names
    for n in names:
        p.output_vertex = n
        ref = p.render(oracle)[1].astype(np.float64)
        gb = p.build(gpu_api)
        gb[2].set_option("sine_mode", 2)
        gb[2].set_option("band_guard_ppb", 1000000000)   # (never done again: the fast frames are what is compared)
        f = p.render(gpu_api, built=gb)[1].astype(np.float64)
        st = gb[2].band_guard_stats()
        assert st["audits"] == 1 and st["redos"] == 0, st
        assert np.array_equal(np.isfinite(f), np.isfinite(ref))
        d2 = np.where(np.isfinite(ref), (f - ref) ** 2, 0.0).max(axis=1)
        true = float(np.sqrt(d2.mean()))
        assert 0.6 * true <= st["last_est"] <= 1.6 * true + 1e-12, (seed, n, true, st)
        assert true < 2e-7          # (of the vertex' own scale: the class' bar with room -- what a graph makes of it is the guard's business)


def test_config3_passes_untouched(gpu_api, oracle):
    """BASELINE config 3 (32-voice Synth -> Adsr -> band-pass -> Normalize) at the full 60 s in the front-end's modes: probed,
    audited by the chain launch itself (no audit launch), never rendered twice; the fast frames are what sine_mode 0 renders."""
    p = W.config3()
    assert p.cs == 2813
    ref_pcm, ref_f = p.render(oracle)
    gb = _front_end_modes(p, gpu_api)
    pcm, f = p.render(gpu_api, built=gb)
    st = gb[2].band_guard_stats()
    assert st["audits"] == 1 and st["redos"] == 0 and 1e-8 < st["last_est"] < 1.5e-7, st
    assert _rms(f, ref_f) <= 1e-6 and st["last_est"] >= _rms(f, ref_f)
    assert np.abs(pcm.astype(np.int64) - ref_pcm.astype(np.int64)).max() <= 1
    plain = p.build(gpu_api)
    plain[2].set_option("band_mode", 2)
    plain[2].set_option("sine_mode", 0)
    pcm0, f0 = p.render(gpu_api, built=plain)
    assert np.array_equal(pcm, pcm0) and np.array_equal(_bits(f), _bits(f0))
    assert plain[2].band_guard_stats()["last_est"] < st["last_est"]      # (the band-pass estimate alone)


def test_the_chain_launch_measures_the_probes_samples_itself(gpu_api, oracle):
    """Config 3's shape at 60 s: one probed vertex, a sample every 256 frames -- the guarded chain launch's tiles evaluate their own
    sixteen samples (BandScanDesc::nz_probe) and no k_sine_probe launch is left; test hook debug.inline_probe 0 brings the launch
    back: the same bytes, the same estimate (the same sample energies, summed in another order); in chunks of one block the stride is
    16 and the probe stays a launch of its own."""
    p = W.config3(seconds=12.0)
    res = {}
    for inl in (1, 0):
        gb = _front_end_modes(p, gpu_api)
        gb[2].set_option("debug.inline_probe", inl)
        gb[2].set_profiling(1)
        pcm, f = p.render(gpu_api, built=gb)
        fam = gb[2].kernel_times()
        gb[2].set_profiling(0)
        st = gb[2].band_guard_stats()
        assert st["audits"] == 1 and st["redos"] == 0, st
        assert ("k_sine_probe" in fam) == (inl == 0), fam
        res[inl] = (pcm, f, st["last_est"])
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(_bits(res[0][1]), _bits(res[1][1]))
    assert abs(res[0][2] - res[1][2]) <= 1e-4 * res[0][2], (res[0][2], res[1][2])
    gb = _front_end_modes(p, gpu_api)
    gb[2].set_option("max_chunk_frames", 1024)
    gb[2].set_profiling(1)
    pcm, f = p.render(gpu_api, built=gb)
    assert "k_sine_probe" in gb[2].kernel_times()
    assert np.abs(pcm.astype(np.int64) - res[1][0].astype(np.int64)).max() <= 1   # (another chunking of the scan class: not the same bits)


def test_the_round_5_outlier_through_the_front_end_with_nothing_set(gpu_api, oracle, tmp_path):
    """Seed 123475 -- 2.6e-6 RMS in every mode round 5 had short of sine_mode 1: now over the bound by the probe's own measurement,
    rendered again, and the oracle's bits; through td_state_* with nothing set, plain, scanned and continued."""
    p = F.random_project(123475, allow_sinf=True)
    ob = p.build(oracle)
    gb = _front_end_modes(p, gpu_api)
    lua = p.to_lua(str(tmp_path / "a"))
    s = gpu_api.State("", 48000, p.bl)
    assert s.refresh(lua), gpu_api.last_error()
    for scan in (False, True, False):
        op, of = p.render(oracle, built=ob, scan=scan)
        gp, gf = p.render(gpu_api, built=gb, scan=scan)
        ok = np.isfinite(of)
        assert np.array_equal(np.isfinite(gf), ok)
        scale = max(1.0, float(np.abs(of[ok]).max()))
        assert _rms(gf, of) / scale <= 1e-6, (scan, _rms(gf, of) / scale, gb[2].band_guard_stats())
        assert np.array_equal(gp, op)           # (done again = mode 1's form and the exact band-pass kernels: the oracle's PCM)
        if scan:
            s.scan_exact()
        assert np.array_equal(s.render_to_memory(), op)
    st = gb[2].band_guard_stats()
    assert st["redos"] >= 3 and st["max_est"] > 1e-6, st
    # ... and what the figure was about: the same graph in the fast forms, unguarded
    fast = p.build(gpu_api)
    fast[2].set_option("sine_mode", 0)
    ff = p.render(gpu_api, built=fast)[1]
    of = p.render(oracle)[1]
    assert _rms(ff, of) > 1e-6


PROJECTS = [("config3", lambda: W.config3(seconds=1.5)), ("synth", lambda: W.synth_project(seconds=1.2)),
            ("synth_bl333", lambda: W.synth_project(seconds=1.0, bl=333))]


@pytest.mark.parametrize("name,mk", PROJECTS)
def test_a_forced_verdict_gives_the_oracles_bits(gpu_api, oracle, name, mk):
    """Bound 0: every audited render is done again from the state it started in -- band-pass exact, sine kinds in mode 1's form:
    bit for bit the oracle's, plain, scanned, continued, in chunks and in block pulls."""
    p = mk()
    for chunk in (0, 4096):
        ob = p.build(oracle)
        gb = _front_end_modes(p, gpu_api, band_guard_ppb=0)
        if chunk:
            gb[2].set_option("max_chunk_frames", chunk)
        for scan in (False, True, False):
            op, of = p.render(oracle, built=ob, scan=scan)
            gp, gf = p.render(gpu_api, built=gb, scan=scan)
            assert np.array_equal(np.isnan(gf), np.isnan(of))
            assert np.array_equal(_bits(gf)[~np.isnan(of)], _bits(of)[~np.isnan(of)]) and np.array_equal(gp, op), (name, chunk, scan)
        assert gb[2].band_guard_stats()["redos"] == 3
    # block pulls (Graph::render, graph.rs:182-193): every pull probed, audited and done again
    osb, ofb, og = p.build(oracle)
    sb, fb, g = _front_end_modes(p, gpu_api, band_guard_ppb=0)
    for b in range(4):
        ol, orr = og.render(osb, ofb)
        gl, gr = g.render(sb, fb)
        ofb.set_time_to_next_block()
        fb.set_time_to_next_block()
        assert np.array_equal(_bits(gl), _bits(ol)) and np.array_equal(_bits(gr), _bits(orr)), (name, b)


def test_a_vertex_the_audit_cannot_follow_takes_glibcs_sine(gpu_api, oracle):
    """Two Normalize vertices in a row behind a Synth vertex: the static path analysis gives up (as it does for a band-pass
    vertex there), the vertex renders in mode 1's form from the start -- the oracle's bits, nothing probed for it."""
    p = W.ProjectScript(48000, 1024)
    p.set_length(1.0)
    p.event_files["n"] = np.array([(0.01, 60.0, 0.8), (0.5, 60.0, 0.0), (0.6, 67.0, 0.5)], dtype=np.float32)
    p.load_midi_floww("n", "n")
    p.add_synth("syn", 0.8, 10.0, "n", 0.4, 0.3, W.HIT_ADSR, 1.0, 0.8, W.NOTE_ADSR, 0.5, W.STD_ADSR)
    p.add_normalize("n1", 1.0, 0.0)
    p.add_normalize("n2", 0.9, 0.0)
    p.connect("syn", "n1")
    p.connect("n1", "n2")
    p.set_output("n2")
    op, of = p.render(oracle)
    gb = _front_end_modes(p, gpu_api)
    gp, gf = p.render(gpu_api, built=gb)
    assert np.array_equal(_bits(gf), _bits(of)) and np.array_equal(gp, op)
    assert gb[2].band_guard_stats()["audits"] == 0


def test_a_batch_settles_every_projects_verdict(gpu_api, oracle):
    """td_batch_*: projects probed in one merged k_sine_probe launch; the one forced over the bound is done again alone."""
    import ctypes as C
    P = 3
    projects = [W.config3(seconds=1.0) for i in range(P)]      # (estimate ~8e-8: under the bound unless forced)
    refs = [q.render(oracle)[0] for q in projects]
    batch = gpu_api.Batch()
    graphs = []
    for i, q in enumerate(projects):
        sb, fb, g = _front_end_modes(q, gpu_api)
        if i == 1:
            g.set_option("band_guard_ppb", 0)
        batch.add(sb, fb, g)
        graphs.append(g)
    batch.rewind()
    batch.render_all(projects[0].cs, 16)
    for i, g in enumerate(graphs):
        pcm = np.zeros((projects[i].cs * 1024, 2), np.int16)
        gpu_api._check(gpu_api.lib().td_graph_read_pcm(g.h, pcm.ctypes.data_as(C.c_void_p), pcm.nbytes))
        if i == 1:
            assert np.array_equal(pcm, refs[i])
        else:
            assert np.abs(pcm.astype(np.int64) - refs[i].astype(np.int64)).max() <= 1
    assert [g.band_guard_stats()["redos"] for g in graphs] == [0, 1, 0]
    assert all(g.band_guard_stats()["audits"] == 1 for g in graphs)
