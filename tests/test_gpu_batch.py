"""GPU parity of the multi-project batch (BASELINE config 5 on one GPU): State::render's loop (state.rs:563-575)
for many independent projects at once -- same-kind launches of different projects share one grid -- must give,
per project, exactly what the oracle gives for that project alone: PCM digest and the per-project peak table."""
import hashlib

import numpy as np
import pytest

from termdaw_amd import workloads as W
from test_gpu_parity import assert_bit_exact, assert_close, _bits

pytestmark = pytest.mark.gpu


def _digest(pcm):
    return hashlib.sha256(np.ascontiguousarray(pcm, dtype="<i2").tobytes()).hexdigest()


@pytest.mark.parametrize("n_projects,seconds", [(8, 60.0), (64, 60.0), (5, 1.7)])
def test_config5_batch_on_one_gpu(gpu_api, oracle, n_projects, seconds):
    """Config 5's per-GPU share: n_projects x config 2 with seeds offset by 64 * project id, all resident, rendered
    by ONE batch submission; every project's PCM and the whole peak table equal the oracle's."""
    batch = gpu_api.Batch()
    want_digest, want_peak = [], []
    cs = None
    for pid in range(n_projects):
        p = W.config2(seconds=seconds, seed_offset=64 * pid)
        cs = p.cs
        batch.add(*p.build(gpu_api))
        osb, ofb, og = p.build(oracle)
        pcm, _ = og.render_all(osb, ofb, p.cs, 16, want_f32=False)
        want_digest.append(_digest(pcm))
        want_peak.append(np.float32(og.get_normalization_value("sum")))
        del p, osb, ofb, og, pcm
    assert len(batch) == n_projects
    for rep in range(2):   # the second pass re-renders from the rewound state with the tables already on the device
        batch.rewind()
        assert batch.render_all(cs, 16) == cs * 1024
        assert [_digest(batch.read_pcm(i, cs)) for i in range(n_projects)] == want_digest
        peaks = batch.peaks()
        assert np.array_equal(_bits(peaks), _bits(np.array(want_peak, np.float32)))
        assert len(set(want_peak)) == n_projects       # (different seeds -> different peaks: the table is not degenerate)
    # scanned workflow for the whole batch: scan_exact per project, then render
    batch.normalize_scan(cs)
    batch.render_all(cs, 16)
    p0 = W.config2(seconds=seconds, seed_offset=0)
    assert_bit_exact((batch.read_pcm(0, cs), _f32_of(gpu_api, batch, 0, cs)), p0.render(oracle, scan=True))


def _f32_of(api, batch, i, cs):
    import ctypes as C
    sb, fb, g = batch.projects[i]
    f = np.zeros((cs * g.bl, 2), np.float32)
    api._check(api.lib().td_graph_read_f32(g.h, f.ctypes.data_as(C.POINTER(C.c_float)), f.size))
    return f


def test_batch_of_different_projects(gpu_api, oracle):
    """Projects of different shape in one batch (different vertex kinds, levels, block lengths): launches merge only
    where level, family and launch parameters agree; every project still equals its own oracle render -- twice
    (carried state across renders, quirks Q4 / Q14)."""
    projects = [W.drum_project(seconds=1.0), W.config1(seconds=1.0), W.config2(seconds=1.0, n_src=9),
                W.drum_project(seconds=1.0), W.synth_project(seconds=1.0), W.config4(seconds=1.0, depth=12),
                W.drum_project(seconds=1.0, bl=1000)]
    exact = [True, True, True, True, False, True, True]
    cs = projects[0].cs
    batch = gpu_api.Batch()
    obuilt = []
    for p in projects:
        batch.add(*p.build(gpu_api))
        obuilt.append(p.build(oracle))
    for rep in range(2):
        batch.render_all(cs, 16)
        for i, p in enumerate(projects):
            osb, ofb, og = obuilt[i]
            ref = og.render_all(osb, ofb, cs, 16)
            got = (batch.read_pcm(i, cs), _f32_of(gpu_api, batch, i, cs))
            (assert_bit_exact if exact[i] else assert_close)(got, ref)
    # a member of a batch can still render on its own (shared stream, own table arena)
    sb, fb, g = batch.projects[0]
    osb, ofb, og = obuilt[0]
    assert_bit_exact(g.render_all(sb, fb, cs, 16), og.render_all(osb, ofb, cs, 16))
    # the handles outlive the batch
    del batch
    assert_bit_exact(g.render_all(sb, fb, cs, 16), og.render_all(osb, ofb, cs, 16))


def test_batch_peak_table_layout(gpu_api, oracle):
    """td_batch_peak_table_device: own entries at first + i * stride, zeros elsewhere (the all-reduce(max) input)."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")   # the runtime the library itself is linked against (already loaded)
    batch = gpu_api.Batch()
    want = []
    for pid in range(3):
        p = W.config2(seconds=0.3, n_src=5, seed_offset=64 * pid, base_len=3000)
        batch.add(*p.build(gpu_api))
        osb, ofb, og = p.build(oracle)
        og.render_all(osb, ofb, p.cs, 16, want_f32=False)
        want.append(og.get_normalization_value("sum"))
        cs = p.cs
    batch.render_all(cs, 16)
    d = C.c_void_p()
    assert hip.hipMalloc(C.byref(d), 32) == 0
    seven = np.full(8, 7.0, np.float32)
    assert hip.hipMemcpy(d, seven.ctypes.data_as(C.c_void_p), 32, 1) == 0   # hipMemcpyHostToDevice
    batch.peak_table_device(d.value, 8, first=1, stride=2)
    batch.sync()
    got = np.zeros(8, np.float32)
    assert hip.hipMemcpy(got.ctypes.data_as(C.c_void_p), d, 32, 2) == 0     # hipMemcpyDeviceToHost
    assert np.array_equal(got, np.array([0, want[0], 0, want[1], 0, want[2], 0, 0], np.float32))
    with pytest.raises(gpu_api.TermdawError):
        batch.peak_table_device(d.value, 8, first=4, stride=2)   # project 2 would land on entry 8
    hip.hipFree(d)


def test_batch_with_chunked_members(gpu_api, oracle):
    """Members whose timelines exceed their chunk cap render chunk after chunk inside the batch (different members finish
    their chunks at different submissions); scanned and un-scanned, twice."""
    mk = [lambda: W.drum_project(seconds=1.2), lambda: W.synth_project(seconds=1.2), lambda: W.config2(seconds=1.2, n_src=5),
          lambda: W.drum_project(seconds=1.2)]
    caps = [7000, 20000, 0, 3000]
    exact = [True, False, True, True]
    projects = [m() for m in mk]
    cs = projects[0].cs
    batch = gpu_api.Batch()
    obuilt = []
    for p, cap in zip(projects, caps):
        sb, fb, g = p.build(gpu_api)
        if cap:
            g.set_option("max_chunk_frames", cap)
        batch.add(sb, fb, g)
        obuilt.append(p.build(oracle))
    for scan in (False, True, False):
        if scan:
            batch.normalize_scan(cs)
            for osb, ofb, og in obuilt:
                og.true_normalize_scan(osb, ofb, cs)
        batch.render_all(cs, 16)
        for i in range(len(projects)):
            osb, ofb, og = obuilt[i]
            ref = og.render_all(osb, ofb, cs, 16)
            got = (batch.read_pcm(i, cs), _f32_of(gpu_api, batch, i, cs))
            (assert_bit_exact if exact[i] else assert_close)(got, ref)


@pytest.mark.parametrize("band_mode", [0, 1])
def test_config4_batch_of_8(gpu_api, oracle, band_mode):
    """Eight config-4 projects (256-vertex chain, asset seeds 7 + project id) in one batch at 10 s: every member against
    its own oracle render -- bit for bit with the exact band-pass kernels, within 1e-6 RMS / +-1 LSB in scan mode (where the
    eight chains are ONE k_band_chain launch)."""
    seconds, n = 10.0, 8
    batch = gpu_api.Batch()
    projects = [W.config4(seconds=seconds, variant=pid) for pid in range(n)]
    cs = projects[0].cs
    for p in projects:
        built = p.build(gpu_api)
        built[2].set_option("band_mode", band_mode)
        batch.add(*built)
    batch.rewind()
    assert batch.render_all(cs, 16) == cs * 1024
    digests = set()
    for i, p in enumerate(projects):
        ref = p.render(oracle)
        got = (batch.read_pcm(i, cs), _f32_of(gpu_api, batch, i, cs))
        (assert_close if band_mode else assert_bit_exact)(got, ref)
        digests.add(_digest(ref[0]))
    assert len(digests) == n   # (different assets -> different renders)


def test_config3_batch_of_8_scan(gpu_api, oracle):
    seconds, n = 6.0, 8
    batch = gpu_api.Batch()
    projects = [W.config3(seconds=seconds, variant=pid) for pid in range(n)]
    cs = projects[0].cs
    for p in projects:
        built = p.build(gpu_api)
        built[2].set_option("band_mode", 1)
        batch.add(*built)
    batch.rewind()
    batch.render_all(cs, 16)
    for i, p in enumerate(projects):
        assert_close((batch.read_pcm(i, cs), _f32_of(gpu_api, batch, i, cs)), p.render(oracle))


def test_a_failing_step_leaves_the_other_projects_where_it_found_them(gpu_api, oracle):
    """A batch step in which a LATER project fails to compile (a sampleloop whose sample index lies outside its bank)
    restores the host side of the projects compiled before it: the reset_normalization consumed from project 0, its loop
    cursors, its FlowwBank cursor, the drum project's carried voices.  Rendered on their own afterwards they give exactly
    what a fresh render gives -- the failed step has left no trace."""
    p0, p1 = W.config2(seconds=1.0, n_src=5), W.drum_project(seconds=1.0)
    cs = p0.cs
    b0, b1 = p0.build(gpu_api), p1.build(gpu_api)
    bad_sb, bad_fb, bad_g = gpu_api.SampleBank(48000), gpu_api.FlowwBank(48000, 1024), gpu_api.Graph(1024, 48000)
    bad_g.add_sampleloop("x", 1.0, 0.0, 7)     # no such sample: the render fails while this project is compiled
    bad_g.add_normalize("n", 1.0, 0.0)
    bad_g.connect("x", "n")
    bad_g.set_output("n")
    batch = gpu_api.Batch()
    batch.add(*b0)
    batch.add(*b1)
    # one good render first: carried state everywhere (running peak, voices)
    batch.render_all(cs, 16)
    ob0, ob1 = p0.build(oracle), p1.build(oracle)
    ob0[2].render_all(ob0[0], ob0[1], cs, 16)
    ob1[2].render_all(ob1[0], ob1[1], cs, 16)
    batch.add(bad_sb, bad_fb, bad_g)
    batch.rewind()
    for (osb, ofb, og) in (ob0, ob1):
        og.reset_normalize_vertices()
        ofb.set_time(0)
    with pytest.raises(gpu_api.TermdawError):
        batch.render_all(cs, 16)
    for (sb, fb, g), (osb, ofb, og) in ((b0, ob0), (b1, ob1)):
        assert_bit_exact(g.render_all(sb, fb, cs, 16), og.render_all(osb, ofb, cs, 16))
