"""Engine option "sine_mode" 1 (include/termdaw_amd.h td_graph_set_option): debug_sine_gen and synth_gen
(/root/reference/src/extensions.rs:423-457, 460-529; `f32::sin` at :450 and :501 = libm's sinf) evaluate their sine with
glibc's algorithm restated operation for operation (kernels.hip sin_glibc; tests/test_sinf_restate.py pins that sequence against
the host's sinf on every finite float) -- the two kinds of the tolerance class then carry the oracle's BITS, like every other
kind: f32 output and PCM, plain / scanned / continued renders, random graphs, the graph that amplifies the tolerance class
beyond its bar (seed 123475, DESIGN.md 5 "Sine class")."""
import numpy as np
import pytest

from termdaw_amd import workloads as W
import test_gpu_fuzz as F

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _exact(p, api):
    built = p.build(api)
    built[2].set_option("sine_mode", 1)
    return built


def _assert_bits(got, ref):
    (gp, gf), (rp, rf) = got, ref
    assert np.array_equal(np.isnan(gf), np.isnan(rf))
    bad = np.nonzero((_bits(gf) != _bits(rf)) & ~np.isnan(rf))[0]
    assert bad.size == 0, "first differing value %d: got %s want %s (%d differ)" % (bad[0], gf.reshape(-1)[bad[0]], rf.reshape(-1)[bad[0]], bad.size)
    assert np.array_equal(gp, rp)


@pytest.mark.parametrize("name,mk", [("synth", lambda: W.synth_project(seconds=2.0)), ("synth_bl333", lambda: W.synth_project(seconds=1.5, bl=333)),
                                     ("config3", lambda: W.config3(seconds=3.0)), ("config3_variant", lambda: W.config3(seconds=2.0, variant=1))])
def test_sine_kinds_bit_exact(gpu_api, oracle, name, mk):
    p = mk()
    ob = p.build(oracle)
    gb = _exact(p, gpu_api)
    for scan in (False, True, False):   # fresh, scanned, continued from carried state
        _assert_bits(p.render(gpu_api, built=gb, scan=scan), p.render(oracle, built=ob, scan=scan))


def test_config3_full_60s_bit_exact(gpu_api, oracle):
    """BASELINE config 3 at the size the bench times it (2 813 blocks; 32 voices x 3 oscillators -> adsr -> band-pass ->
    normalize) with every kind exact: PCM and f32 the oracle's, bit for bit."""
    p = W.config3()
    assert p.cs == 2813
    gb = _exact(p, gpu_api)
    gb[2].set_option("band_mode", 0)
    _assert_bits(p.render(gpu_api, built=gb), p.render(oracle))


def test_the_fast_sine_differs_and_the_bare_graph_defaults_to_the_exact_one(gpu_api, oracle):
    """sine_mode 0 (the front-end's default, and what workloads.ProjectScript builds with) is the tolerance class -- <= 1e-6 RMS,
    not the same bits: the option is what does it.  A bare td_graph (nothing set) evaluates the exact sine, as it runs the exact
    band-pass kernels."""
    p = W.synth_project(seconds=1.0)
    rp, rf = p.render(oracle)
    gf = p.render(gpu_api)[1]
    assert not np.array_equal(_bits(gf), _bits(rf))
    assert float(np.sqrt(np.mean((gf.astype(np.float64) - rf.astype(np.float64)) ** 2))) <= 1e-6
    sb, fb, g = gpu_api.SampleBank(48000), gpu_api.FlowwBank(48000, 256), gpu_api.Graph(256, 48000)
    ob, of_, og = oracle.SampleBank(48000), oracle.FlowwBank(48000, 256), oracle.Graph(256, 48000)
    ev = np.array([(0.0, 57.0, 0.9), (0.21, 64.0, 0.7), (0.4, 57.0, 0.0)], np.float32)
    for f_, g_ in ((fb, g), (of_, og)):
        f_.add_events("f", ev)
        g_.add_debug_sine("s", 0.9, 10.0, 0)
        g_.set_output("s")
    got, ref = g.render_all(sb, fb, 100), og.render_all(ob, of_, 100)
    _assert_bits(got, ref)


@pytest.mark.parametrize("seed", list(range(100, 124)) + [123475, 131214, 133930])
def test_random_graphs_with_sine_kinds_bit_exact(gpu_api, oracle, seed):
    """tests/test_gpu_fuzz.py's random graphs with debug_sine / synth sources allowed, band-pass vertices in the exact mode: every
    render the oracle's bits -- the soak's sine-class outliers (123475: 2.6e-6 in the default mode) included."""
    p = F.random_project(seed, allow_sinf=True)
    try:
        ob = p.build(oracle)
    except (RuntimeError, KeyError):
        return
    gb = _exact(p, gpu_api)
    gb[2].set_option("band_mode", 0)
    for scan in (False, True, False):
        _assert_bits(p.render(gpu_api, built=gb, scan=scan), p.render(oracle, built=ob, scan=scan))


def test_front_end_option(gpu_api, oracle, tmp_path):
    """td_state_set_option(s, "sine_mode", 1) reaches the vertices a refresh builds afterwards: the WAV bytes of a project
    with Synth vertices are the oracle's (band-pass vertices in the exact mode)."""
    p = W.synth_project(seconds=1.0)
    ref_pcm = p.render(oracle)[0]
    s = gpu_api.State("", 48000, p.bl)
    s.set_option("sine_mode", 1)
    s.set_option("band_mode", 0)
    assert s.refresh(p.to_lua(str(tmp_path / "a"))), gpu_api.last_error()
    assert np.array_equal(s.render_to_memory(), ref_pcm)
    assert s.refresh(p.to_lua(str(tmp_path / "a"))), gpu_api.last_error()      # (options survive State::refresh)
    assert np.array_equal(s.render_to_memory(), ref_pcm)


def test_the_device_sine_is_the_hosts_sinf_over_every_magnitude(gpu_api, tmp_path):
    """td_device_sinf(.., sine_mode 1) = kernels.hip sin_glibc on the device, against THIS host's sinf (a two-line C shim, built
    here) on EVERY float there is -- all 2^32 bit patterns, denormals, both zeros, infinities and NaNs included: the same bits,
    NaN where sinf gives NaN."""
    import ctypes as C
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("needs gcc for the host shim")
    src = tmp_path / "shim.c"
    src.write_text("#include <math.h>\n#include <stddef.h>\nvoid host_sinf(const float* in, float* out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = sinf(in[i]); }\n")
    so = str(tmp_path / "shim.so")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", str(src), "-o", so, "-lm"])
    shim = C.CDLL(so)
    fp = C.POINTER(C.c_float)
    shim.host_sinf.argtypes = [fp, fp, C.c_size_t]
    lib = gpu_api.lib()

    def check(bits):
        x = bits.view(np.float32)
        dev = np.empty_like(x)
        ref = np.empty_like(x)
        assert lib.td_device_sinf(x.ctypes.data_as(fp), dev.ctypes.data_as(fp), x.size, 1), gpu_api.last_error()
        shim.host_sinf(x.ctypes.data_as(fp), ref.ctypes.data_as(fp), x.size)
        nan = np.isnan(ref)
        assert np.array_equal(np.isnan(dev), nan)
        bad = np.nonzero((dev.view(np.uint32) != ref.view(np.uint32)) & ~nan)[0]
        assert bad.size == 0, "sin(%r): device %r host %r (%d of %d differ)" % (x[bad[0]], dev[bad[0]], ref[bad[0]], bad.size, x.size)

    step = 1 << 24
    for lo in range(0, 1 << 32, step):        # 256 chunks of 2^24: every bit pattern there is (22 s on the MI355X box)
        check(np.arange(lo, lo + step, dtype=np.uint64).astype(np.uint32))
