"""k_synth's affine envelope form (synth_gen, extensions.rs:460-529; adsr.rs:46-92): the host cuts a Synth vertex'
intervals at every frame where a voice changes envelope piece and hands the kernel (s1, s2, A, B) per oscillator; a wave
whose frames lie in several intervals runs the voice loop once per interval.  Tolerance class (device sine): <= 1e-6 RMS on
the f32 output, +-1 LSB on the PCM, against the oracle."""
import numpy as np
import pytest

from termdaw_amd import workloads as W
from test_gpu_parity import assert_close

pytestmark = pytest.mark.gpu

HIT = [0.001, 0.02, 0.0, 0.0, 0.0, 0.0]
NOTE = [0.01, 0.1, 0.8, 5.0, 0.2, 0.5]
NINE = [0.1, 0.004, 0.9, 0.03, 0.5, 0.05, 0.3, 0.2, 0.05]


def _staggered(seconds, voices, step, hold):
    ev = []
    for j in range(voices):
        t = 0.003 + step * j
        while t < seconds:
            ev.append((t, 40.0 + 3 * j, 0.3 + 0.05 * j))
            ev.append((t + hold * (1.0 + 0.13 * j), 40.0 + 3 * j, 0.0))
            t += 2.7 * hold
    ev.sort(key=lambda e: e[0])
    return np.array(ev, np.float32)


@pytest.mark.parametrize("bl", [1024, 1000, 333, 2048])
@pytest.mark.parametrize("confs", [(HIT, NOTE, NOTE), (NINE, NINE, HIT), (NOTE, HIT, NINE)])
def test_staggered_voices_cut_everywhere(gpu_api, oracle, bl, confs):
    """Every voice has its own note-on time, so the attack / decay / sustain / release breakpoints of the three confs fall on
    different frames for every voice: hundreds of cuts, most waves near them straddle two or three intervals."""
    p = W.ProjectScript(48000, bl)
    p.set_length(2.0)
    p.event_files["n"] = _staggered(2.0, 7, 0.0137, 0.11)
    p.load_midi_floww("n", "n")
    p.add_synth("syn", 0.9, 10.0, "n", 0.4, 0.3, confs[0], 1.0, 0.8, confs[1], 0.5, confs[2])
    p.add_sum("out", 1.0, 0.0)
    p.connect("syn", "out")
    p.set_output("out")
    got, ref = p.render(gpu_api), p.render(oracle)
    assert_close(got, ref)
    assert np.abs(got[1].astype(np.float64) - ref[1].astype(np.float64)).max() < 3e-6
    # twice in a row: carried voices (env_t, rel_t) enter the next chunk's first interval
    gb, ob = p.build(gpu_api), p.build(oracle)
    for _ in range(2):
        assert_close(p.render(gpu_api, built=gb), p.render(oracle, built=ob))


@pytest.mark.parametrize("chunk", [0, 5000])
def test_affine_and_generic_vertices_in_one_graph(gpu_api, oracle, chunk):
    """Two Synth vertices at the same level: one whose confs take the affine form, one with a zero-length attack (quirk Q6:
    0 / 0 at the note-on frame) that keeps the generic per-frame form -- two launches (k_synth, k_synth_affine), NaN frames
    exactly where the oracle has them; oscillators switched off in different combinations."""
    p = W.ProjectScript(48000, 1024)
    p.set_length(1.5)
    p.event_files["n"] = _staggered(1.5, 4, 0.021, 0.2)
    p.load_midi_floww("n", "n")
    zero_attack = [0.0, 0.05, 0.6, 0.1, 0.3, 0.1]
    p.add_synth("a", 1.0, 0.0, "n", 0.4, 0.3, HIT, 0.0, 0.8, NOTE, 0.5, NOTE)        # top-flat off
    p.add_synth("b", 0.8, -20.0, "n", 0.0, 0.3, NOTE, 1.0, 0.7, zero_attack, 0.0, NOTE)   # generic: attack_sec 0
    p.add_synth("c", 0.7, 20.0, "n", 0.0, 0.3, NOTE, 0.0, 0.8, NOTE, 0.9, NINE)      # triangle only
    p.add_sum("ab", 1.0, 0.0)
    p.add_sum("out", 1.0, 0.0)
    p.connect("a", "ab")
    p.connect("c", "ab")
    p.connect("ab", "out")
    p.set_output("out")
    built = p.build(gpu_api)
    if chunk:
        built[2].set_option("max_chunk_frames", chunk)
    assert_close(p.render(gpu_api, built=built), p.render(oracle))
    # ... and with the NaN-producing vertex mixed in: same NaN mask as the oracle, finite frames within tolerance
    p.connect("b", "out")
    built = p.build(gpu_api)
    built[2].set_profiling(1)
    gp, gf = p.render(gpu_api, built=built)
    fam = built[2].kernel_times()
    built[2].set_profiling(0)
    assert fam["k_synth"][1] == 2, fam      # two launches of the family: generic + affine
    op, of = p.render(oracle)
    assert np.array_equal(np.isnan(gf), np.isnan(of)) and np.isnan(of).any()
    ok = np.isfinite(of)
    assert np.sqrt(np.mean((gf[ok].astype(np.float64) - of[ok].astype(np.float64)) ** 2)) <= 1e-6
    assert np.abs(gp.astype(np.int64) - op.astype(np.int64)).max() <= 1


@pytest.mark.parametrize("start_s", [0, 23, 400])
def test_sine_arguments_on_both_sides_of_the_half_turn_bound(gpu_api, oracle, start_s):
    """The affine form's sine rounds to half turns by adding 1.5 * 2^23 while the host can bound the chunk's arguments
    below 2e6 rad (SynthDesc::small_args), and by v_rndne beyond: a 12.5 kHz voice passes that bound 25.4 s
    into the timeline.  Rendered from 0 s (every chunk below), from 23 s in one-second chunks (the first two below,
    the rest above) and from 400 s (all above; arguments of 3e7 rad, an f32 ulp of 2 rad -- the rounding of
    `time * hz * 2 pi` IS the signal there, extensions.rs:501)."""
    p = W.ProjectScript(48000, 1024)
    p.set_length(6.0)
    ev = []
    for k in range(6):
        t = start_s + 0.05 + k
        ev += [(t, 127.0, 0.5), (t + 0.7, 127.0, 0.0), (t + 0.1, 52.0 + k, 0.4), (t + 0.8, 52.0 + k, 0.0)]
    ev.sort(key=lambda e: e[0])
    p.event_files["n"] = np.array(ev, np.float32)
    p.load_midi_floww("n", "n")
    p.add_synth("syn", 0.9, 0.0, "n", 0.4, 0.3, HIT, 1.0, 0.8, NOTE, 0.5, NOTE)
    p.add_sum("out", 1.0, 0.0)
    p.connect("syn", "out")
    p.set_output("out")
    res = []
    for backend in (gpu_api, oracle):
        sb, fb, g = p.build(backend)
        if backend is gpu_api:
            g.set_option("max_chunk_frames", 48000)
        fb.set_time(start_s * 48000)
        g.set_time(start_s * 48000)
        res.append(g.render_all(sb, fb, p.cs, p.bd))
    assert np.abs(res[1][1]).max() > 0.05
    assert_close(res[0], res[1])
