"""k_sources: the launches of a level that read no edge buffer -- affine Synth (synth_gen, extensions.rs:460-529), the
wavetable voice (sampsyn_gen, extensions.rs:532-578), SampleLerp (extensions.rs:384-421) and the envelope buffers of the
Adsr vertices (adsr_gen, extensions.rs:593-651) -- go out as ONE grid (engine option one_grid_sources, default 1).  Every
workgroup runs the block function of its own family unchanged, so the render is the separate launches' bit for bit; the
oracle comparisons of the other test files run with the option at its default."""
import numpy as np
import pytest

from termdaw_amd import workloads as W
from test_gpu_parity import assert_close, _bits

pytestmark = pytest.mark.gpu


def _render(api, p, opts, cs=None):
    sb, fb, g = p.build(api)
    for k, v in opts.items():
        g.set_option(k, v)
    g.set_profiling(1)
    out = [g.render_all(sb, fb, cs or p.cs, 16) for _ in range(2)]   # (the second continues the carried state)
    fam = g.kernel_times()
    g.set_profiling(0)
    return out, fam


@pytest.mark.parametrize("band_mode", [0, 1])
@pytest.mark.parametrize("mk,parts,gone", [
    (lambda: W.config3(seconds=4.0), 2, {"k_synth", "k_adsr_env"}),
    (lambda: W.config4(seconds=3.0, depth=30), 3, {"k_sampsyn", "k_sample_lerp", "k_adsr_env"}),
])
def test_one_grid_is_the_separate_launches_bit_for_bit(gpu_api, mk, parts, gone, band_mode):
    (a0, a1), fam_a = _render(gpu_api, mk(), {"band_mode": band_mode, "debug.one_grid_sources": 1})
    (b0, b1), fam_b = _render(gpu_api, mk(), {"band_mode": band_mode, "debug.one_grid_sources": 0})
    for x, y in ((a0, b0), (a1, b1)):
        assert np.array_equal(x[0], y[0]) and np.array_equal(_bits(x[1]), _bits(y[1]))
    assert fam_a["k_sources"][1] == 2 and not (gone & set(fam_a)), fam_a        # one launch per render instead of `parts`
    assert "k_sources" not in fam_b and gone <= set(fam_b), fam_b
    assert sum(n for _, n in fam_b.values()) - sum(n for _, n in fam_a.values()) == 2 * (parts - 1)


def test_one_grid_against_the_oracle_chunked_and_in_a_batch(gpu_api, oracle):
    """Chunked renders (every chunk its own grid), and a batch whose members bring different source kinds: the merged level
    holds an affine Synth launch, a wavetable voice, a SampleLerp and two envelope buffers -> one grid of four parts."""
    p = W.config3(seconds=2.5)
    sb, fb, g = p.build(gpu_api)
    g.set_option("max_chunk_frames", 30000)
    assert_close(g.render_all(sb, fb, p.cs, 16), p.render(oracle))
    projects = [W.config3(seconds=1.0), W.config4(seconds=1.0, depth=12), W.synth_project(seconds=1.0), W.config3(seconds=1.0, variant=1)]
    cs = projects[0].cs
    batch = gpu_api.Batch()
    for q in projects:
        batch.add(*q.build(gpu_api))
    batch.set_profiling(1)
    batch.render_all(cs, 16)
    fam = batch.kernel_times()
    batch.set_profiling(0)
    assert fam["k_sources"][1] == 1 and not ({"k_synth", "k_sampsyn", "k_sample_lerp", "k_adsr_env"} & set(fam)), fam
    import ctypes as C
    for i, q in enumerate(projects):
        ref = q.render(oracle)
        _, _, gi = batch.projects[i]
        f = np.zeros((cs * gi.bl, 2), np.float32)
        gpu_api._check(gpu_api.lib().td_graph_read_f32(gi.h, f.ctypes.data_as(C.POINTER(C.c_float)), f.size))
        assert_close((batch.read_pcm(i, cs), f), ref)
