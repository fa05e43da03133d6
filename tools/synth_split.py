"""k_synth cost by oscillator (config 3's 32-voice note pattern): which part of the per-voice work dominates."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from termdaw_amd import api, workloads as W

def proj(sq, tf, tr):
    p = W.config3()
    c = list(p.calls["add_synth"][0])
    # (name, gain, angle, floww, sq_vel, sq_z, sq_adsr, tf_vel, tf_z, tf_adsr, tr_vel, tr_adsr)
    c[4], c[7], c[10] = sq, tf, tr
    p.calls["add_synth"][0] = tuple(c)
    return p

for name, v in (("all", (0.4, 1.0, 0.5)), ("square", (0.4, 0.0, 0.0)), ("topflat", (0.0, 1.0, 0.0)), ("triangle", (0.0, 0.0, 0.5))):
    p = proj(*v)
    sb, fb, g = p.build(api)
    g.render_all(sb, fb, p.cs, 16, want_f32=False, want_pcm=False)
    g.set_profiling(1)
    for _ in range(3):
        g.reset_normalize_vertices(); fb.set_time(0); g.set_time(0); g.render_all_async(sb, fb, p.cs, 16)
    g.sync()
    kt = g.kernel_times()
    print(name, "k_synth %.3f ms" % (kt["k_synth"][0] / kt["k_synth"][1]))
