"""Sweep of the long warm-up length (W = x / gamma) of the speculative band-pass on the deep chain: render time
and repair counts.  Exactness never depends on it; a shorter W trades walk length against repairs."""
import sys, time
sys.path.insert(0, '.')
from termdaw_amd import api, workloads as W
for name, p in (("config4", W.config4()), ("config3", W.config3()), ("drum60", W.drum_project(seconds=60.0)), ("synth60", W.synth_project(seconds=60.0))):
    for x in (40, 32, 28, 24, 20, 17):
        sb, fb, g = p.build(api)
        g.set_option("debug.band_short", x)
        g.render_all(sb, fb, p.cs, 16, want_f32=False, want_pcm=False)
        t0 = time.perf_counter()
        for _ in range(3):
            g.reset_normalize_vertices(); fb.set_time(0); g.set_time(0); g.render_all_async(sb, fb, p.cs, 16)
        g.sync()
        dt = (time.perf_counter() - t0) / 3
        print("%-8s Ws = %3d/gamma: %8.3f ms per render, last band vertex %s" % (name, x, dt * 1e3, g.band_stats()))
