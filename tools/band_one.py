"""One band-pass vertex over 60 s of looping noise: k_band_spec time and repair counts for warm-up settings."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from termdaw_amd import api, workloads as W
p = W.ProjectScript(48000, 1024)
p.set_length(60.0)
p.assets["a"] = W.Asset(W.noise_int16(5, 77777))
p.load_sample("a", "a", "")
p.add_sampleloop("l", 0.5, 0.0, "a")
p.add_bandpass("bp", 1.0, 0.0, 1.0, 20.0, 18000.0, True)
p.add_normalize("out", 1.0, 0.0)
p.connect("l", "bp"); p.connect("bp", "out"); p.set_output("out")
for opts in ({"debug.band_quick": 0, "debug.band_short": 40}, {"debug.band_quick": 16}, {"debug.band_quick": 12}, {"debug.band_quick": 10}, {"debug.band_quick": 8},
             {"debug.band_quick": 1, "debug.band_medium": 1}):   # (the last one: hardly any warm-up walk -- what the launch costs besides it)
    sb, fb, g = p.build(api)
    for k, v in opts.items(): g.set_option(k, v)
    g.render_all(sb, fb, p.cs, 16, want_f32=False, want_pcm=False)
    g.set_profiling(1)
    for _ in range(3):
        g.reset_normalize_vertices(); fb.set_time(0); g.set_time(0); g.render_all_async(sb, fb, p.cs, 16)
    g.sync()
    kt = g.kernel_times(); g.set_profiling(0)
    print(opts, {k: round(v[0] / max(v[1], 1), 4) for k, v in kt.items()}, g.band_stats(), flush=True)
