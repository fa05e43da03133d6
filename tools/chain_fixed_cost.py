"""(round 6) k_band_chain: what a launch costs besides its stages.  sampleloop -> S x bandpass [-> normalize], 60 s, band_mode 1."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from termdaw_amd import api, workloads as W

def project(stages, norm, synth_in=False):
    p = W.ProjectScript(48000, 1024)
    p.set_length(60.0)
    p.assets["a"] = W.Asset(W.noise_int16(5, 77777))
    p.load_sample("a", "a", "")
    p.add_sampleloop("l", 0.5, 0.0, "a")
    p.add_sampleloop("l2", 0.25, 10.0, "a")
    p.add_sum("mix", 1.0, 0.0)
    p.connect("l", "mix"); p.connect("l2", "mix")
    prev = "mix"
    for i in range(stages):
        p.add_bandpass("bp%d" % i, 1.0, 0.0, 1.0, 200.0, 4000.0, True)
        p.connect(prev, "bp%d" % i)
        prev = "bp%d" % i
    if norm:
        p.add_normalize("n", 1.0, 0.0)
        p.connect(prev, "n")
        prev = "n"
    p.set_output(prev)
    return p

for norm in (0, 1):
    for stages in (1, 2, 4, 8):
        p = project(stages, norm)
        sb, fb, g = p.build(api)
        g.set_option("band_mode", 1); g.set_option("output_f32", 0)
        def render():
            g.reset_normalize_vertices(); fb.set_time(0); g.set_time(0)
            g.render_all_async(sb, fb, p.cs, 16)
        for _ in range(50): render()
        g.sync()
        g.set_profiling(True)
        for _ in range(50): render()
        g.sync()
        kt = g.kernel_times()
        g.set_profiling(False)
        print("stages %d normalize %d: " % (stages, norm) + "  ".join("%s %.4f ms x%d" % (k, ms / n, n // 50) for k, (ms, n) in sorted(kt.items(), key=lambda kv: -kv[1][0])), flush=True)
