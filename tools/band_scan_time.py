"""k_band_scan on simple 60 s projects: plain edge input / adsr-through input, slow and fast smoothers (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from termdaw_amd import api, workloads as W


def project(kind, lo, hi, seconds=60.0):
    p = W.ProjectScript(48000, 1024)
    p.set_length(seconds)
    p.assets["a"] = W.Asset(W.noise_int16(5, 77777))
    p.load_sample("a", "a", "")
    p.add_sampleloop("l", 0.5, 0.0, "a")
    p.event_files["g"] = np.array([(0.25 * i + 0.05, 60.0, 0.9) for i in range(int(seconds * 4))], np.float32)
    p.load_midi_floww("g", "g")
    src = "l"
    if kind == "adsr":
        p.add_adsr("gate", 1.0, 0.0, 0.5, "g", False, True, -1, [0.01, 0.1, 0.8, 0.1, 0.2, 0.01])
        p.connect("l", "gate")
        src = "gate"
    p.add_bandpass("bp", 1.0, 0.0, 1.0, lo, hi, True)
    p.add_normalize("out", 1.0, 0.0)
    p.connect(src, "bp")
    p.connect("bp", "out")
    p.set_output("out")
    return p


def run(name, p, opts):
    sb, fb, g = p.build(api)
    for k, v in opts.items():
        g.set_option(k, v)
    def render():
        g.reset_normalize_vertices(); fb.set_time(0); g.set_time(0)
        g.render_all_async(sb, fb, p.cs, 16)
    for _ in range(20):
        render()
    g.sync()
    g.set_profiling(True)
    for _ in range(50):
        render()
    g.sync()
    kt = g.kernel_times()
    g.set_profiling(False)
    print("%-34s " % name + "  ".join("%s %.4f ms x%d" % (k, ms / n, n // 50) for k, (ms, n) in sorted(kt.items(), key=lambda kv: -kv[1][0])))


if __name__ == "__main__":
    for kind in ("edge", "adsr"):
        for lo, hi in ((20.0, 18000.0), (200.0, 4000.0)):
            for nf in (16, 8):
                for dbg in (0, 2):
                    run("%s %g-%g nf%d dbg%d" % (kind, lo, hi, nf, dbg), project(kind, lo, hi),
                        {"band_mode": 1, "fuse_sources": 0, "debug.band_scan_nf": nf, "debug.band_scan": dbg})
