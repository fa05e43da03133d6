"""Per-kernel HIP-event times of the non-headline configs (run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from termdaw_amd import api, workloads as W

def run(name, p, reps=None):
    sb, fb, g = p.build(api)
    for kv in filter(None, os.environ.get("TD_OPTS", "").split(",")):   # e.g. TD_OPTS=band_mode=1,debug.band_scan_nf=8
        k, v = kv.split("=")
        g.set_option(k, int(v))
    def render():
        g.reset_normalize_vertices(); fb.set_time(0); g.set_time(0)
        g.render_all_async(sb, fb, p.cs, 16)
    render(); g.sync()
    # steady device clocks first (the first tens of ms after an idle period run slower), then enough renders for >= 50 ms
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 0.2:
        render(); n += 1
        if n % 16 == 0: g.sync()
    g.sync()
    per = (time.perf_counter() - t0) / n
    if reps is None: reps = max(3, min(5000, int(0.05 / per)))
    t0 = time.perf_counter()
    for _ in range(reps):   # plain timing first: the per-launch HIP events of profiling mode widen the gaps
        render()
    g.sync()
    plain = (time.perf_counter() - t0) / reps
    g.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        render()
    g.sync()
    dt = (time.perf_counter() - t0) / reps
    kt = g.kernel_times()
    g.set_profiling(False)
    frames = p.cs * p.bl
    print("%-10s %8.3f ms/render (%.3f with launch events)  %9.1f Msamples/s   " % (name, plain * 1e3, dt * 1e3, frames / plain / 1e6) +
          "  ".join("%s %.3f ms x%d" % (k, ms / n, n // reps) for k, (ms, n) in sorted(kt.items(), key=lambda kv: -kv[1][0])))

if __name__ == "__main__":
    sel = sys.argv[1:] or ["c1", "c2", "c3", "drum", "synth"]
    if "c1" in sel: run("config1", W.config1())
    if "c2" in sel: run("config2", W.config2())
    if "c3" in sel: run("config3", W.config3())
    if "drum" in sel: run("drum60", W.drum_project(seconds=60.0))
    if "synth" in sel: run("synth60", W.synth_project(seconds=60.0))
    if "c4" in sel: run("config4", W.config4())
    if "c4s" in sel: run("config4_6s", W.config4(seconds=6.0))
