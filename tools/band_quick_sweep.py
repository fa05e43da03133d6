"""Quick warm-up sweep for the block-response guess of the speculative band-pass (run on the GPU box):
config 4 per-render time, k_band_spec time and the number of repaired segments for band_quick = 0 (no guess) .. 30."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from termdaw_amd import api, workloads as W

def run(p, quick, reps=3, depth=100):
    sb, fb, g = p.build(api)
    g.set_option("debug.band_quick", quick)
    g.set_option("debug.band_depth", depth)
    def render():
        g.reset_normalize_vertices(); fb.set_time(0); g.set_time(0)
        g.render_all_async(sb, fb, p.cs, 16)
    render(); g.sync()
    t0 = time.perf_counter()
    for _ in range(reps): render()
    g.sync()
    ms = (time.perf_counter() - t0) / reps * 1e3
    g.set_profiling(1); render(); g.sync(); kt = g.kernel_times(); g.set_profiling(0)
    st = g.band_stats()
    return ms, {k: round(v[0] / max(v[1], 1), 4) for k, v in kt.items() if k.startswith("k_band") or k == "k_sum"}, st

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "c4"
    p = {"c4": W.config4, "c3": W.config3, "drum": lambda: W.drum_project(seconds=60.0)}[which]()
    for depth in (100,):
        for q in (0, 12, 8):
            if q == 0 and depth != 100: continue
            ms, kt, st = run(p, q, depth=depth)
            print("band_depth %3d band_quick %2d: %8.3f ms/render  %s  all-stage stats %s" % (depth, q, ms, kt, st), flush=True)
