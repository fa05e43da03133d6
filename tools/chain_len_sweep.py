"""k_band_chain time against the number of tiles (config 4, scan mode): flat = a latency chain, rising = contention / issue."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from termdaw_amd import api, workloads as W
for secs in [float(x) for x in (sys.argv[1:] or ["2.5", "5", "10", "21", "32", "43", "60", "90", "120"])]:
    p = W.config4(seconds=secs)
    sb, fb, g = p.build(api)
    g.set_option("band_mode", 1)
    for kv in filter(None, os.environ.get("TD_OPTS", "").split(",")):
        k, v = kv.split("="); g.set_option(k, int(v))
    def render():
        g.reset_normalize_vertices(); fb.set_time(0); g.set_time(0)
        g.render_all_async(sb, fb, p.cs, 16)
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 0.15:
        render(); n += 1
        if n % 16 == 0: g.sync()
    g.sync()
    g.set_profiling(1)
    for _ in range(4): render()
    g.sync()
    kt = g.kernel_times()
    g.set_profiling(0)
    tiles = (p.cs * 1024 + 4095) // 4096
    ms = kt["k_band_scan"][0] / kt["k_band_scan"][1]
    print("%6.1f s  %5d tiles (%.2f per CU)  k_band_chain %.4f ms = %.3f us per stage" % (secs, tiles, tiles / 256.0, ms, ms * 1e3 / 84), flush=True)
    del sb, fb, g
