"""Host time per call against total time per render, config 4 in scan mode (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from termdaw_amd import api, workloads as W
for name, p in (("config4", W.config4()), ("config3", W.config3())):
    for mode in (1, 0):
        sb, fb, g = p.build(api)
        g.set_option("band_mode", mode)
        for _ in range(3):
            g.reset_normalize_vertices(); fb.set_time(0); g.set_time(0); g.render_all_async(sb, fb, p.cs, 16)
        g.sync()
        g.host_times(reset=True)
        N = 20
        t0 = time.perf_counter(); a = b = c = 0.0
        for _ in range(N):
            t = time.perf_counter(); g.reset_normalize_vertices(); fb.set_time(0); a += time.perf_counter() - t
            t = time.perf_counter(); g.set_time(0); b += time.perf_counter() - t
            t = time.perf_counter(); g.render_all_async(sb, fb, p.cs, 16); c += time.perf_counter() - t
        th = time.perf_counter() - t0
        g.sync()
        tt = time.perf_counter() - t0
        print(name, "band_mode", mode, "host/render %.3f ms (reset+fb %.3f, set_time %.3f, render call %.3f) total/render %.3f ms" % (th/N*1e3, a/N*1e3, b/N*1e3, c/N*1e3, tt/N*1e3), g.host_times())
