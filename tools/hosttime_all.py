"""Host time per render_all_async call (event compile + descriptor build + launches) vs total time per
render, for every bench config (run on the GPU box)."""
import sys, time
sys.path.insert(0, '.')
from termdaw_amd import api, workloads as W
for name, p in (("config2", W.config2()), ("config3", W.config3()), ("drum60", W.drum_project(seconds=60.0)),
                ("synth60", W.synth_project(seconds=60.0)), ("config4", W.config4())):
    sb, fb, g = p.build(api)
    for _ in range(3):
        g.reset_normalize_vertices(); fb.set_time(0); g.set_time(0); g.render_all_async(sb, fb, p.cs, 16)
    g.sync()
    N = 10
    g.host_times()
    t0 = time.perf_counter(); c = 0.0
    for _ in range(N):
        g.reset_normalize_vertices(); fb.set_time(0); g.set_time(0)
        t = time.perf_counter(); g.render_all_async(sb, fb, p.cs, 16); c += time.perf_counter() - t
    g.sync()
    tt = time.perf_counter() - t0
    ht = g.host_times()
    print("%-8s host %.3f ms per render call, total %.3f ms per render; phases per render (ms): " % (name, c / N * 1e3, tt / N * 1e3) +
          ", ".join("%s %.3f" % (k, ht[k] / N) for k in ("compile", "descriptors", "upload", "launch")))
