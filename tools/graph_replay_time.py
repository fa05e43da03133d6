import os, sys, time
sys.path.insert(0, "/root/repo")
from termdaw_amd import api, workloads as W
for name, mk in (("config1", W.config1), ("config2", W.config2), ("config3", W.config3), ("config4", W.config4)):
    for opt in (0, 1):
        p = mk(); sb, fb, g = p.build(api)
        g.set_option("graph_replay", opt); g.set_option("output_f32", 0)
        def render():
            g.reset_normalize_vertices(); fb.set_time(0); g.set_time(0); g.render_all_async(sb, fb, p.cs, 16)
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < 0.3:
            render(); n += 1
            if n % 16 == 0: g.sync()
        g.sync(); g.host_times(reset=True)
        reps = max(5, min(3000, int(0.1 / ((time.perf_counter() - t0) / n))))
        t0 = time.perf_counter()
        for _ in range(reps): render()
        g.sync()
        ms = (time.perf_counter() - t0) / reps * 1e3
        h = g.host_times()
        # latency of ONE render from an idle stream
        lat = []
        for _ in range(10):
            g.sync(); t1 = time.perf_counter(); render(); g.sync(); lat.append(time.perf_counter() - t1)
        print("%s graph_replay %d: %.4f ms/render pipelined, host launch %.4f ms/render, single-render latency %.4f ms" % (name, opt, ms, h["launch"] / max(h["chunks"], 1), sorted(lat)[5] * 1e3), flush=True)
