"""Latency of the realtime path: one td_graph_render_block call (1024 frames, 21.3 ms of audio) for the bench projects."""
import sys, time
sys.path.insert(0, '.')
from termdaw_amd import api, workloads as W
for name, p in (("config1", W.config1(seconds=3.0)), ("config2", W.config2(seconds=3.0)), ("drum", W.drum_project(seconds=3.0)),
                ("synth", W.synth_project(seconds=3.0)), ("config4", W.config4(seconds=3.0))):
    sb, fb, g = p.build(api)
    ts = []
    for b in range(p.cs):
        t0 = time.perf_counter()
        g.render(sb, fb)
        ts.append(time.perf_counter() - t0)
        fb.set_time_to_next_block()
    ts = sorted(ts[5:])
    print("%-8s block pull: median %.3f ms, p99 %.3f ms (budget %.1f ms)" % (name, ts[len(ts) // 2] * 1e3, ts[int(len(ts) * 0.99)] * 1e3, p.bl / 48.0))
