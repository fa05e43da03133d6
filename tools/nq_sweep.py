"""Tuning aid: config-2-like renders of several lengths with the packed loop sum forced to 4 / 8 / 16 frames per lane
(TD_FORCE_NQ = 1 | 2 | 4; read once per process, hence one subprocess per setting)."""
import os, subprocess, sys
code = r'''
import sys, time
sys.path.insert(0, ".")
from termdaw_amd import api, workloads as W
for secs in (3.0, 6.0, 12.0, 24.0, 45.0, 60.0, 120.0, 300.0):
    p = W.config2(seconds=secs)
    sb, fb, g = p.build(api)
    for _ in range(5):
        g.reset_normalize_vertices(); fb.set_time(0); g.render_all_async(sb, fb, p.cs, 16)
    g.sync()
    t0 = time.perf_counter(); N = 30
    for _ in range(N):
        g.reset_normalize_vertices(); fb.set_time(0); g.render_all_async(sb, fb, p.cs, 16)
    g.sync()
    dt = (time.perf_counter() - t0) / N
    print("%6.0f s %5d tiles: %.4f ms  %8.0f Msamples/s" % (secs, p.cs, dt * 1e3, p.cs * 1024 / dt / 1e6))
'''
for nq in ("1", "2", "4"):
    print("TD_FORCE_NQ =", nq, flush=True)
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, TD_FORCE_NQ=nq))
