"""End-to-end State.render timing (render + D2H + WAV write) for config 2, run on the GPU box."""
import os, sys, tempfile, time
sys.path.insert(0, '.')
from termdaw_amd import api, workloads as W
d = tempfile.mkdtemp()
p = W.config2()
lua = p.to_lua(os.path.join(d, "a"))
s = api.State("", 48000, 1024)
t0 = time.perf_counter(); assert s.refresh(lua), api.last_error(); t_load = time.perf_counter() - t0
out = os.path.join(d, "o.wav")
s.render(out)
for label, fn in (("render+D2H+WAV write", lambda: s.render(out)), ("render+D2H (memory)", lambda: s.render_to_memory()),
                  ("render+D2H (pinned view)", lambda: s.render_view())):
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    ts.sort()
    print("%-24s median %.3f ms  -> %.0f Msamples/s" % (label, ts[2] * 1e3, p.cs * 1024 / ts[2] / 1e6))
print("refresh (script + 64 WAV loads + device pipeline): %.1f ms" % (t_load * 1e3))
