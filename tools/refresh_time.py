"""Project load time (State::refresh: script + 64 WAV files through the device-side load pipeline), first and warm."""
import os, sys, tempfile, time
sys.path.insert(0, '.')
from termdaw_amd import api, workloads as W
d = tempfile.mkdtemp()
p = W.config2()
lua = p.to_lua(os.path.join(d, "a"))
for i in range(4):
    s = api.State("", 48000, 1024)
    t0 = time.perf_counter(); assert s.refresh(lua), api.last_error(); dt = time.perf_counter() - t0
    print("refresh #%d: %.1f ms" % (i, dt * 1e3))
    del s
