"""BASELINE config 5 on one GPU: P projects (config 2 with seed offset 64*p), each on its own graph handle
and HIP stream, rendered round-robin without host synchronisation in between.  Reports aggregate Msamples/s
against the same projects rendered strictly one after another."""
import sys, time
sys.path.insert(0, '.')
from termdaw_amd import api, workloads as W

P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
R = int(sys.argv[2]) if len(sys.argv) > 2 else 20
projs = [W.config2(seed_offset=64 * p) for p in range(P)]
built = [p.build(api) for p in projs]
frames = projs[0].cs * projs[0].bl

def once(sync_each):
    for (sb, fb, g) in built:
        g.reset_normalize_vertices(); fb.set_time(0)
        g.render_all_async(sb, fb, projs[0].cs, 16)
        if sync_each: g.sync()
    if not sync_each:
        for (_, _, g) in built: g.sync()

for mode in (True, False):
    once(mode); once(mode)
    t0 = time.perf_counter()
    for _ in range(R): once(mode)
    dt = (time.perf_counter() - t0) / R
    print("%d projects, %s: %.3f ms per batch, %.1f Msamples/s aggregate" % (P, "one after another (sync each)" if mode else "all streams in flight", dt * 1e3, P * frames / dt / 1e6))
