import sys, time
sys.path.insert(0, '.')
from termdaw_amd import api, workloads as W
p = W.config2()
sb, fb, g = p.build(api)
for fuse in (1, 0):
    g.set_option("fuse_sources", fuse)
    for prof in (0, 1):
        g.set_profiling(prof)
        for _ in range(3):
            g.reset_normalize_vertices(); fb.set_time(0); g.render_all_async(sb, fb, p.cs, 16)
        g.sync()
        t0 = time.perf_counter(); a = b = c = 0.0
        N = 20
        for _ in range(N):
            t = time.perf_counter(); g.reset_normalize_vertices(); a += time.perf_counter() - t
            t = time.perf_counter(); fb.set_time(0); b += time.perf_counter() - t
            t = time.perf_counter(); g.render_all_async(sb, fb, p.cs, 16); c += time.perf_counter() - t
        th = time.perf_counter() - t0
        g.sync()
        tt = time.perf_counter() - t0
        print("fuse", fuse, "prof", prof, "host/step %.3f ms (reset %.3f, fbset %.3f, render %.3f) total/step %.3f ms" % (th/N*1e3, a/N*1e3, b/N*1e3, c/N*1e3, tt/N*1e3))
