import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from termdaw_amd import api, workloads as W
for depth in (3, 6, 12, 30, 90):
    for q in (0, 10):
        p = W.config4(depth=depth)
        sb, fb, g = p.build(api)
        g.set_option("band_quick", q)
        g.render_all(sb, fb, p.cs, 16, want_f32=False, want_pcm=False)
        print("depth", depth, "band stages", (depth + 1) // 3, "quick", q, g.band_stats(), flush=True)
