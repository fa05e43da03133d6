import sys
sys.path.insert(0, '.')
import numpy as np
from termdaw_amd import api, workloads as W
from oracle import binding as oracle
for name, p in (("drum4", W.drum_project()), ("config3_4s", W.config3(seconds=4.0))):
    sb, fb, g = p.build(api)
    gp, gf = p.render(api, built=(sb, fb, g))
    print(name, "stats", g.band_stats())
    op, of = p.render(oracle)
    bad = np.nonzero((gf.view(np.uint32) != of.view(np.uint32)).any(axis=1))[0]
    print(name, "bad frames", bad.size, bad[:10], "pcm diff", int((gp != op).sum()))
    if bad.size:
        i = bad[0]
        print("  first bad", i, "seg", i // 256, gf[i], of[i], "max abs diff", np.abs(gf - of).max())
        runs = np.split(bad, np.nonzero(np.diff(bad) > 1)[0] + 1)
        print("  runs:", [(int(r[0]), int(r[-1])) for r in runs[:8]])
