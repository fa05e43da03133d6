"""(round 6) The sine guard's probe against the truth AT THE VERTEX: every debug_sine / synth vertex of a random project made the
output vertex in turn -- no gain, no Normalize vertex behind it -- rendered fast (sine_mode 2, whose estimate is then the probe's
own measurement) and exact (sine_mode 1).  python tools/sine_guard_vertex.py <seed> ..."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from termdaw_amd import api
from oracle import binding as oracle
import test_gpu_fuzz as F

for seed in [int(a) for a in sys.argv[1:]]:
    p = F.random_project(seed, allow_sinf=True)
    names = [c[0] for k in ("add_debug_sine", "add_synth") for c in p.calls.get(k, [])]
    print("seed", seed, "bl", p.bl, "cs", p.cs, "sine vertices:", names, flush=True)
    for k in ("add_synth", "add_debug_sine"):
        for c in p.calls.get(k, []):
            print("   ", k, c)
    keep = p.output_vertex
    for n in names:
        p.output_vertex = n
        ref = p.render(oracle)[1]
        gb = p.build(api)
        gb[2].set_option("sine_mode", 2)
        gb[2].set_option("band_guard_ppb", 1000000000)   # (never redone: the fast frames are what is compared)
        f = p.render(api, built=gb)[1]
        st = gb[2].band_guard_stats()
        d = f.astype(np.float64) - ref.astype(np.float64)
        ok = np.isfinite(d)
        worst = int(np.argmax(np.abs(np.where(ok, d, 0)).max(axis=1)))
        print("  %-10s true rms %.3g  max |d| %.3g at frame %d (ref %s fast %s)  probe est %.3g  nonfinite mismatch %d" % (
            n, float(np.sqrt(np.mean(np.where(ok, d, 0) ** 2))), float(np.abs(np.where(ok, d, 0)).max()), worst, ref[worst], f[worst], st["last_est"],
            int((np.isfinite(f) != np.isfinite(ref)).sum())), flush=True)
    p.output_vertex = keep
