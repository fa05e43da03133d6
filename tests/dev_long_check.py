"""One-off: a 400 s drum project (two chunks at the 2^24-frame cap, 65536-segment band-pass launches) and a
6-minute gappy band-pass project vs the oracle."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from termdaw_amd import api, workloads as W
from oracle import binding as oracle
import test_gpu_parity as T
for name, p in (("drum400", W.drum_project(seconds=400.0)), ("gappy360", T._gappy_project(300.0, 3000.0, True, seconds=360.0))):
    t0 = time.time(); op, of = p.render(oracle); t1 = time.time()
    built = p.build(api)
    gp, gf = p.render(api, built=built); t2 = time.time()
    bad = ((gf.view(np.uint32) != of.view(np.uint32)) & ~np.isnan(of)).any(axis=1).sum()
    print(name, "frames", gf.shape[0], "oracle %.1f s, gpu %.2f s" % (t1 - t0, t2 - t1), "bad frames", int(bad), "pcm equal", bool(np.array_equal(gp, op)), built[2].band_stats())
