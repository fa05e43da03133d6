import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from termdaw_amd import api, workloads as W
from oracle import binding as oracle
import test_gpu_parity as T
p = T._stutter_project(30.0, 0.02, 3000.0, 9000.0, seed=30)
op, of = p.render(oracle)
for par in (0, 1, 1, 1, 1, 1, 1):
    built = p.build(api)
    built[2].set_option("band_parallel", par)
    gp, gf = p.render(api, built=built)
    bad = np.nonzero((gf.view(np.uint32) != of.view(np.uint32)).any(axis=1))[0]
    print("parallel", par, "stats", built[2].band_stats(), "bad frames", bad.size, bad[:8], "first got", gf[:3].ravel(), "ref", of[:3].ravel())
