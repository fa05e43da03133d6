"""One-off soak (not collected by pytest): the bit-exact fuzz over many more seeds, longer timelines included."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from termdaw_amd import api
from oracle import binding as oracle
import test_gpu_fuzz as F

lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad_seeds = []
for seed in range(lo, hi):
    p = F.random_project(seed)
    if seed % 3 == 0:
        p.set_length(6.0)
    try:
        ob = p.build(oracle)
    except (RuntimeError, KeyError):
        continue
    gb = p.build(api)
    for scan in (False, True, False):
        gp, gf = p.render(api, built=gb, scan=scan)
        op, of = p.render(oracle, built=ob, scan=scan)
        ok = np.array_equal(np.isnan(gf), np.isnan(of)) and not ((gf.view(np.uint32) != of.view(np.uint32)) & ~np.isnan(of)).any() and np.array_equal(gp, op)
        if not ok:
            bad_seeds.append((seed, scan))
            break
print("seeds", lo, hi, "bad:", bad_seeds)
