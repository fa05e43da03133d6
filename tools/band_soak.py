"""Randomised soak of the speculative band-pass against the oracle over a seed range (the cases of
tests/test_gpu_parity.py::test_band_pass_random_soak):  python tools/band_soak.py 0 200"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from termdaw_amd import api
from oracle import binding as oracle
import test_gpu_parity as T

lo_seed, hi_seed = int(sys.argv[1]), int(sys.argv[2])
bad, events = [], 0
for seed in range(lo_seed, hi_seed):
    p, chunk = T._soak_case(seed)
    gb, ob = p.build(api), p.build(oracle)
    if chunk:
        gb[2].set_option("max_chunk_frames", chunk)
    for scan in (False, True):
        gp, gf = p.render(api, built=gb, scan=scan)
        op, of = p.render(oracle, built=ob, scan=scan)
        events += gb[2].band_stats()["mismatched"]
        if ((gf.view(np.uint32) != of.view(np.uint32)) & ~np.isnan(of)).any() or not np.array_equal(gp, op):
            bad.append((seed, scan))
print("seeds", lo_seed, hi_seed, "repair cascades in the last chunks:", events, "mismatching renders:", bad)
sys.exit(1 if bad else 0)
