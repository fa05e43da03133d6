"""One-off soak of the speculative band-pass (k_band_spec / k_band_fix / k_band_fill): random stutter projects --
random cut-offs, gap spacings, lengths, block sizes, chunk caps -- against the oracle, bit for bit."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from termdaw_amd import api
from oracle import binding as oracle
import test_gpu_parity as T

lo_seed, hi_seed = int(sys.argv[1]), int(sys.argv[2])
bad = []
tot_ev = 0
for seed in range(lo_seed, hi_seed):
    rng = np.random.default_rng(seed)
    seconds = float(rng.choice([3.0, 8.0, 20.0, 45.0]))
    spacing = float(rng.choice([0.004, 0.01, 0.03, 0.08, 0.3]))
    lo = float(rng.choice([0.0, 30.0, 200.0, 1000.0, 4000.0, 9000.0]))
    hi = float(rng.choice([0.0, 60.0, 500.0, 3000.0, 12000.0, 18000.0]))
    if lo == 0.0 and hi == 0.0:
        hi = 700.0
    p = T._stutter_project(seconds, spacing, lo, hi, seed)
    gb, ob = p.build(api), p.build(oracle)
    if rng.random() < 0.3:
        gb[2].set_option("max_chunk_frames", int(rng.choice([65536, 300000, 1 << 20])))
    ok = True
    for scan in (False, True):
        gp, gf = p.render(api, built=gb, scan=scan)
        op, of = p.render(oracle, built=ob, scan=scan)
        st = gb[2].band_stats()
        tot_ev += st["mismatched"]
        if ((gf.view(np.uint32) != of.view(np.uint32)) & ~np.isnan(of)).any() or not np.array_equal(gp, op):
            ok = False
    if not ok:
        bad.append((seed, seconds, spacing, lo, hi))
print("seeds", lo_seed, hi_seed, "repair cascades in total", tot_ev, "bad:", bad)
