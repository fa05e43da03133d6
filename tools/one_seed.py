"""One random project (tests/test_gpu_fuzz.py random_project) in every band mode against the oracle -- which part of a distance
is the filter arithmetic's and which the sine class':  python tools/one_seed.py <seed> [...]   (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from termdaw_amd import api
from oracle import binding as oracle
import test_gpu_fuzz as F
for seed in [int(a) for a in sys.argv[1:]]:
    p = F.random_project(seed, allow_sinf=True)
    ob = p.build(oracle)
    refs = [p.render(oracle, built=ob, scan=sc)[1] for sc in (False, True, False)]
    for mode in (0, 1, 2):
        gb = p.build(api)
        gb[2].set_option("band_mode", mode)
        out = []
        for k, sc in enumerate((False, True, False)):
            gf = p.render(api, built=gb, scan=sc)[1]
            of = refs[k]
            ok = np.isfinite(of)
            scale = max(1.0, float(np.abs(of[ok]).max())) if ok.any() else 1.0
            out.append(float(np.sqrt(np.mean(((gf[ok].astype(np.float64) - of[ok]) / scale) ** 2))) if ok.any() else 0.0)
        print("seed", seed, "band_mode", mode, "rms", ["%.3g" % r for r in out], "guard", gb[2].band_guard_stats(), flush=True)
os._exit(0)
