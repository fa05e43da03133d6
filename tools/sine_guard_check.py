"""(round 6) The sine kinds' guard (engine option sine_mode 2): what k_sine_probe measures against the true deviation.
python tools/sine_guard_check.py            on the GPU box"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from termdaw_amd import api, workloads as W
from oracle import binding as oracle
import test_gpu_fuzz as F


def rms(a, b):
    ok = np.isfinite(b)
    return float(np.sqrt(np.mean((a[ok].astype(np.float64) - b[ok].astype(np.float64)) ** 2))) if ok.any() else 0.0


def one(name, p, band_mode=2, sine_mode=2, **opts):
    ref_pcm, ref_f = p.render(oracle)
    scale = max(1.0, float(np.abs(ref_f[np.isfinite(ref_f)]).max()))
    gb = p.build(api)
    gb[2].set_option("band_mode", band_mode)
    gb[2].set_option("sine_mode", sine_mode)
    for k, v in opts.items():
        gb[2].set_option(k, v)
    pcm, f = p.render(api, built=gb)
    st = gb[2].band_guard_stats()
    print("%-28s band %d sine %d: rms %.3g (scale %.3g) est %.3g audits %d redos %d  pcm maxdiff %d" % (
        name, band_mode, sine_mode, rms(f, ref_f) / scale, scale, st["last_est"], st["audits"], st["redos"],
        int(np.abs(pcm.astype(np.int64) - ref_pcm.astype(np.int64)).max())), flush=True)
    return st


for sm in (0, 2):
    one("config3 4s", W.config3(seconds=4.0), 2, sm)
one("config3 4s", W.config3(seconds=4.0), 0, 2)
one("synth_project 3s", W.synth_project(seconds=3.0), 2, 0)
one("synth_project 3s", W.synth_project(seconds=3.0), 2, 2)
one("synth_project 3s", W.synth_project(seconds=3.0), 0, 2)
one("synth_project 0.1s", W.synth_project(seconds=0.1), 2, 0)
one("synth_project 0.1s", W.synth_project(seconds=0.1), 2, 2)
for seed in (123475, 16622, 7, 11, 19, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71):
    p = F.random_project(seed, allow_sinf=True)
    for sm in (0, 2):
        try:
            one("seed %d" % seed, p, 2, sm)
        except Exception as e:
            print("seed", seed, "failed:", e)
