"""(round 6 experiment) exact band-pass with longer segments (TD_BAND_SEG): bit-exact against the oracle?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from termdaw_amd import api, workloads as W
from oracle import binding as oracle
for name, p in (("config4 6s depth 24", W.config4(seconds=6.0, depth=24)), ("config3 4s", W.config3(seconds=4.0)), ("synth 3s", W.synth_project(seconds=3.0)), ("drum 5s", W.drum_project(seconds=5.0))):
    op, of = p.render(oracle)
    gb = p.build(api)
    gb[2].set_option("sine_mode", 1)
    gp, gf = p.render(api, built=gb)
    same = np.array_equal(gp, op) and np.array_equal(np.ascontiguousarray(gf).view(np.uint32)[~np.isnan(of)], np.ascontiguousarray(of).view(np.uint32)[~np.isnan(of)])
    print("TD_BAND_SEG=%s %-22s bit-exact %s  band stats %s" % (os.environ.get("TD_BAND_SEG", "-"), name, same, gb[2].band_stats() if hasattr(gb[2], "band_stats") else ""), flush=True)
