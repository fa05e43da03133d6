"""RMS deviation of the scan-mode band-pass (band_mode 1) from the oracle along BASELINE config 4's chain (GPU box).
Prints, per chain depth, the RMS of (HIP f32 output - oracle f32 output), the exact mode's (0 by construction) and the
largest PCM difference."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from termdaw_amd import api, workloads as W
from oracle import binding as oracle

def rms(a, b):
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))

if __name__ == "__main__":
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    for name, mk in [("config3", lambda: W.config3(seconds=seconds))] + [("config4 depth %d" % d, (lambda d=d: W.config4(seconds=seconds, depth=d))) for d in (12, 48, 126, 252)]:
        p = mk()
        ref = p.render(oracle)
        out = []
        for opts in ({"band_mode": 1}, {"band_mode": 1, "debug.band_chain": 0}, {"band_mode": 1, "debug.band_scan_nf": 8}):
            b = p.build(api)
            for k, v in opts.items():
                b[2].set_option(k, v)
            got = p.render(api, built=b)
            out.append("%s rms %.3g pcm %d" % (",".join("%s=%d" % kv for kv in opts.items()), rms(got[1], ref[1]),
                                               int(np.abs(got[0].astype(np.int64) - ref[0].astype(np.int64)).max())))
        sig = float(np.sqrt(np.mean(ref[1].astype(np.float64) ** 2)))
        print("%-18s signal rms %.3g | " % (name, sig) + " | ".join(out), flush=True)
