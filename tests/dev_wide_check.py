"""One-off: very wide fan-in (300 sources into one normalize, then through a band-pass) vs the oracle."""
import sys
sys.path.insert(0, '.')
import numpy as np
from termdaw_amd import api, workloads as W
from oracle import binding as oracle
p = W.config2(seconds=2.0, n_src=300)
p.calls["add_bandpass"] = []
p.add_bandpass("bp", 1.0, 0.0, 1.0, 300.0, 5000.0, True)
p.add_normalize("final", 1.0, 0.0)
p.connect("sum", "bp"); p.connect("bp", "final"); p.set_output("final")
op, of = p.render(oracle)
for fuse in (1, 0):
    b = p.build(api); b[2].set_option("fuse_sources", fuse)
    gp, gf = p.render(api, built=b)
    print("fuse", fuse, "bad", int((gf.view(np.uint32) != of.view(np.uint32)).any(axis=1).sum()), "pcm equal", bool(np.array_equal(gp, op)))
