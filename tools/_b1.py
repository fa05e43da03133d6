import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from termdaw_amd import api, workloads as W
p = W.ProjectScript(48000, 1024)
p.set_length(60.0)
p.assets["a"] = W.Asset(W.noise_int16(5, 77777))
p.load_sample("a", "a", "")
p.add_sampleloop("l", 0.5, 0.0, "a")
p.add_bandpass("bp", 1.0, 0.0, 1.0, 20.0, 18000.0, True)
p.add_normalize("out", 1.0, 0.0)
p.connect("l", "bp"); p.connect("bp", "out"); p.set_output("out")
sb, fb, g = p.build(api)
for i in range(3):
    g.reset_normalize_vertices(); fb.set_time(0); g.set_time(0); g.render_all_async(sb, fb, p.cs, 16); g.sync()
    print("--- render", i, flush=True)
