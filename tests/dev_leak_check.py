"""One-off: build / render / drop projects repeatedly and watch free device memory (hipMemGetInfo via torch)."""
import sys, gc
sys.path.insert(0, '.')
import torch
from termdaw_amd import api, workloads as W
def free_mb():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 2**20
base = None
for it in range(12):
    for p in (W.config2(seconds=5.0), W.drum_project(seconds=5.0), W.config3(seconds=3.0), W.config4(seconds=1.0)):
        sb, fb, g = p.build(api)
        g.render_all(sb, fb, p.cs, 16)
        g.true_normalize_scan(sb, fb, p.cs)
        g.render_all(sb, fb, p.cs, 24)
        del sb, fb, g
    gc.collect()
    f = free_mb()
    if it == 1: base = f
    print("iteration %2d: free %.1f MiB" % (it, f))
print("drift since iteration 1: %.1f MiB" % (base - free_mb()))
