"""Band-pass speculation statistics of the bench configs (run on the GPU box): per band vertex
(segments, mismatched segments repaired by k_band_fix, ...) as td_graph_band_stats reports them."""
import sys
sys.path.insert(0, '.')
from termdaw_amd import api, workloads as W

for name, p in (("config3", W.config3()), ("drum60", W.drum_project(seconds=60.0)), ("synth60", W.synth_project(seconds=60.0)),
                ("config4", W.config4())):
    sb, fb, g = p.build(api)
    g.render_all(sb, fb, p.cs, 16, want_f32=False, want_pcm=False)
    print(name, g.band_stats())
