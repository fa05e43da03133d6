import sys
sys.path.insert(0, '/root/repo')
from termdaw_amd import api, workloads as W
p = W.config4()
sb, fb, g = p.build(api)
g.render_all(sb, fb, p.cs, 16, want_f32=False, want_pcm=False)
g.reset_normalize_vertices(); fb.set_time(0); g.set_time(0)
g.render_all(sb, fb, p.cs, 16, want_f32=False, want_pcm=False)
