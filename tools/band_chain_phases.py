"""Where a k_band_chain stage spends its time: the kernel's own clock stamps (engine option band_scan_debug 4 makes every
wave write its cycle counter at eight points of each stage over its frames of the chain's output).  GPU box only.
  0 stage start | 1 zero-state runs done | 2 wave scan done | 3 past barrier 1 | 4 totals published (wave 0)
  5 look-back read and summed (wave 0) | 6 past barrier 2 | 7 output + pan/gain done"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from termdaw_amd import api, workloads as W

STAGES = int(os.environ.get("STAGES", "84"))

def project(seconds=60.0):
    p = W.ProjectScript(48000, 1024)
    p.set_length(seconds)
    p.assets["a"] = W.Asset(W.noise_int16(5, 77777))
    p.load_sample("a", "a", "")
    p.add_sampleloop("l", 0.5, 0.0, "a")
    prev = "l"
    for i in range(STAGES):
        p.add_bandpass("bp%d" % i, 1.0, 0.0, 1.0, 100.0 + 10.0 * i, 8000.0 - 20.0 * i, True)
        p.connect(prev, "bp%d" % i)
        prev = "bp%d" % i
    p.set_output(prev)
    return p

if __name__ == "__main__":
    p = project(float(os.environ.get("SECONDS_", "60")))
    sb, fb, g = p.build(api)
    g.set_option("band_mode", int(os.environ.get("MODE", "1")))   # (2: the guarded instantiation)
    def render():
        g.reset_normalize_vertices(); fb.set_time(0); g.set_time(0)
        return g.render_all(sb, fb, p.cs, 16)
    for _ in range(3):
        render()
    g.set_option("debug.band_scan", 4)
    pcm, f = render()
    st = np.ascontiguousarray(f).view(np.uint64).reshape(-1)          # one u64 per frame
    WT = 1024
    n_w = len(st) // WT - 1
    t = st[: n_w * WT].reshape(n_w, WT)[:, : STAGES * 8].reshape(n_w, STAGES, 8).astype(np.int64)
    w0 = t[0::4]                                                       # wave 0 of every workgroup
    base = t[:, :1, :1].min()
    print("waves %d, stages %d; clock ticks (median over waves 0 of the tiles / all waves), stages 2..%d" % (n_w, STAGES, STAGES - 1))
    names = ["runs", "scan", "barrier 1", "totals + publish (w0)", "look-back (w0)", "barrier 2 (w0)", "entry state + output"]
    for k in range(7):
        src = w0 if k in (3, 4, 5) else t
        d = (src[:, 2:, k + 1] - src[:, 2:, k])
        print("  %-16s median %7.0f   p10 %7.0f   p90 %7.0f" % (names[k], np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
    other = t[1::4]
    print("  barrier 2 (wave 1: 3 -> 6)  median %7.0f" % np.median(other[:, 2:, 6] - other[:, 2:, 3]))
    raw = st[: n_w * WT].reshape(n_w, WT)
    hw = raw[0::4, WT - 1]; tiles = raw[0::4, WT - 2].astype(np.int64)
    hwid = (hw >> np.uint64(32)).astype(np.int64); xcc = (hw & np.uint64(0xF)).astype(np.int64)
    cu = (hwid >> 8) & 0xF; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
    where = xcc * 1000 + se * 100 + sh * 10 + cu     # (a label, not an index)
    by = {}
    for w_, t_ in zip(where, tiles): by.setdefault(int(w_), []).append(int(t_))
    print("  %d distinct CUs; tiles per CU: %s" % (len(by), np.bincount([len(v) for v in by.values()])))
    for k_ in sorted(by)[:12]: print("    xcc/se/sh/cu %05d: tiles %s" % (k_, sorted(by[k_])))
    stage = t[:, 3:, 0] - t[:, 2:-1, 0]
    print("  stage to stage   median %7.0f   (links = stage - sum of the above)" % np.median(stage))
    total = (t[:, -1, 7] - t[:, 0, 0])
    print("  whole chain per wave: median %d ticks; first start .. last end over all waves: %d ticks" % (np.median(total), t[:, -1, 7].max() - t[:, 0, 0].min()))
    print("  wave start skew (p90 - p10 of stage-0 start): %d ticks" % (np.percentile(t[:, 0, 0], 90) - np.percentile(t[:, 0, 0], 10)))
