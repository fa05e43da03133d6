"""Developer aid (not a test): numpy model of k_band_spec's speculation on config 3's band-pass input, to see
which segments enter with a wrong state and what their warm-up windows look like.  Uses the oracle for the
band-pass input, hence lives under tests/."""
import sys, math
sys.path.insert(0, '.')
import numpy as np
from termdaw_amd import workloads as W
from oracle import binding as oracle

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 12.0
p = W.config3(seconds=secs)
# same project, output taken at the adsr vertex
p.calls["add_bandpass"] = []
p.calls["add_normalize"] = []
p.calls["connect"] = [c for c in p.calls["connect"] if c[1] not in ("band", "sum")]
p.set_output("env")
_, x = p.render(oracle)
x = np.ascontiguousarray(x, dtype=np.float32)
M = x.shape[0]
f32 = np.float32
def gamma(hz): return f32(1.0) - f32(math.e) ** (f32(-2.0) * f32(math.pi) * f32(hz) / f32(48000.0))
gl = f32(1.0 - np.float32(np.exp(np.float32(-2.0 * np.pi * 200.0 / 48000.0))))
S = 256
Ws = (int(40.0 / gl + 64) + 31) & ~31; Ws = (Ws + 255) & ~255
Wl = (int(150.0 / gl + 64) + 31) & ~31
print("M", M, "gamma", gl, "Ws", Ws, "W", Wl)
# exact trajectory (serial, low-L chain only)
xl = x[:, 0]
y = np.empty(M + 1, f32); y[0] = xl[0]
yy = f32(xl[0])
for n in range(M):
    yy = f32(yy + f32(gl * f32(xl[n] - yy))); y[n + 1] = yy
nseg = (M + S - 1) // S
blk = np.abs(x).reshape(-1, 256, 2).max(axis=(1, 2)) if M % 256 == 0 else None
const = (x.reshape(-1, 256, 2) == x.reshape(-1, 256, 2)[:, :1, :]).all(axis=(1, 2))
bp = np.where(const, -1.0, blk).astype(f32)
bad = []
for s in range(1, nseg):
    start = s * S
    w = Wl
    if start > Ws:
        b = bp[(start - Ws) // 256: start // 256]
        lo, hi = b.min(), b.max()
        if lo >= 1e-30 and lo >= hi * 1e-6: w = Ws
    b0 = max(start - w, 0)
    yy = f32(xl[b0]) if b0 > 0 else f32(xl[0])
    for n in range(b0, start):
        yy = f32(yy + f32(gl * f32(xl[n] - yy)))
    if yy.view(np.uint32) != y[start].view(np.uint32):
        bad.append((s, w, float(yy), float(y[start])))
print("mismatches", len(bad), "of", nseg)
for s, w, a, b in bad[:40]:
    start = s * S
    pk = bp[max(start - Wl, 0) // 256: start // 256]
    print("seg", s, "t=%.3fs" % (start / 48000.0), "w", w, "spec %.9g true %.9g" % (a, b), "window peaks(last 8 blks)", pk[-8:], "min/max in long", pk.min(), pk.max())
