"""Independent numpy (float32) twin of part of the reference algorithm, used to cross-check the C++
oracle: whole-timeline, vectorised, written separately from oracle/termdaw_oracle.cpp.  Covers
sampleloop, sample_multi, sample_lerp, sum, normalize (running peak), bandpass, pan/gain and the
16-bit quantiser -- the bit-exact class.  Reference lines are cited next to each piece.
"""
import ctypes

import numpy as np

_libm = ctypes.CDLL("libm.so.6")
for _n in ("cosf", "sinf", "powf"):
    getattr(_libm, _n).restype = ctypes.c_float
    getattr(_libm, _n).argtypes = [ctypes.c_float] * (2 if _n == "powf" else 1)
f32 = np.float32


def load_sample(pcm, mode=""):
    """sample.rs:262-303: ints as f32 (unscaled), de-interleave, mode, peak normalise."""
    x = np.asarray(pcm).astype(np.float32)
    l, r = x[:, 0].copy(), x[:, 1].copy()
    if mode == "mix-down":
        m = l + r
        m = m * (f32(1.0) / np.abs(m).max())
        return m, m.copy()
    if mode == "normalize-seperate":
        return l * (f32(1.0) / np.abs(l).max()), r * (f32(1.0) / np.abs(r).max())
    if mode == "left":
        r = l.copy()
    elif mode == "right":
        l = r.copy()
    s = f32(1.0) / max(np.abs(l).max(), np.abs(r).max())
    return l * s, r * s


def pan_gain(l, r, gain, angle):
    """sample.rs:97-114 (pan first, then gain; both thresholded)."""
    gain, angle = f32(gain), f32(min(max(angle, -90.0), 90.0))
    if not abs(angle) < f32(0.001):
        a = f32(f32(angle * f32(0.5)) * f32(0.01745329))
        k = f32(0.70710678118654752440)
        l = l * f32(k * f32(f32(_libm.cosf(a)) + f32(_libm.sinf(a))))
        r = r * f32(k * f32(f32(_libm.cosf(a)) - f32(_libm.sinf(a))))
    if not abs(f32(gain - f32(1.0))) < f32(0.001):
        l, r = l * gain, r * gain
    return l, r


def frame_of(t, sr):
    return int(f32(t) * f32(sr))   # floww.rs:75 (non-negative inputs only in the twin)


def drum_hits(events, sr, note):
    """floww.rs:99-121 for time-sorted events: first on-event of each frame; then the vertex' note filter."""
    hits, seen = [], set()
    for t, n, v in np.asarray(events, np.float32).reshape(-1, 3):
        f = frame_of(t, sr)
        if f in seen or not v > f32(0.001):
            continue
        seen.add(f)
        if note < 0 or abs(f32(n) - f32(note)) < f32(0.01):
            hits.append((f, f32(v)))
    return hits


def render(p, scan=False):
    """refresh -> [scan] -> render for projects using only the kinds above. Returns (pcm int16 [N,2], f32 [N,2])."""
    N, bl, sr = p.cs * p.bl, p.bl, p.psr
    samples = {name: load_sample(p.assets[path].pcm, mode) for name, path, mode in p.calls["load_sample"]}
    verts, inputs = {}, {}
    for kind in ("add_sum", "add_normalize", "add_sampleloop", "add_sample_multi", "add_sample_lerp", "add_bandpass"):
        for args in p.calls[kind]:
            verts[args[0]] = (kind, args)
            inputs[args[0]] = []
    has_input = ("add_sum", "add_normalize", "add_bandpass")
    for a, b in p.calls["connect"]:
        if a in verts and b in verts and a != b and verts[b][0] in has_input:
            inputs[b].append(a)   # (no cycles in the twin's projects)
    n = np.arange(N)
    norm_max = {}

    def run(name, is_scan, memo):
        if name in memo:
            return memo[name]
        kind, a = verts[name]
        gain, angle = a[1], a[2]
        if kind in has_input:
            l, r = np.zeros(N, f32), np.zeros(N, f32)
            for src in inputs[name]:
                sl, sr_ = run(src, is_scan, memo)
                l, r = l + sl, r + sr_
        if kind == "add_sampleloop":
            sl, sr_ = samples[a[3]]
            idx = n % len(sl)   # extensions.rs:337 (t restarts at 0: fresh state / set_time(0))
            l, r = sl[idx], sr_[idx]
        elif kind == "add_sample_multi":   # extensions.rs:344-381
            sl, sr_ = samples[a[3]]
            l, r = np.zeros(N, f32), np.zeros(N, f32)
            for f, v in drum_hits(p.event_files[dict(p.calls["load_midi_floww"])[a[4]]], sr, a[5]):
                m = min(len(sl), N - f)
                if m > 0:
                    l[f:f + m] += sl[:m] * v
                    r[f:f + m] += sr_[:m] * v
        elif kind == "add_sample_lerp":   # extensions.rs:384-421
            sl, sr_ = samples[a[3]]
            L, ll = len(sl), max(a[6], 0)
            hits = [(None, f32(0.0))] + drum_hits(p.event_files[dict(p.calls["load_midi_floww"])[a[4]]], sr, a[5])
            l, r = np.zeros(N, f32), np.zeros(N, f32)
            for j, (f, v) in enumerate(hits):
                start = 0 if f is None else f
                end = hits[j + 1][0] if j + 1 < len(hits) else N
                if start >= N:
                    break
                m = n[start:min(end, N)]
                pos = np.minimum(m - (0 if f is None else f), L - 1)
                pl, pr = sl[pos] * v, sr_[pos] * v
                if f is not None and ll > 0:
                    d = m - f
                    fade = d < ll
                    if fade.any():
                        gf, gv = hits[j - 1]
                        gpos = np.minimum(m - (0 if gf is None else gf), L - 1)
                        t = ((ll - 1 - d).astype(np.float32) / f32(ll))
                        gl, gr = sl[gpos] * gv, sr_[gpos] * gv
                        pl = np.where(fade, gl * t + pl * (f32(1.0) - t), pl)
                        pr = np.where(fade, gr * t + pr * (f32(1.0) - t), pr)
                l[start:min(end, N)], r[start:min(end, N)] = pl, pr
        elif kind == "add_normalize":   # extensions.rs:321-329, initial max 1e-6 (state.rs:467)
            pk = np.maximum(np.abs(l), np.abs(r)).reshape(-1, bl).max(axis=1)
            if is_scan:
                cur = norm_max.setdefault(name, f32(0.000001))
                norm_max[name + "/scan"] = max(f32(0.0), pk.max())
                scale = np.full(len(pk), f32(1.0) / cur, f32)
            else:
                run_max = np.maximum.accumulate(np.concatenate([[norm_max.get(name, f32(0.000001))], pk]).astype(f32))[1:]
                norm_max[name] = run_max[-1]
                scale = f32(1.0) / run_max
            s = np.repeat(scale, bl)
            l, r = l * s, r * s
        elif kind == "add_bandpass":   # extensions.rs:654-689 (sequential; wet gates only)
            wet, lo, hi, pas = a[3], a[4], a[5], a[6]

            def gamma(hz):
                co = f32(min(max(hz, 0.0), 20000.0))
                return f32(f32(1.0) - f32(_libm.powf(f32(2.718281828459045), f32(f32(f32(-2.0) * f32(3.14159274)) * co) / f32(sr))))
            lg, hg = gamma(lo), gamma(hi)
            if not (wet < 1e-4 or (lg == 0 and hg == 0)):
                lmul, hmul = f32(0.0 if lg == 0 else 1.0), f32(0.0 if hg == 0 else 1.0)
                ll_, lr_, hl_, hr_ = l[0], r[0], l[0], r[0]
                ol, or_ = np.empty(N, f32), np.empty(N, f32)
                for i in range(N):
                    x, y = l[i], r[i]
                    ll_ = f32(ll_ + f32(lg * f32(x - ll_)))
                    lr_ = f32(lr_ + f32(lg * f32(y - lr_)))
                    hl_ = f32(hl_ + f32(hg * f32(x - hl_)))
                    hr_ = f32(hr_ + f32(hg * f32(y - hr_)))
                    cutl = f32(f32(f32(lmul * ll_) + f32(hmul * f32(x - hl_))) * f32(0.5))
                    cutr = f32(f32(f32(lmul * lr_) + f32(hmul * f32(y - hr_))) * f32(0.5))
                    if pas:
                        ol[i], or_[i] = f32(f32(cutl * f32(0.0)) + f32(f32(x - cutl) * f32(1.0))), f32(f32(cutr * f32(0.0)) + f32(f32(y - cutl) * f32(1.0)))
                    else:
                        ol[i], or_[i] = f32(f32(cutl * f32(1.0)) + f32(f32(x - cutl) * f32(0.0))), f32(f32(cutr * f32(1.0)) + f32(f32(y - cutl) * f32(0.0)))
                l, r = ol, or_
        memo[name] = pan_gain(l, r, gain, angle)
        return memo[name]

    with np.errstate(all="ignore"):
        if scan:
            run(p.output_vertex, True, {})
            for k in [k for k in norm_max if k.endswith("/scan")]:
                norm_max[k[:-5]] = norm_max.pop(k)
        l, r = run(p.output_vertex, False, {})
        f = np.stack([l, r], axis=1).astype(f32)
        q = f * f32(32767.0)                                   # state.rs:515-522
        q = np.where(np.isnan(q), f32(0), np.clip(q, -32768.0, 32767.0))
        pcm = np.trunc(q).astype(np.int16)
    return pcm, f


# ------------------------------------------------------------------------------------------------
# second part: event-driven float kinds (debug_sine, synth, adsr vertex), per-sample loops in np.float32
# scalars; sinf / powf / floorf through libm so that the twin can be compared bit for bit with the oracle
# ------------------------------------------------------------------------------------------------
for _n in ("floorf",):
    getattr(_libm, _n).restype = ctypes.c_float
    getattr(_libm, _n).argtypes = [ctypes.c_float]
PI32 = f32(3.14159274101257324)


def fmin(a, b):
    """Rust f32::min / C fminf: a NaN operand is ignored (adsr.rs:72 relies on (t / 0.0).min(1.0) == 1.0, quirk Q6)."""
    return f32(np.fmin(a, b))


def fmax(a, b):
    return f32(np.fmax(a, b))


def adsr_conf(arr):
    """build_adsr_conf (adsr.rs:94-114) -> 9 np.float32"""
    a = [f32(x) for x in arr]
    if len(a) == 0:
        return [f32(0)] * 9
    if len(a) == 6:
        return [f32(0), a[0], f32(1), a[1], a[2], a[3], a[4], a[5], f32(0)]
    assert len(a) == 9
    return a


def _lerp(a, b, t):
    return f32(a + f32(t * f32(b - a)))


def ads_internal(c, t):   # adsr.rs:46-60
    std, a_s, a_v, d_s, d_v, s_s, s_v, r_s, r_v = c
    if t <= a_s:
        return _lerp(std, a_v, f32(t / a_s))
    if t <= f32(a_s + d_s):
        return _lerp(a_v, d_v, f32(f32(t - a_s) / d_s))
    if t <= f32(f32(a_s + d_s) + s_s):
        return _lerp(d_v, s_v, f32(f32(f32(t - a_s) - d_s) / s_s))
    return f32(-1000.0)


def apply_ads(c, t):
    r = ads_internal(c, t)
    return c[6] if r <= f32(-1.0) else r


def apply_r(c, t, old):
    return _lerp(old, c[8], fmin(f32(t / c[7]), f32(1.0)))


def apply_adsr(c, t):
    r = ads_internal(c, t)
    if r <= f32(-1.0):
        return _lerp(c[6], c[8], fmin(f32(f32(f32(f32(t - c[1]) - c[3]) - c[5]) / c[7]), f32(1.0)))
    return r


def apply_r_rt(c, t, rt):
    return apply_r(c, t, apply_ads(c, rt))


def events_by_frame(events, sr):
    """frame -> [(on, note, vel)] in list order (get_block_simple, floww.rs:124-141, sorted events)."""
    out = {}
    for t, n, v in np.asarray(events, np.float32).reshape(-1, 3):
        out.setdefault(frame_of(t, sr), []).append((bool(v > f32(0.001)), f32(n), f32(v)))
    return out


def hz_of(note):
    return f32(f32(440.0) * f32(_libm.powf(f32(2.0), f32(f32(note - f32(69.0)) / f32(12.0)))))


def debug_sine(events, N, bl, sr):   # extensions.rs:423-457
    ev = events_by_frame(events, sr)
    notes, out = [], np.zeros(N, f32)
    with np.errstate(all="ignore"):
        for m in range(N):
            for on, note, vel in ev.get(m, []):
                if on:
                    for nv in notes:
                        if abs(f32(nv[0] - note)) < f32(0.001):
                            nv[1] = vel
                            break
                    else:
                        notes.append([note, vel])
                else:
                    notes = [x for x in notes if abs(f32(x[0] - note)) > f32(0.001)]
            acc = f32(0)
            time = f32(f32(m) / f32(sr))
            for note, vel in notes:
                acc = f32(acc + f32(f32(_libm.sinf(f32(f32(f32(time * hz_of(note)) * f32(2.0)) * PI32))) * vel))
            out[m] = acc
    return out


def synth(events, N, bl, sr, sq, tf, tr):   # extensions.rs:460-529; sq/tf/tr = (volume, param, conf9)
    ev = events_by_frame(events, sr)
    maxv = lambda c: fmax(fmax(fmax(fmax(c[0], c[2]), c[4]), c[6]), c[8])   # noqa: E731
    with np.errstate(all="ignore"):
        mult = f32(f32(1.0) / f32(f32(f32(sq[0] * maxv(sq[2])) + f32(tf[0] * maxv(tf[2]))) + f32(tr[0] * maxv(tr[2]))))
        release_sec = f32(0)
        if sq[0] > 0: release_sec = sq[2][7]
        if tf[0] > 0: release_sec = fmax(release_sec, tf[2][7])
        if tr[0] > 0: release_sec = fmax(release_sec, tr[2][7])
        notes, out = [], np.zeros(N, f32)   # [note, vel, env_t, rel_t]
        for m in range(N):
            i = m % bl
            off = f32(f32(i) / f32(sr))
            for on, note, vel in ev.get(m, []):
                if on:
                    notes.append([note, vel, f32(-off), f32(0)])
                else:
                    notes = [x for x in notes if abs(f32(x[0] - note)) > f32(0.001) or x[3] == 0]
                    for x in notes:
                        if abs(f32(x[0] - note)) > f32(0.001):
                            continue
                        assert x[3] == 0
                        x[3] = f32(x[2] + off)
                        x[2] = f32(-off)
            time = f32(f32(m) / f32(sr))
            acc = f32(0)
            for note, vel, env_t, rel_t in notes:
                env_time = f32(env_t + off)
                hz = hz_of(note)
                env = (lambda c: apply_ads(c, env_time)) if rel_t == 0 else (lambda c: apply_r_rt(c, env_time, rel_t))
                s = f32(0)
                sn = f32(_libm.sinf(f32(f32(f32(time * hz) * f32(2.0)) * PI32)))
                if sq[0] > 0:
                    z = sq[1]
                    osc = f32(fmin(fmax(sn, f32(-z)), z) * f32(f32(1.0) / z))
                    s = f32(s + f32(f32(f32(osc * vel) * env(sq[2])) * sq[0]))
                if tf[0] > 0:
                    z = tf[1]
                    osc = f32(f32(fmin(sn, z) + f32(f32(f32(1.0) - z) / f32(2.0))) * f32(f32(2.0) / f32(f32(1.0) + z)))
                    s = f32(s + f32(f32(f32(osc * vel) * env(tf[2])) * tf[0]))
                if tr[0] > 0:
                    th = f32(time * hz)
                    osc = f32(f32(f32(4.0) * abs(f32(th - f32(_libm.floorf(f32(th + f32(0.5))))))) - f32(1.0))
                    s = f32(s + f32(f32(f32(osc * vel) * env(tr[2])) * tr[0]))
                s = f32(s * mult)
                acc = f32(acc + s)
            out[m] = acc
            if i == bl - 1:   # end of block: clocks advance, finished voices leave
                for x in notes:
                    x[2] = f32(x[2] + f32(f32(bl) / f32(sr)))
                notes = [x for x in notes if x[3] == 0 or x[2] <= release_sec]
    return out


def adsr_vertex(x, events, bl, sr, wet, use_off, use_max, note, conf):   # extensions.rs:593-651
    """x: [N] mono input (both channels get the same factor); returns the per-frame multiplier applied."""
    N = len(x)
    wet = f32(wet)
    vel_out = np.ones(N, f32)
    if wet < f32(0.0001):
        return vel_out
    maxmul = f32(1.0 if use_max else 0.0)
    minmul = f32(f32(1.0) - maxmul)
    p, g = [f32(0), f32(0), f32(0)], [f32(0), f32(0), f32(0)]
    ev = events_by_frame(events, sr)
    drum = {}
    for t, n, v in np.asarray(events, np.float32).reshape(-1, 3):   # first on-event per frame (floww.rs:99-121)
        f = frame_of(t, sr)
        if f not in drum and v > f32(0.001):
            drum[f] = (f32(n), f32(v))
    with np.errstate(all="ignore"):
        for m in range(N):
            i = m % bl
            off = f32(f32(i) / f32(sr))
            skip = False
            if use_off:
                for on, n, v in ev.get(m, []):
                    if note >= 0 and abs(f32(f32(note) - n)) > f32(0.01):
                        continue
                    if on:
                        g = list(p)
                        p = [f32(-off), v, f32(0)]
                    elif g[2] == 0:
                        g[0] = f32(-off)
                        g[2] = f32(apply_ads(conf, f32(g[0] + off)) * g[1])
                    else:
                        p[0] = f32(-off)
                        p[2] = f32(apply_ads(conf, f32(p[0] + off)) * p[1])
                pv = f32(apply_ads(conf, f32(p[0] + off)) * p[1]) if p[2] == 0 else f32(apply_r(conf, f32(p[0] + off), p[2]) * p[1])
                gv = f32(apply_ads(conf, f32(g[0] + off)) * g[1]) if g[2] == 0 else f32(apply_r(conf, f32(g[0] + off), g[2]) * g[1])
            else:
                if m in drum:
                    n, v = drum[m]
                    if note >= 0 and abs(f32(f32(note) - n)) > f32(0.01):
                        skip = True      # extensions.rs:632-635: `continue` leaves the frame untouched
                    else:
                        g = list(p)
                        p = [f32(-off), v, f32(0)]
                if not skip:
                    pv = f32(apply_adsr(conf, f32(p[0] + off)) * p[1])
                    gv = f32(apply_adsr(conf, f32(g[0] + off)) * g[1])
            if not skip:
                av = f32(f32(fmax(pv, gv) * maxmul) + f32(fmin(pv, gv) * minmul))
                vel_out[m] = _lerp(f32(1.0), av, wet)
            if i == bl - 1:
                p[0] = f32(p[0] + f32(f32(bl) / f32(sr)))
                g[0] = f32(g[0] + f32(f32(bl) / f32(sr)))
    return vel_out
