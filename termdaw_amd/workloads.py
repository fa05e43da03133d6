"""Synthetic projects for the BASELINE configs + a backend-neutral project script recorder.

`ProjectScript` records termdaw's Lua API calls (same names, arity and argument order as the
globals registered at /root/reference/src/state.rs:103-157) and replays them into a *backend*: any
object exposing `SampleBank`, `FlowwBank` and `Graph` classes with the reference's method names.
`termdaw_amd.api` (the HIP engine behind the C ABI) is one backend; the CPU oracle's binding is the
other one, but this module never imports it -- callers pass the backend in.

The replay follows State::refresh (state.rs:202-467): banks first, vertices grouped by type in the
fixed order sums, norms, sampleloops, samplemultis, samplelerps, debugsines, synths, adsrs,
bandpasses, then edges in call order, set_output, check_graph, reset_normalize_vertices.

Assets are generated with integer arithmetic only (SplitMix64 + integer envelopes) so that every
machine produces byte-identical PCM; nothing is copied from the reference (it ships no assets).
"""
import math
import os
import struct

import numpy as np

# ------------------------------------------------------------------------------------------------
# deterministic integer asset generators
# ------------------------------------------------------------------------------------------------
_M64 = (1 << 64) - 1


def splitmix64(seed, n):
    """n outputs of SplitMix64 started at `seed` (vectorised; uint64 wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        i = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed & _M64) + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def noise_int16(seed, frames):
    """Interleaved stereo i.i.d. uniform int16 noise, shape [frames, 2] (BASELINE config 2 assets)."""
    z = splitmix64(seed, frames * 2)
    v = (z & np.uint64(0xFFFF)).astype(np.int64) - 32768
    return v.astype(np.int16).reshape(frames, 2)


def snare_int16(seed, frames):
    """Quadratically decaying noise burst (config 1 'snare')."""
    n = noise_int16(seed, frames).astype(np.int64)
    i = np.arange(frames, dtype=np.int64)
    env = (32767 * (frames - i) ** 2) // (frames * frames)
    return ((n * env[:, None]) >> 15).astype(np.int16)


def kick_int16(seed, frames, sr=48000):
    """Decaying triangle sweep 200 Hz -> 40 Hz with a little seeded noise (config 1 'kick')."""
    i = np.arange(frames, dtype=np.int64)
    # phase in 1/2^32 turns; frequency falls linearly
    f_hz_q16 = (200 << 16) - ((160 << 16) * i) // frames
    step = (f_hz_q16 << 16) // sr
    phase = np.cumsum(step) & 0xFFFFFFFF
    tri = np.abs(((phase + (1 << 30)) & 0xFFFFFFFF) - (1 << 31)) - (1 << 30)  # [-2^30, 2^30]
    env = (32767 * (frames - i) ** 2) // (frames * frames)
    body = ((tri >> 15) * env) >> 15
    nz = noise_int16(seed, frames).astype(np.int64) >> 6
    out = np.stack([body + nz[:, 0], body - nz[:, 1]], axis=1)
    return np.clip(out, -32768, 32767).astype(np.int16)


def tone_int16(seed, frames, period=97):
    """Short decaying saw 'pluck' used for sample_multi / sample_lerp tests."""
    i = np.arange(frames, dtype=np.int64)
    saw = ((i * 65536) // period) % 65536 - 32768
    env = (32767 * (frames - i)) // frames
    nz = noise_int16(seed, frames).astype(np.int64) >> 4
    l = (saw * env >> 15) + nz[:, 0]
    r = ((-saw) * env >> 15) + nz[:, 1]
    return np.clip(np.stack([l, r], axis=1), -32768, 32767).astype(np.int16)


def wavetable_bytes(seed, n_frames=64, frame_len=2048, table_seconds=2.0):
    """A wavetable resource in this engine's own format (the reference's lives in the un-vendored sampsyn
    crate): "TDWT", u32 version 1, u32 n_frames, u32 frame_len, f32 table_seconds, then the f32 frames.
    Frame f is a seeded additive mix whose upper harmonics fade in with f."""
    rnd = (splitmix64(seed, 16).astype(np.float64) / 2.0 ** 64)
    i = np.arange(frame_len, dtype=np.float64) / frame_len
    frames = np.zeros((n_frames, frame_len), np.float64)
    for f in range(n_frames):
        w = f / max(n_frames - 1, 1)
        for h in range(1, 9):
            amp = (1.0 / h) * (1.0 if h == 1 else w ** (0.5 * h)) * (0.5 + rnd[h])
            frames[f] += amp * np.sin(2 * np.pi * (h * i + rnd[8 + h - 1]))
        frames[f] /= np.abs(frames[f]).max()
    return b"TDWT" + struct.pack("<IIIf", 1, n_frames, frame_len, table_seconds) + frames.astype("<f4").tobytes()


def write_wav_int16(path, pcm, sr):
    """Canonical 44-byte-header 16-bit PCM RIFF/WAVE (what hound writes for 16-bit int)."""
    pcm = np.ascontiguousarray(pcm, dtype="<i2")
    ch = pcm.shape[1] if pcm.ndim == 2 else 1
    data = pcm.tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE" + b"fmt " + struct.pack(
        "<IHHIIHH", 16, 1, ch, sr, sr * ch * 2, ch * 2, 16) + b"data" + struct.pack("<I", len(data))
    with open(path, "wb") as f:
        f.write(hdr + data)


# ------------------------------------------------------------------------------------------------
# project script recorder (Lua API surface, state.rs:103-157)
# ------------------------------------------------------------------------------------------------

def _vlq(v):
    out = [v & 0x7F]
    v >>= 7
    while v:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    return bytes(reversed(out))


def midi_bytes(events, ppq=480, us_per_quarter=500000, channel=0):
    """A format-0 Standard MIDI File holding `events` = [(t_sec, note, vel)] (vel 0 -> note-off), times
    quantised to ticks.  Returns (bytes, quantised_events) -- the second is what a reader following
    csrc/midi.h must produce: (f32(tick * us_per_quarter / ppq * 1e-6), note, f32(int(vel*127+0.5)) / 127)."""
    trk = bytearray()
    trk += _vlq(0) + bytes([0xFF, 0x51, 0x03]) + int(us_per_quarter).to_bytes(3, "big")
    last = 0
    quant = []
    rows = []
    for t, note, vel in events:
        tick = int(round(float(t) * 1e6 / us_per_quarter * ppq))
        rows.append((tick, int(note) & 0x7F, int(float(vel) * 127.0 + 0.5)))
    rows.sort(key=lambda r: r[0])
    for tick, note, v in rows:
        trk += _vlq(tick - last)
        last = tick
        trk += bytes([(0x90 if v > 0 else 0x80) | channel, note, v if v > 0 else 0x40])
        sec = np.float32(np.float64(tick) * np.float64(us_per_quarter) / np.float64(ppq) * 1e-6)
        quant.append((sec, np.float32(note), np.float32(v) / np.float32(127.0) if v > 0 else np.float32(0.0)))
    trk += _vlq(0) + bytes([0xFF, 0x2F, 0x00])
    data = b"MThd" + (6).to_bytes(4, "big") + (0).to_bytes(2, "big") + (1).to_bytes(2, "big") + int(ppq).to_bytes(2, "big")
    data += b"MTrk" + len(trk).to_bytes(4, "big") + bytes(trk)
    return data, np.array(quant, dtype=np.float32).reshape(-1, 3)

class Asset:
    def __init__(self, pcm, sr=48000, bits=16):
        self.pcm = np.asarray(pcm)
        self.sr = sr
        self.bits = bits
        self.channels = 1 if self.pcm.ndim == 1 else self.pcm.shape[1]


def chunk_count(psr, seconds, bl):
    """state.rs:104: cs = (psr as f32 * seconds / bl as f32).ceil() as usize"""
    v = np.float32(psr) * np.float32(seconds) / np.float32(bl)
    return int(np.ceil(v))


class ProjectScript:
    def __init__(self, project_samplerate=48000, buffer_length=1024):
        self.psr = project_samplerate       # config.rs:62-64 default 44100; BASELINE configs use 48000
        self.bl = buffer_length             # config.rs:58-60 default 1024
        self.cs = 0
        self.render_sr = 48000              # main.rs:89
        self.bd = 16                        # main.rs:90
        self.output_file = "outp.wav"       # main.rs:92
        self.output_vertex = ""
        self.assets = {}                    # path -> Asset
        self.event_files = {}               # path -> ndarray [n,3] (t_sec, note, vel)
        self.calls = {k: [] for k in (
            "load_sample", "load_resource", "load_midi_floww", "add_sum", "add_normalize", "add_sampleloop",
            "add_sample_multi", "add_sample_lerp", "add_debug_sine", "add_synth", "add_sampsyn", "add_adsr",
            "add_bandpass", "connect")}
        self.resources = {}                 # path -> bytes (load_resource)
        self.script_order = []              # (fn, args) in call order, for to_lua()

    # -- settings --
    def set_length(self, seconds):
        self.cs = chunk_count(self.psr, seconds, self.bl)
        self.script_order.append(("set_length", (float(seconds),)))

    def set_render_samplerate(self, sr):
        self.render_sr = sr
        self.script_order.append(("set_render_samplerate", (sr,)))

    def set_render_bitdepth(self, bd):
        self.bd = bd
        self.script_order.append(("set_render_bitdepth", (bd,)))

    def set_output_file(self, f):
        self.output_file = f
        self.script_order.append(("set_output_file", (f,)))

    def set_output(self, v):
        self.output_vertex = v
        self.script_order.append(("set_output", (v,)))

    def _rec(self, fn, *args):
        self.calls[fn].append(args)
        self.script_order.append((fn, args))

    # -- resources --
    def load_sample(self, name, path, mode=""):
        self._rec("load_sample", name, path, mode)

    def load_midi_floww(self, name, path):
        self._rec("load_midi_floww", name, path)

    def load_resource(self, name, path):
        self._rec("load_resource", name, path)

    # -- graph --
    def add_sum(self, name, gain, angle):
        self._rec("add_sum", name, gain, angle)

    def add_normalize(self, name, gain, angle):
        self._rec("add_normalize", name, gain, angle)

    def add_sampleloop(self, name, gain, angle, sample):
        self._rec("add_sampleloop", name, gain, angle, sample)

    def add_sample_multi(self, name, gain, angle, sample, floww, note):
        self._rec("add_sample_multi", name, gain, angle, sample, floww, note)

    def add_sample_lerp(self, name, gain, angle, sample, floww, note, lerp_len):
        self._rec("add_sample_lerp", name, gain, angle, sample, floww, note, lerp_len)

    def add_debug_sine(self, name, gain, angle, floww):
        self._rec("add_debug_sine", name, gain, angle, floww)

    def add_synth(self, name, gain, angle, floww, sq_vel, sq_z, sq_adsr, tf_vel, tf_z, tf_adsr, tr_vel, tr_adsr):
        self._rec("add_synth", name, gain, angle, floww, sq_vel, sq_z, list(sq_adsr), tf_vel, tf_z,
                  list(tf_adsr), tr_vel, list(tr_adsr))

    def add_sampsyn(self, name, gain, angle, floww, adsr, resource):
        self._rec("add_sampsyn", name, gain, angle, floww, list(adsr), resource)

    def add_adsr(self, name, gain, angle, wet, floww, use_off, use_max, note, adsr):
        self._rec("add_adsr", name, gain, angle, wet, floww, use_off, use_max, note, list(adsr))

    def add_bandpass(self, name, gain, angle, wet, lo_hz, hi_hz, pass_):
        self._rec("add_bandpass", name, gain, angle, wet, lo_hz, hi_hz, pass_)

    def connect(self, a, b):
        self._rec("connect", a, b)

    # -- State::refresh (state.rs:202-467) --
    def build(self, backend):
        """Returns (sb, fb, g) built exactly in the reference's order, or raises on a failed refresh."""
        sb = backend.SampleBank(self.psr)
        for name, path, mode in self.calls["load_sample"]:
            a = self.assets[path]
            sb.add_decoded(name, a.pcm.astype(np.float32).reshape(-1), a.channels, a.sr, a.bits, mode)
        fb = backend.FlowwBank(self.psr, self.bl)
        for name, path in self.calls["load_midi_floww"]:
            fb.add_events(name, self.event_files[path])
        g = backend.Graph(self.bl, self.psr)
        if hasattr(g, "set_option"):   # (the GPU engine; the oracle has one sine)
            # A bare td_graph evaluates glibc's sinf bit for bit (engine option sine_mode 1); the synthetic projects are built the
            # way the front-end builds them -- the tolerance-class sine, what the bench times and most parity tests pin -- unless
            # a test asks: p.sine_mode = 1, or set_option("sine_mode", 1) on the built graph.
            g.set_option("sine_mode", getattr(self, "sine_mode", 0))

        def sidx(s, vname):
            i = sb.get_index(s)
            if i is None:
                raise KeyError("Could not get sample index for vertex \"%s\"." % vname)
            return i

        def fidx(f, vname):
            i = fb.get_index(f)
            if i is None:
                raise KeyError("Could not get floww index for vertex \"%s\"." % vname)
            return i

        for name, gain, angle in self.calls["add_sum"]:
            g.add_sum(name, gain, angle)
        for name, gain, angle in self.calls["add_normalize"]:
            g.add_normalize(name, gain, angle)
        for name, gain, angle, s in self.calls["add_sampleloop"]:
            g.add_sampleloop(name, gain, angle, sidx(s, name))
        for name, gain, angle, s, f, note in self.calls["add_sample_multi"]:
            g.add_sample_multi(name, gain, angle, sidx(s, name), fidx(f, name), note)
        for name, gain, angle, s, f, note, ll in self.calls["add_sample_lerp"]:
            g.add_sample_lerp(name, gain, angle, sidx(s, name), fidx(f, name), note, ll)
        for name, gain, angle, f in self.calls["add_debug_sine"]:
            g.add_debug_sine(name, gain, angle, fidx(f, name))
        for name, gain, angle, f, sv, sz, sa, tv, tz, ta, rv, ra in self.calls["add_synth"]:
            g.add_synth(name, gain, angle, fidx(f, name), sv, sz, sa, tv, tz, ta, rv, ra)
        res = dict(self.calls["load_resource"])
        for name, gain, angle, f, conf, resource in self.calls["add_sampsyn"]:
            if resource not in res:
                raise KeyError("Could not find resource named %s!" % resource)
            g.add_sampsyn(name, gain, angle, fidx(f, name), conf, self.resources[res[resource]])
        for name, gain, angle, wet, f, uo, um, note, conf in self.calls["add_adsr"]:
            g.add_adsr(name, gain, angle, wet, fidx(f, name), uo, um, note, conf)
        for name, gain, angle, wet, lo, hi, p in self.calls["add_bandpass"]:
            g.add_bandpass(name, gain, angle, wet, lo, hi, p)
        for a, b in self.calls["connect"]:
            g.connect(a, b)
        g.set_output(self.output_vertex)
        if not g.check_graph():
            raise RuntimeError("TermDaw: graph check failed!")
        g.reset_normalize_vertices()
        return sb, fb, g

    def render(self, backend, scan=False, built=None, **kw):
        """refresh -> [scan_exact] -> render; returns (pcm, f32) like State::render would write."""
        sb, fb, g = built if built is not None else self.build(backend)
        if scan:
            g.true_normalize_scan(sb, fb, self.cs)
        return g.render_all(sb, fb, self.cs, self.bd, **kw)

    # -- Lua text + files for the project front-end --
    def to_lua(self, asset_dir):
        """Writes assets (WAV) and event lists (.flw text) under asset_dir; returns the Lua source."""
        os.makedirs(asset_dir, exist_ok=True)
        paths = {}
        for p, a in self.assets.items():
            out = os.path.join(asset_dir, p.replace("/", "_") + ".wav")
            write_wav_int16(out, a.pcm, a.sr)
            paths[p] = out
        for p, ev in self.event_files.items():
            out = os.path.join(asset_dir, p.replace("/", "_") + ".flw")
            with open(out, "w") as f:
                for t, n, v in np.asarray(ev, dtype=np.float32).reshape(-1, 3):
                    f.write("%s %s %s\n" % (float(t).hex(), float(n).hex(), float(v).hex()))
            paths[p] = out

        for p, blob in self.resources.items():
            out = os.path.join(asset_dir, p.replace("/", "_") + ".tdwt")
            with open(out, "wb") as f:
                f.write(blob)
            paths[p] = out

        def lit(x):
            if isinstance(x, bool):
                return "true" if x else "false"
            if isinstance(x, str):
                return '"%s"' % x
            if isinstance(x, (list, tuple)):
                return "{ " + ", ".join(lit(y) for y in x) + " }" if len(x) else "{}"
            if isinstance(x, (int, np.integer)):
                return str(int(x))
            return repr(float(x))

        lines = ["-- generated by termdaw_amd.workloads.ProjectScript.to_lua"]
        for fn, args in self.script_order:
            if fn in ("load_sample", "load_midi_floww", "load_resource"):
                args = (args[0], paths[args[1]]) + tuple(args[2:])
            lines.append("%s(%s);" % (fn, ", ".join(lit(a) for a in args)))
        return "\n".join(lines) + "\n"


# ------------------------------------------------------------------------------------------------
# BASELINE configs
# ------------------------------------------------------------------------------------------------
def config1(seconds=3.0):
    """README example: 2 x sampleloop -> normalize 'sum' (README.md:94-110), render_sr forced to 48000."""
    p = ProjectScript(48000, 1024)
    p.assets["snare"] = Asset(snare_int16(1, 12000))
    p.assets["kick"] = Asset(kick_int16(2, 24000))
    p.set_length(seconds)
    p.set_render_samplerate(48000)
    p.set_render_bitdepth(16)
    p.set_output_file("outp.wav")
    p.load_sample("snare", "snare", "")
    p.load_sample("kick", "kick", "")
    p.add_sampleloop("one", 1.0, 0.0, "snare")
    p.add_sampleloop("two", 1.0, 0.0, "kick")
    p.add_normalize("sum", 1.0, 0.0)
    p.connect("one", "sum")
    p.connect("two", "sum")
    p.set_output("sum")
    return p


def config2(seconds=60.0, n_src=64, seed_offset=0, base_len=48000):
    """64 x sampleloop -> one normalize (headline config)."""
    p = ProjectScript(48000, 1024)
    p.set_length(seconds)
    p.set_render_samplerate(48000)
    p.set_render_bitdepth(16)
    for k in range(n_src):
        nm = "s%02d" % k
        p.assets[nm] = Asset(noise_int16(1000 + k + seed_offset, base_len + 977 * k))
        p.load_sample(nm, nm, "")
    for k in range(n_src):
        nm = "s%02d" % k
        gain = float(np.float32(0.5) + np.float32(k) / np.float32(64.0))
        angle = float(np.float32(-90.0) + np.float32(180.0) * np.float32(k) / np.float32(max(n_src - 1, 1)))
        p.add_sampleloop("v" + nm, gain, angle, nm)
    p.add_normalize("sum", 1.0, 0.0)
    for k in range(n_src):
        p.connect("vs%02d" % k, "sum")
    p.set_output("sum")
    return p


HIT_ADSR = [0.001, 0.02, 0.0, 0.0, 0.0, 0.0]    # project.lua:35
NOTE_ADSR = [0.01, 0.1, 0.8, 5.0, 0.2, 0.5]     # project.lua:36
STD_ADSR = [0.01, 1.0, 1.0, 1.0, 1.0, 0.4]      # project.lua:37


def config3(seconds=60.0, voices=32, variant=0):
    """synth (32 simultaneous voices, 3 oscillators) -> adsr -> bandpass -> normalize.
    variant: the projects of a batch differ like config 5's do by their seeds -- here the chord is transposed by
    variant % 7 semitones and the velocities scaled (same shape, same launches)."""
    p = ProjectScript(48000, 1024)
    p.set_length(seconds)
    p.set_render_samplerate(48000)
    p.set_render_bitdepth(16)
    ev = []
    t = 0.0
    while t < seconds:
        for j in range(voices):
            ev.append((t, 36.0 + j + variant % 7, (0.25 + 0.02 * j) * (1.0 - 0.03 * (variant % 5))))
        for j in range(voices):
            ev.append((t + 1.5, 36.0 + j + variant % 7, 0.0))
        t += 2.0
    ev.sort(key=lambda e: e[0])
    p.event_files["notes"] = np.array(ev, dtype=np.float32)
    drum = [(0.5 * i, 38.0, 0.9) for i in range(int(seconds / 0.5))]
    p.event_files["drum"] = np.array(drum, dtype=np.float32)
    p.load_midi_floww("notes", "notes")
    p.load_midi_floww("drum", "drum")
    p.add_synth("syn", 1.0, 0.0, "notes", 0.4, 0.3, HIT_ADSR, 1.0, 0.8, NOTE_ADSR, 0.5, NOTE_ADSR)
    p.add_adsr("env", 1.0, 0.0, 1.0, "drum", False, True, -1, [0.01, 0.1, 0.8, 0.1, 0.2, 0.01])
    p.add_bandpass("band", 1.0, 0.0, 1.0, 200.0, 4000.0, True)
    p.add_normalize("sum", 1.0, 0.0)
    p.connect("syn", "env")
    p.connect("env", "band")
    p.connect("band", "sum")
    p.set_output("sum")
    return p


def drum_project(seconds=4.0, bl=1024):
    """Small project covering sample_multi / sample_lerp / sum / adsr-vertex / bandpass chains."""
    p = ProjectScript(48000, bl)
    p.set_length(seconds)
    p.assets["pluck"] = Asset(tone_int16(11, 9000))
    p.assets["kick"] = Asset(kick_int16(12, 15000))
    p.assets["bg"] = Asset(noise_int16(13, 20011))
    p.load_sample("pluck", "pluck", "")
    p.load_sample("kick", "kick", "mix-down")
    p.load_sample("bg", "bg", "normalize-seperate")
    n = int(seconds / 0.125)
    hits = []
    for i in range(n):
        t = 0.125 * i + 0.003
        hits.append((t, 60.0 + (i % 3), 0.3 + 0.05 * (i % 7)))
        if i % 4 == 1:
            hits.append((t, 61.0, 0.8))            # second hit on the same frame (Q10)
        if i % 5 == 2:
            hits.append((t + 0.05, 60.0, 0.0))      # note-off, ignored by drum pulls
    hits.sort(key=lambda e: e[0])
    p.event_files["hits"] = np.array(hits, dtype=np.float32)
    kicks = [(0.5 * i, 36.0, 1.0) for i in range(int(seconds / 0.5))]
    p.event_files["kicks"] = np.array(kicks, dtype=np.float32)
    p.load_midi_floww("hits", "hits")
    p.load_midi_floww("kicks", "kicks")
    p.add_sample_multi("multi", 0.8, 20.0, "pluck", "hits", -1)
    p.add_sample_multi("multi60", 1.2, -35.0, "pluck", "hits", 60)
    p.add_sample_lerp("lerp", 1.0, 0.0, "kick", "kicks", -1, 40)
    p.add_sample_lerp("lerpfast", 0.7, 50.0, "pluck", "hits", -1, 400)
    p.add_sampleloop("bg", 0.25, 0.0, "bg")
    dip = 0.3
    p.add_adsr("duck", 1.0, 0.0, 1.0, "kicks", False, False, -1,
               [1.0, 0.01, dip, 0.2, dip, 0.0, 0.0, 0.05, 1.0])   # examples/neg-adsr-env-example.lua:15-17
    p.add_bandpass("kickband", 1.0, 0.0, 1.0, 0.0, 50.0, True)    # project.lua:47
    p.add_bandpass("band", 1.0, 0.0, 1.0, 1000.0, 0.0, True)      # project.lua:46
    p.add_sum("drums", 0.9, -10.0)
    p.add_normalize("sum", 1.0, 0.0)
    p.connect("lerp", "kickband")
    p.connect("kickband", "drums")
    p.connect("multi", "drums")
    p.connect("multi60", "drums")
    p.connect("lerpfast", "drums")
    p.connect("bg", "duck")
    p.connect("duck", "band")
    p.connect("drums", "band")
    p.connect("band", "sum")
    p.set_output("sum")
    return p


def synth_project(seconds=3.0, bl=1024, voices=5):
    """Small tolerance-class project: debug_sine + synth -> adsr(use_off) -> bandpass(cut) -> normalize."""
    p = ProjectScript(48000, bl)
    p.set_length(seconds)
    ev = []
    t = 0.01
    k = 0
    while t < seconds:
        for j in range(voices):
            ev.append((t + 0.0007 * j, 48.0 + 3 * j, 0.3 + 0.1 * j))
        for j in range(voices):
            ev.append((t + 0.4 + 0.0011 * j, 48.0 + 3 * j, 0.0))
        if k % 2 == 0:
            ev.append((t + 0.2, 48.0, 0.5))     # re-strike while held
            ev.append((t + 0.2, 99.0, 0.0))     # off for a note that is not sounding
        t += 0.7
        k += 1
    ev.sort(key=lambda e: e[0])
    p.event_files["notes"] = np.array(ev, dtype=np.float32)
    p.load_midi_floww("notes", "notes")
    p.add_debug_sine("sine", 0.2, 30.0, "notes")
    p.add_synth("syn", 0.8, -15.0, "notes", 0.4, 0.3, HIT_ADSR, 1.0, 0.8, NOTE_ADSR, 0.5, STD_ADSR)
    p.add_adsr("env", 1.0, 0.0, 0.9, "notes", True, True, -1, NOTE_ADSR)
    p.add_adsr("env48", 1.0, 0.0, 1.0, "notes", True, False, 48, NOTE_ADSR)
    p.add_bandpass("cut", 1.1, 0.0, 1.0, 300.0, 5000.0, False)
    p.add_sum("mix", 1.0, 0.0)
    p.add_normalize("sum", 0.7, 0.0)
    p.connect("sine", "env")
    p.connect("syn", "env48")
    p.connect("env", "mix")
    p.connect("env48", "mix")
    p.connect("mix", "cut")
    p.connect("cut", "sum")
    p.set_output("sum")
    return p


def config4(seconds=60.0, depth=252, variant=0):
    """Deep chain (BASELINE config 4): wavetable synth (sampsyn, 64 x 2048 table, seed 7) + sample_lerp over a
    44.1 kHz asset (hits every 0.25 s, lerp_len 40) -> sum -> `depth` single-input vertices alternating
    sum(gain 1.41, angle +-1) / bandpass(20 Hz, 18 kHz) / adsr -> normalize = depth + 4 vertices.
    The wavetable oscillator and the 44.1 k -> 48 k resample sit on un-vendored crates (sampsyn, rubato) in
    the reference; here they are this engine's own documented stand-ins (parity unpinned vs the reference,
    bit-exact vs the oracle).  variant: asset seeds 7 + variant (the projects of a batch, like config 5's seed offsets)."""
    p = ProjectScript(48000, 1024)
    p.set_length(seconds)
    p.assets["kick"] = Asset(kick_int16(7 + variant, 18375, sr=44100), sr=44100)
    p.load_sample("kick", "kick", "")
    p.resources["table"] = wavetable_bytes(7 + variant)
    p.load_resource("table", "table")
    hits = [(0.25 * i, 36.0, 0.9) for i in range(int(seconds / 0.25))]
    p.event_files["hits"] = np.array(hits, dtype=np.float32)
    notes = []
    t = 0.0
    while t < seconds:
        for j in range(4):
            notes.append((t + 0.01 * j, 50.0 + 4 * j, 0.4))
        for j in range(4):
            notes.append((t + 0.8 + 0.01 * j, 50.0 + 4 * j, 0.0))
        t += 1.0
    notes.sort(key=lambda e: e[0])
    p.event_files["notes"] = np.array(notes, dtype=np.float32)
    p.load_midi_floww("hits", "hits")
    p.load_midi_floww("notes", "notes")
    p.add_sampsyn("syn", 0.5, 0.0, "notes", STD_ADSR, "table")
    p.add_sample_lerp("lerp", 1.0, 0.0, "kick", "hits", -1, 40)
    p.add_sum("mix", 1.0, 0.0)
    p.connect("syn", "mix")
    p.connect("lerp", "mix")
    prev = "mix"
    for i in range(depth):
        name = "c%03d" % i
        if i % 3 == 0:
            # a 1 degree pan already costs x0.713 / x0.701 (constant-power law, quirk Q1): gain 1.41 keeps the chain near unity
            p.add_sum(name, 1.41, 1.0 if (i // 3) % 2 == 0 else -1.0)
        elif i % 3 == 1:
            p.add_bandpass(name, 1.0, 0.0, 1.0, 20.0, 18000.0, True)
        else:
            p.add_adsr(name, 1.0, 0.0, 0.5, "hits", False, True, -1, [0.01, 0.1, 0.8, 0.1, 0.2, 0.01])
        p.connect(prev, name)
        prev = name
    p.add_normalize("sum", 1.0, 0.0)
    p.connect(prev, "sum")
    p.set_output("sum")
    return p
