"""Device-side sample load pipeline (SampleBank::add, sample.rs:224-314) vs the oracle: every load mode, mono
and stereo, odd lengths, all PCM encodings hound reads -- the bank entries must be bit-identical."""
import struct

import numpy as np
import pytest

from termdaw_amd import workloads as W

pytestmark = pytest.mark.gpu
MODES = ["", "left", "right", "loudest", "normalize-seperate", "mix-down"]


def _same_bank_entry(a, b):
    la, ra = a
    lb, rb = b
    assert la.shape == lb.shape and ra.shape == rb.shape
    assert np.array_equal(la.view(np.uint32), lb.view(np.uint32)) and np.array_equal(ra.view(np.uint32), rb.view(np.uint32))


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("channels,n", [(2, 20000), (2, 20001), (1, 7777), (2, 2), (2, 300001)])
def test_add_decoded_modes(gpu_api, oracle, mode, channels, n):
    vals = W.noise_int16(5 + n, (n + 1) // 2 + 1).reshape(-1)[:n].astype(np.float32)
    if mode == "loudest":
        vals[1::2] *= 0.5        # make the decision robust (still exercises the ordered f32 sum)
    out = []
    for be, exc in ((gpu_api, gpu_api.TermdawError), (oracle, ValueError)):
        sb = be.SampleBank(48000)
        try:
            sb.add_decoded("s", vals, channels, 48000, 16, mode)
            out.append(sb.get_sample(sb.get_index("s")))
        except exc as e:
            out.append(str(e))
    if isinstance(out[1], str):
        assert isinstance(out[0], str), "engine accepted what the oracle rejects: %s" % out[1]
    else:
        assert not isinstance(out[0], str), out[0]
        _same_bank_entry(out[0], out[1])


def _write_wav(path, data, channels, sr, bits, is_float=False):
    fmt = 3 if is_float else 1
    bps = bits // 8
    hdr = b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE" + b"fmt " + struct.pack(
        "<IHHIIHH", 16, fmt, channels, sr, sr * channels * bps, channels * bps, bits) + b"data" + struct.pack("<I", len(data))
    open(path, "wb").write(hdr + data)


@pytest.mark.parametrize("bits,is_float", [(8, False), (16, False), (24, False), (32, False), (32, True)])
def test_add_file_encodings(gpu_api, oracle, tmp_path, bits, is_float):
    n = 5000
    rng = W.splitmix64(bits * 7 + is_float, n * 2)
    if is_float:
        data = ((rng % np.uint64(20001)).astype(np.float64) / 10000.0 - 1.0).astype("<f4").tobytes()
    elif bits == 8:
        data = (rng & np.uint64(0xFF)).astype(np.uint8).tobytes()
    elif bits == 16:
        data = ((rng & np.uint64(0xFFFF)).astype(np.int64) - 32768).astype("<i2").tobytes()
    elif bits == 24:
        v = (rng & np.uint64(0xFFFFFF)).astype(np.uint32)
        data = b"".join(int(x).to_bytes(3, "little") for x in v)
    else:
        data = ((rng & np.uint64(0xFFFFFFFF)).astype(np.int64) - (1 << 31)).astype("<i4").tobytes()
    path = str(tmp_path / ("a%d%s.wav" % (bits, "f" if is_float else "")))
    _write_wav(path, data, 2, 48000, bits, is_float)
    entries = []
    for be in (gpu_api, oracle):
        sb = be.SampleBank(48000)
        sb.add("s", path, "")
        entries.append(sb.get_sample(sb.get_index("s")))
    _same_bank_entry(*entries)
    peak = max(np.abs(entries[0][0]).max(), np.abs(entries[0][1]).max())
    assert 0.9999 < peak <= 1.0     # x * (1.0 / max) need not round to exactly 1.0


def test_silent_file_gives_nan_like_reference(gpu_api, oracle):
    z = np.zeros(64, np.float32)
    for be in (gpu_api, oracle):
        sb = be.SampleBank(48000)
        sb.add_decoded("z", z, 2, 48000, 16, "")
        l, r = sb.get_sample(0)
        assert np.isnan(l).all() and np.isnan(r).all()     # 0 * (1/0) (quirk noted at SURVEY a22)


def test_state_renders_project_from_files(gpu_api, oracle, tmp_path):
    """project.toml + project.lua + WAV / event files on disk -> td_state_* -> WAV on disk; the data chunk
    must equal the oracle's render of the same project."""
    p = W.drum_project(seconds=1.0)
    lua = p.to_lua(str(tmp_path / "assets"))
    (tmp_path / "project.toml").write_text('[project]\nname = "t"\n[settings]\nmain = "project.lua"\nbuffer_length = 1024\nproject_samplerate = 48000\n')
    (tmp_path / "project.lua").write_text(lua + 'set_output_file("%s")\n' % str(tmp_path / "out.wav"))
    s = gpu_api.State(open_dir=str(tmp_path))
    s.set_option("band_mode", 0)   # (the front-end's default is scan mode: this test compares bytes)
    assert s.refresh(), gpu_api.last_error()
    s.scan_exact()
    s.render()
    raw = open(str(tmp_path / "out.wav"), "rb").read()
    assert raw[:4] == b"RIFF" and raw[8:16] == b"WAVEfmt " and raw[36:40] == b"data"
    ch, sr, bits = struct.unpack("<H", raw[22:24])[0], struct.unpack("<I", raw[24:28])[0], struct.unpack("<H", raw[34:36])[0]
    assert (ch, sr, bits) == (2, 48000, 16)
    got = np.frombuffer(raw[44:], "<i2").reshape(-1, 2)
    want, _ = p.render(oracle, scan=True)
    assert np.array_equal(got, want)


def test_headless_driver(gpu_api, oracle, tmp_path):
    """python -m termdaw_amd <dir> --scan -o out.wav == refresh -> scan_exact -> render of the reference's TUI."""
    import subprocess
    import sys
    p = W.config1(seconds=0.5)
    lua = p.to_lua(str(tmp_path / "assets"))
    (tmp_path / "project.toml").write_text('[settings]\nmain = "main.lua"\nproject_samplerate = 48000\n')
    (tmp_path / "main.lua").write_text(lua)
    out = str(tmp_path / "x.wav")
    r = subprocess.run([sys.executable, "-m", "termdaw_amd", str(tmp_path), "--scan", "-o", out], capture_output=True, text=True,
                       cwd=str(__import__("pathlib").Path(__file__).resolve().parents[1]))
    assert r.returncode == 0, r.stderr + r.stdout
    got = np.frombuffer(open(out, "rb").read()[44:], "<i2").reshape(-1, 2)
    want, _ = p.render(oracle, scan=True)
    assert np.array_equal(got, want)


def test_state_render_view_matches_render_to_memory(gpu_api, tmp_path):
    p = W.config1(seconds=0.25)
    lua = p.to_lua(str(tmp_path / "a"))
    s = gpu_api.State("", 48000, 1024)
    assert s.refresh(lua), gpu_api.last_error()
    a = s.render_to_memory()
    v = s.render_view()
    # (two renders of one State: the second continues carried state exactly like the reference would)
    b = s.render_to_memory()
    s2 = gpu_api.State("", 48000, 1024)
    assert s2.refresh(lua)
    assert np.array_equal(s2.render_view(), a)
    assert v.shape == a.shape == b.shape and v.dtype == a.dtype and not v.flags.writeable


def test_headless_stream_mode(gpu_api, oracle, tmp_path):
    """python -m termdaw_amd <dir> --stream: text events on stdin -> declared streams -> block pulls -> WAV, against
    the oracle driven through the same stream_workflow.rs sequence."""
    import struct
    import types
    from termdaw_amd import __main__ as cli
    d = tmp_path / "proj"
    d.mkdir()
    W.write_wav_int16(str(d / "kick.wav"), W.kick_int16(12, 9000), 48000)
    (d / "project.toml").write_text('[settings]\nmain = "project.lua"\nbuffer_length = 512\nproject_samplerate = 48000\n')
    (d / "project.lua").write_text(
        'set_length(1.0); set_render_samplerate(48000); set_render_bitdepth(16); set_output_file("%s");\n'
        'load_sample("kick", "%s", "");\n'
        'declare_stream("live");\n'
        'add_sample_multi("hits", 0.9, 10.0, "kick", "live", -1);\n'
        'add_adsr("env", 1.0, 0.0, 1.0, "live", false, true, -1, { 0.01, 0.05, 0.7, 0.05, 0.2, 0.02 });\n'
        'add_sum("out", 1.0, 0.0);\n'
        'connect("hits", "env"); connect("env", "out"); set_output("out");\n' % (d / "stream.wav", d / "kick.wav"))
    packets = [[(0.010, 36.0, 0.9), (0.120, 36.0, 0.5)], [(0.260, 36.0, 0.8)], [(0.300, 40.0, 0.4), (0.410, 36.0, 1.0)]]
    lines = []
    for pk in packets:
        lines += ["live %r %r %r" % e for e in pk] + [""]
    lines.append("end 0.6")
    s = gpu_api.State(open_dir=str(d))
    rc = cli.stream(s, types.SimpleNamespace(realtime=False, output=None), lines=lines)
    assert rc == 0
    raw = (d / "stream.wav").read_bytes()
    assert struct.unpack("<I", raw[24:28])[0] == 48000
    got = np.frombuffer(raw[44:], np.int16).reshape(-1, 2)
    # the same sequence on the oracle
    sb = oracle.SampleBank(48000)
    sb.add_decoded("kick", W.kick_int16(12, 9000).astype(np.float32).reshape(-1), 2, 48000, 16, "")
    fb = oracle.FlowwBank(48000, 512)
    fb.declare_stream("live")
    g = oracle.Graph(512, 48000)
    g.add_sample_multi("hits", 0.9, 10.0, sb.get_index("kick"), fb.get_index("live"), -1)
    g.add_adsr("env", 1.0, 0.0, 1.0, fb.get_index("live"), False, True, -1, [0.01, 0.05, 0.7, 0.05, 0.2, 0.02])
    g.add_sum("out", 1.0, 0.0)
    assert g.connect("hits", "env") and g.connect("env", "out") and g.set_output("out")
    blocks = []

    def pull_until(t):
        while g.get_time() < int(t * 48000):
            fb.set_time(g.get_time())
            l, r = g.render(sb, fb)
            blocks.append(np.stack([l, r], axis=1))
            fb.set_time_to_next_block()
    latest = 0.0
    for pk in packets:
        fb.trim_streams()
        fb.append_stream("live", pk)
        fb.set_time(g.get_time())
        latest = max(latest, max(e[0] for e in pk))
        pull_until(latest)
    pull_until(max(latest, 0.6) + 512 / 48000.0)
    f = np.concatenate(blocks).astype(np.float32) * np.float32(32767.0)
    want = np.clip(np.trunc(f), -32768, 32767).astype(np.int16)
    assert got.shape == want.shape and np.array_equal(got, want)
    assert np.abs(got).max() > 1000
