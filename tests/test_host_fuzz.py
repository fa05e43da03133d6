"""Sanitizer fuzz of the host-side readers (MIDI, WAV, project script): see tests/fuzz_host.cpp.  CPU only."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from termdaw_amd import workloads as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "termdaw_amd", "csrc")


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_host_readers_survive_mutated_inputs(tmp_path):
    exe = str(tmp_path / "fuzz_host")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-I", CSRC,
           "-I", os.path.join(ROOT, "include"), "-o", exe, os.path.join(ROOT, "tests", "fuzz_host.cpp"),
           os.path.join(CSRC, "midi.cpp"), os.path.join(CSRC, "wav.cpp"), os.path.join(CSRC, "lua_subset.cpp")]
    subprocess.check_call(cmd)
    data, _ = W.midi_bytes([(0.1 * i, 60 + i % 5, 0.5) for i in range(20)])
    (tmp_path / "seed.mid").write_bytes(data)
    W.write_wav_int16(str(tmp_path / "seed.wav"), W.kick_int16(3, 300), 48000)
    lua = W.drum_project(seconds=0.5).to_lua(str(tmp_path / "assets"))
    for word in ("load_sample", "load_midi_floww", "load_resource", "set_length", "set_render_samplerate", "set_render_bitdepth",
                 "set_output_file", "set_output", "connect"):
        lua = lua.replace(word + "(", "f(")
    lua = "\n".join("f(" + l.split("(", 1)[1] if l.startswith("add_") else l for l in lua.splitlines())
    lua += "\nlocal t = {1, 2, 3}\nfor i = 1, 3 do if t[i] > 1 then f(i) elseif i == 1 then f(2) end end\n" \
           "while false do end\nf(string.format('%d', 3), math.floor(2.5))\n"
    (tmp_path / "seed.lua").write_text(lua)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:allocator_may_return_null=0", UBSAN_OPTIONS="halt_on_error=1")
    out = subprocess.run([exe, "30000", str(tmp_path / "work.bin"), str(tmp_path / "seed.mid"), str(tmp_path / "seed.wav"),
                          str(tmp_path / "seed.lua")], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "fuzz done" in out.stdout
