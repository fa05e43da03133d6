"""CPU-side tests of the product's host logic: the C-ABI library loads and exports every symbol the
header declares, graph construction rules, the project front-end (Lua subset + State::refresh), and the
loud failure of every render entry point without a GPU (there is no CPU fallback)."""
import os
import re

import numpy as np
import pytest

from termdaw_amd import workloads as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(api):
    hdr = open(os.path.join(ROOT, "include", "termdaw_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(td_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 60
    L = api.lib()
    missing = [n for n in sorted(declared) if not hasattr(L, n)]
    assert not missing, missing
    assert declared == set(api.SIGNATURES), declared ^ set(api.SIGNATURES)


def test_graph_rules_match_reference(api):   # graph.rs:58-174
    g = api.Graph(8, 48000)
    g.add_sum("a", 1, 0)
    g.add_sum("b", 1, 0)
    g.add_sampleloop("src", 1, 0, 0)
    assert not g.check_graph()
    assert g.connect("src", "a") and g.connect("a", "b")
    assert not g.connect("b", "a")
    assert not g.connect("a", "a")
    assert not g.connect("a", "src")
    assert not g.connect("ghost", "a") and not g.connect("a", "ghost")
    assert not g.set_output("ghost") and g.set_output("b")
    assert g.check_graph()
    g2 = api.Graph(8, 48000)
    g2.add_sum("lonely", 1, 0)
    g2.set_output("lonely")
    assert not g2.check_graph()
    with pytest.raises(api.TermdawError):
        g2.add_adsr("e", 1, 0, 1, 0, False, True, -1, [1, 2, 3])      # state.rs:444: 0, 6 or 9 floats
    g2.add_normalize("n", 1, 0)
    assert g2.get_normalization_value("n") == 0.0                      # extensions.rs:87-92
    g2.reset_normalize_vertices()
    assert g2.get_normalization_value("n") == np.float32(0.000001)     # extensions.rs:295-299
    assert g2.get_normalization_value("lonely") == -1.0
    g2.set_time(4096)
    assert g2.get_time() == 4096 and g2.change_time(5000, False) == 0 and g2.change_time(100, True) == 100


def test_flowwbank_cursor(api):
    fb = api.FlowwBank(100, 16)
    assert fb.add_events("f", [(0.05, 60, 0.5), (0.2, 61, 0.5)]) == 0
    assert fb.declare_stream("s") == 1
    assert fb.get_index("f") == 0 and fb.get_index("s") == 1 and fb.get_index("x") is None


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="only meaningful on a box without a GPU")
def test_no_cpu_fallback(api):
    assert api.device_count() == 0
    sb = api.SampleBank(48000)
    with pytest.raises(api.TermdawError, match="no HIP device"):
        sb.add_decoded("a", np.zeros(8, np.float32), 2, 48000, 16, "")
    g = api.Graph(8, 48000)
    g.add_debug_sine("s", 1, 0, 0)
    g.set_output("s")
    fb = api.FlowwBank(48000, 8)
    with pytest.raises(api.TermdawError, match="no HIP device"):
        g.render_all(sb, fb, 2)
    with pytest.raises(api.TermdawError, match="no HIP device"):
        g.render(sb, fb)


def test_lua_front_end_records_like_the_script_recorder(api, tmp_path):
    """ProjectScript.to_lua() -> td_state_refresh_source must record exactly the calls the recorder made."""
    for p in (W.config3(seconds=2.0), W.synth_project(seconds=1.0)):
        src = p.to_lua(str(tmp_path))
        s = api.State("", 48000, 1024)
        assert s.refresh(src), api.last_error()
        dump = s.dump_calls().strip().split("\n")
        assert len(dump) == len(p.script_order)
        for line, (fn, args) in zip(dump, p.script_order):
            assert line.startswith(fn + "(")
            if fn.startswith("add_") or fn == "connect":
                assert '"%s"' % args[0] in line
        assert s.cs == p.cs and s.bd == p.bd and s.render_sr == p.render_sr
        g = s.g
        assert api.lib().td_graph_vertex_count(g.h) == sum(len(p.calls[k]) for k in p.calls if k.startswith("add_"))


def test_lua_subset_semantics(api):
    s = api.State("", 48000, 1024)
    src = '''
    -- comment
    --[[ block
         comment ]]
    local n = 3
    base = "v"
    for k = 0, n - 1 do
      add_sum(base .. k, 0.5 + k / 64, -90 + 180 * k / 63);
    end
    for i, name in ipairs({"x", "y"}) do add_sum(name .. i, 1, 0) end
    add_sum(string.format("s%02d", 7), 1.0, 0.0)
    local t = { 1, 2.5, x = 3 }
    if #t == 2 and t.x == 3 and not (t[1] ~= 1) then add_normalize("n", 1.0, 0.0) elseif true then add_sum("bad", 1, 0) end
    local i = 0
    while i < 2 do i = i + 1 end
    add_bandpass("bp" .. i, 1, 0, 1, 1000, 0, true)
    adsr = { 0.01, 0.1, 0.8, 0.1, 0.2, 0.01 }
    connect("v0", "n") connect("v1", "n")
    set_length(3.0); set_render_bitdepth(24) set_render_samplerate(96000)
    set_output("n")
    set_output_file('out.wav')
    '''
    assert s.refresh(src), api.last_error()
    d = s.dump_calls()
    assert 'add_sum("v0",0.5,-90)' in d and 'add_sum("v2",0.53125,' in d
    assert 'add_sum("x1",1,0)' in d and 'add_sum("y2",1,0)' in d and 'add_sum("s07",1,0)' in d
    assert 'add_normalize("n",1,0)' in d and "bad" not in d
    assert 'add_bandpass("bp2",1,0,1,1000,0,true)' in d
    assert s.cs == 141 and s.bd == 24 and s.render_sr == 96000 and s.output_file == "out.wav"


@pytest.mark.parametrize("src,needle", [
    ("add_sum('a', 1)", "error converting Lua nil to f32"),
    ("add_sum('a', {}, 0)", "error converting Lua table to f32"),
    ("add_sample_multi('m', 1, 0, 's', 'f', 1.5)", "to i32"),
    ("undefined_fn(1)", "attempt to call a nil value"),
    ("x = = 3", "unexpected symbol"),
    ("function f() end", "not supported"),
    ("add_sum('a', 1, 0) set_output('nope')", "graph check failed"),
    ("add_sampleloop('l', 1, 0, 'missing') set_output('l')", "Could not get sample index"),
    ("add_debug_sine('l', 1, 0, 'missing') set_output('l')", "Could not get floww index"),
    ("declare_stream('f') add_sampsyn('w', 1, 0, 'f', {}, 'tab') set_output('w')", "Could not find resource named tab"),
    ("load_midi_floww('f', '/nonexistent/x.mid')", "Could not read midi file"),
])
def test_refresh_failures_are_reported(api, src, needle):
    s = api.State("", 48000, 1024)
    assert not s.refresh(src)
    assert needle in api.last_error(), api.last_error()
    with pytest.raises(api.TermdawError, match="not loaded"):
        s.render("/tmp/never.wav")


def test_refresh_takes_output_settings(api):
    """State::refresh std::mem::take()s output_file / output_vertex into locals before the script runs and stores
    them back only after it ran (state.rs:79-80, 169-170): a script without set_output keeps the previous vertex,
    but a script that FAILS leaves both fields empty -- the next script must set them again or fail check_graph."""
    s = api.State("", 48000, 1024)
    one = 'set_length(1.0) add_sum("a", 1, 0) declare_stream("f") add_debug_sine("s", 1, 0, "f") connect("s", "a")\n'
    assert s.refresh(one + 'set_output("a") set_output_file("x.wav")'), api.last_error()
    assert s.output_file == "x.wav"
    assert s.refresh(one), api.last_error()            # no set_output: the taken value is put back
    assert s.output_file == "x.wav"
    assert not s.refresh("x = = 3")                    # a failing script: taken, never put back
    assert s.output_file == ""
    assert not s.refresh(one)                          # ... so the old output vertex is gone too
    assert "graph check failed" in api.last_error(), api.last_error()
    assert s.refresh(one + 'set_output("a")'), api.last_error()
    assert s.output_file == ""


def test_reference_example_scripts_parse(api):
    """The reference's own example scripts run through the front-end up to the first missing asset file
    (their /home/cody/... paths do not exist anywhere but the author's machine)."""
    ref = "/root/reference"
    if not os.path.isdir(ref):
        pytest.skip("reference tree not present on this box")
    for f in ("project.lua", "examples/neg-adsr-env-example.lua", "examples/sample-project.lua",
              "examples/sample-synth-adsr-lv2fx-example.lua", "examples/stream.lua"):
        s = api.State("", 48000, 1024)
        assert not s.refresh(open(os.path.join(ref, f)).read())
        assert "could not open file" in api.last_error()
        assert "set_output(" in s.dump_calls() and "connect(" in s.dump_calls()


def test_project_toml(api, tmp_path):
    (tmp_path / "project.toml").write_text('[project]\nname = "x"\n\n[settings]\n# main = "other.lua"\nmain = "p.lua"\nbuffer_length = 512\n')
    (tmp_path / "p.lua").write_text('set_length(1.0)\nadd_sum("a", 1, 0)\ndeclare_stream("f")\nadd_debug_sine("s", 1, 0, "f")\nconnect("s", "a")\nset_output("a")\n')
    s = api.State(open_dir=str(tmp_path))
    assert s.refresh(), api.last_error()
    assert s.cs == int(np.ceil(np.float32(44100) * np.float32(1.0) / np.float32(512)))   # psr defaults to 44100 (config.rs:62-64)


def test_c_abi_from_plain_c(api, tmp_path):
    """include/termdaw_amd.h compiled as C (gcc -std=c99 -pedantic) and linked against the library."""
    import subprocess
    exe = str(tmp_path / "c_abi_smoke")
    libdir = os.path.dirname(api.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_abi_smoke.c"), "-o", exe, "-L", libdir, "-ltermdaw_amd",
                           "-Wl,-rpath," + libdir])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "c abi ok" in out.stdout, out.stdout + out.stderr


def test_bench_helpers_format_in_both_normalize_forms():
    """bench.py's pure helpers (no GPU needed to import the module): the roofline note formats for both Normalize forms, the
    byte model follows the engine mode."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("td_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for single in (False, True):
        note = mod.k_sum_roofline_note(single, 20.3, 2880512, 1)
        assert ("(4k+4)" in note) == single and "20 MB" in note
    assert mod.algorithmic_bytes_per_frame(64, True, True)["k_sum"] == 4.0 * 64 + 8.0
    assert mod.algorithmic_bytes_per_frame(64, False, False)["k_sum"] == 8.0 * 64 + 8.0


def test_bench_line_compact_form_fits_the_drivers_tail():
    """bench.py prints the compact form of its line: every headline number, no prose -- the long form of a committed run
    compacts to well under the 6 KB the driver's stored tail holds, with the derived-mode entries ahead of `configs`."""
    import json
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_full.json")))
    full["config"]["normalize_id"] = "one launch"
    full["config"]["parallelism_id"] = "project p on rank p mod N; one RCCL all-reduce(max) of the 1-entry peak table"
    c = bench.compact(full)
    line = json.dumps(c)
    assert len(line) <= 6000, len(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in c, k
    # (`bound` names what `peak` is: "hbm" / "mfma", or the cache level whose gather rate it is for a launch the caches serve)
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(c["roofline"])
    assert c["roofline"]["bound"] in ("hbm", "mfma") or "gather" in c["roofline"]["bound"] or "l2 stream" in c["roofline"]["bound"]
    assert "model" not in c["config"] and "workload" in c["config"]
    keys = list(c)
    assert keys.index("edge_buffer_mode") < keys.index("configs") and keys.index("config5") < keys.index("configs")
    assert keys.index("scanned") < keys.index("configs")
    assert os.path.exists(os.path.join(ROOT, "profiles", "NOTES.md"))


def test_the_complete_rust_binding_follows_the_header():
    """docs/amd_ffi.rs (the full `extern "C"` block INTEGRATION.md points to) is what tools/gen_rust_extern.py makes of
    include/termdaw_amd.h today, and names every entry point the header declares."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_rust_extern.py")], capture_output=True, text=True, check=True).stdout
    assert out == open(os.path.join(root, "docs", "amd_ffi.rs")).read()
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(root, "include", "termdaw_amd.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(td_[a-z0-9_]+)\s*\(", header))
    bound = set(re.findall(r"pub fn (td_[a-z0-9_]+)\(", out))
    assert declared == bound and len(bound) > 90


def test_headline_block_is_the_generated_one():
    """README.md and DESIGN.md quote ONE end-of-round headline: the block tools/headline.py makes of the committed driver-form
    bench line (profiles/r06_bench_k20.json).  A hand-edited or stale figure fails here."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("headline", os.path.join(ROOT, "tools", "headline.py"))
    h = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(h)
    if not os.path.exists(h.SRC):
        pytest.skip("no committed r06 driver-form bench line yet")
    want = h.block()
    for name in ("README.md", "DESIGN.md"):
        assert h.current(os.path.join(ROOT, name)) == want, name


def test_the_two_default_sets_differ_in_two_keys(api):
    """include/termdaw_amd.h: seven supported engine options; a bare td_graph (what a Rust host binds: the reference's bytes) and a
    State's graph (the front-end: the stated tolerance, checked per render) differ in exactly band_mode (0 | 2) and sine_mode
    (1 | 2) -- every other key, the test hooks under "debug." included, has ONE default."""
    import ctypes as C
    keys = []
    while True:
        k = api.lib().td_graph_option_key(len(keys))
        if not k:
            break
        keys.append(k.decode())
    public = [k for k in keys if not k.startswith("debug.")]
    assert sorted(public) == sorted(["fuse_sources", "packed_samples", "band_mode", "band_guard_ppb", "sine_mode", "output_f32", "max_chunk_frames"])
    assert len(keys) <= 32 and len(set(keys)) == len(keys)
    g = api.Graph(1024, 48000)
    s = api.State("", 48000, 1024)
    sg = api.Graph.__new__(api.Graph)
    sg.h = api.lib().td_state_graph(s.h)
    diff = {}
    for k in keys:
        a, b = g.get_option(k), api.Graph.get_option(sg, k)
        if a != b:
            diff[k] = (a, b)
    sg.h = None   # (the State owns it)
    assert diff == {"band_mode": (0, 2), "sine_mode": (1, 2)}, diff
    with pytest.raises(api.TermdawError):
        g.set_option("branch_streams", 1)      # gone in round 6
    with pytest.raises(api.TermdawError):
        g.get_option("no_such_key")
