"""One-off: random project scripts (random calls, argument types, arities, table sizes) through td_state_refresh_source
on the CPU -- refresh may fail, the process must not crash."""
import sys, random
sys.path.insert(0, '.')
from termdaw_amd import api
fns = ["set_length", "set_render_samplerate", "set_render_bitdepth", "set_output_file", "load_midi_floww", "declare_stream",
       "add_sum", "add_normalize", "add_sample_multi", "add_sample_lerp", "add_debug_sine", "add_synth", "add_sampsyn", "add_lv2fx",
       "add_adsr", "add_bandpass", "connect", "set_output", "load_lv2", "parameter", "load_resource"]
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
def arg():
    k = rnd.randrange(9)
    if k == 0: return str(rnd.choice([0, 1, -1, 2**31, -2**31, 10**18, 48000, 16, 24]))
    if k == 1: return repr(rnd.choice([0.0, -0.0, 1e30, -1e30, 0.5, 1e-9, 90.5, -91.0, float(rnd.random())]))
    if k == 2: return '"%s"' % rnd.choice(["a", "b", "c", "", "out", "live", "x y", "nope"])
    if k == 3: return rnd.choice(["true", "false", "nil"])
    if k == 4: return "{" + ", ".join(repr(rnd.random() * rnd.choice([0, 1, -1, 100])) for _ in range(rnd.choice([0, 1, 5, 6, 7, 9, 10]))) + "}"
    if k == 5: return "0/0"
    if k == 6: return "1/0"
    if k == 7: return "math.floor(%r)" % (rnd.random() * 10)
    return '"v%d"' % rnd.randrange(4)
S, F, I, B, T = "s", "f", "i", "b", "t"
sigs = {"set_length": [F], "set_render_samplerate": [I], "set_render_bitdepth": [I], "set_output_file": [S], "declare_stream": [S],
        "add_sum": [S, F, F], "add_normalize": [S, F, F], "add_sample_multi": [S, F, F, S, S, I], "add_sample_lerp": [S, F, F, S, S, I, I],
        "add_debug_sine": [S, F, F, S], "add_synth": [S, F, F, S, F, F, T, F, F, T, F, T], "add_adsr": [S, F, F, F, S, B, B, I, T],
        "add_bandpass": [S, F, F, F, F, F, B], "connect": [S, S], "set_output": [S], "add_lv2fx": [S, F, F, F, S]}
names = ["v%d" % i for i in range(6)] + ["live", "nope"]
def good(kind):
    if kind == S: return '"%s"' % rnd.choice(names)
    if kind == F: return repr(rnd.choice([0.0, 1.0, 0.5, -0.3, 90.0, 200.0, 4000.0, 1e-5, float(rnd.random()) * 2]))
    if kind == I: return str(rnd.choice([-1, 0, 1, 16, 24, 32, 60, 400, 48000, 44100]))
    if kind == B: return rnd.choice(["true", "false"])
    return "{" + ", ".join(repr(round(rnd.random(), 3)) for _ in range(rnd.choice([0, 6, 6, 9, 9, 5]))) + "}"
ok = bad = 0
for it in range(4000):
    lines = ['declare_stream("live");']
    for _ in range(rnd.randrange(1, 16)):
        f = rnd.choice(list(sigs))
        args = [good(k) if rnd.random() < 0.93 else arg() for k in sigs[f]]
        if rnd.random() < 0.05: args = args[:-1]
        if f.startswith("add_") and f not in ("add_sum", "add_normalize", "add_bandpass", "add_lv2fx", "add_adsr"):
            pass
        lines.append("%s(%s);" % (f, ", ".join(args)))
    s = api.State("", rnd.choice([48000, 44100]), rnd.choice([1024, 256, 100]))
    r = s.refresh("\n".join(lines))
    ok += bool(r); bad += (not r)
print("refresh fuzz:", ok, "accepted,", bad, "rejected, no crash")
