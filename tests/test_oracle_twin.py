"""Cross-check of the C++ oracle against the independent numpy twin (tests/np_twin.py) -- two separately
written restatements of the reference must agree bit for bit on the integer/index class."""
import numpy as np
import pytest

import np_twin
from termdaw_amd import workloads as W


def _same(a, b):
    assert a[0].shape == b[0].shape
    assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
    assert np.array_equal(a[0], b[0])


@pytest.mark.parametrize("scan", [False, True])
def test_config1(oracle, scan):
    p = W.config1()
    _same(p.render(oracle, scan=scan), np_twin.render(p, scan=scan))


def test_config2_small(oracle):
    p = W.config2(seconds=1.0, n_src=64)
    _same(p.render(oracle), np_twin.render(p))


def _one_shot_project(kind, bl=1024):
    p = W.ProjectScript(48000, bl)
    p.set_length(0.6)
    p.assets["pluck"] = W.Asset(W.tone_int16(5, 3000))
    p.load_sample("pluck", "pluck", "mix-down" if kind == "lerp" else "")
    ev = [(0.01 + 0.05 * i, 60.0 + (i % 2), 0.2 + 0.1 * (i % 5)) for i in range(11)]
    ev.insert(3, ev[2][:1] + (72.0, 0.9))    # second on-event on the same frame: dropped by drum pulls
    ev.insert(6, (0.2, 60.0, 0.0))           # a note-off: ignored by drum pulls
    ev.sort(key=lambda e: e[0])
    p.event_files["ev"] = np.array(ev, np.float32)
    p.load_midi_floww("ev", "ev")
    if kind == "multi":
        p.add_sample_multi("v", 0.8, 25.0, "pluck", "ev", -1)
    elif kind == "multi60":
        p.add_sample_multi("v", 1.0, 0.0, "pluck", "ev", 60)
    else:
        p.add_sample_lerp("v", 1.3, -40.0, "pluck", "ev", -1, 300)
    p.add_bandpass("bp", 1.0, 0.0, 1.0, 500.0, 3000.0, kind != "multi")
    p.add_normalize("out", 0.9, 0.0)
    p.connect("v", "bp")
    p.connect("bp", "out")
    p.set_output("out")
    return p


@pytest.mark.parametrize("kind", ["multi", "multi60", "lerp"])
def test_one_shot_kinds_and_bandpass(oracle, kind):
    p = _one_shot_project(kind)
    _same(p.render(oracle), np_twin.render(p))
