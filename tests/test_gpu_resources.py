"""Device memory is handed back: sample-bank slabs, per-vertex event-table buffers, edge-buffer pools, graph and batch
arenas, PCM buffers.  Builds, renders (alone, in a batch, scanned, chunked, after refresh) and frees many projects and
compares hipMemGetInfo before and after."""
import ctypes as C
import gc

import numpy as np
import pytest

from termdaw_amd import workloads as W

pytestmark = pytest.mark.gpu


def _free_bytes():
    hip = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def _cycle(api, round_no):
    projects = [W.drum_project(seconds=0.5), W.synth_project(seconds=0.5), W.config2(seconds=0.5, n_src=6),
                W.config4(seconds=0.5, depth=9)]
    batch = api.Batch()
    built = []
    for p in projects:
        b = p.build(api)
        built.append(b)
        batch.add(*b)
    cs = projects[0].cs
    batch.render_all(cs, 16)
    batch.normalize_scan(cs)
    batch.render_all(cs, 24)
    sb, fb, g = built[round_no % len(built)]
    g.set_option("max_chunk_frames", 5000)
    g.render_all(sb, fb, cs, 16)
    if round_no % 2:
        del batch            # batch first ...
        del built, sb, fb, g
    else:
        del built, sb, fb, g
        del batch            # ... or the projects first (the batch must not keep dangling handles)
    s = api.State("", 48000, 1024)
    lua = W.config1(seconds=0.2)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        src = lua.to_lua(d)
        assert s.refresh(src), api.last_error()
        s.render_to_memory()
        assert s.refresh(src.replace('load_sample("kick"', 'load_sample("kick2"').replace('"kick")', '"kick2")')), api.last_error()
        s.render_to_memory()
    del s
    gc.collect()


def test_device_memory_is_returned(gpu_api):
    _cycle(gpu_api, 0)          # first use: code objects, runtime pools
    _cycle(gpu_api, 1)
    gc.collect()
    before = _free_bytes()
    for r in range(6):
        _cycle(gpu_api, r)
    gc.collect()
    after = _free_bytes()
    assert before - after < (8 << 20), "device memory not returned: %d bytes" % (before - after)
