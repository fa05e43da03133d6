"""The per-process memory cache (include/termdaw_amd.h td_trim_memory / td_cached_memory_bytes; csrc/devmem.cpp): blocks of
freed handles are handed out again -- with whatever the last owner left in them -- and a project set up in such blocks renders
the oracle's bytes; trimming gives everything back.  (Why the cache exists: DESIGN.md 7 "One process of 40".)"""
import gc

import numpy as np
import pytest

from termdaw_amd import workloads as W

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_blocks_of_freed_handles_are_used_again_and_trimmed(gpu_api, oracle):
    lib = gpu_api.lib()
    p = W.drum_project(seconds=1.5)
    ref_pcm, ref_f = p.render(oracle)
    built = p.build(gpu_api)
    pcm, f = p.render(gpu_api, built=built)
    assert np.array_equal(_bits(f), _bits(ref_f)) and np.array_equal(pcm, ref_pcm)
    del built
    gc.collect()
    held = lib.td_cached_memory_bytes()
    assert held > 0
    # a different project first (its tables land in the drum project's old blocks), then the same one again
    q = W.config1(seconds=1.0)
    q_pcm, q_f = q.render(oracle)
    g_pcm, g_f = q.render(gpu_api)
    assert np.array_equal(_bits(g_f), _bits(q_f)) and np.array_equal(g_pcm, q_pcm)
    gc.collect()
    for _ in range(3):
        pcm, f = p.render(gpu_api)
        assert np.array_equal(_bits(f), _bits(ref_f)) and np.array_equal(pcm, ref_pcm)
        gc.collect()
    assert lib.td_cached_memory_bytes() <= 4 * held + (64 << 20)   # (set-up and tear-down go round in the same blocks)
    lib.td_trim_memory()
    assert lib.td_cached_memory_bytes() == 0
    pcm, f = p.render(gpu_api)
    assert np.array_equal(_bits(f), _bits(ref_f)) and np.array_equal(pcm, ref_pcm)
