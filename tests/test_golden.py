"""Committed fixtures (tests/golden, made by tests/golden/make_golden.py): the oracle on CPU, the HIP engine
on the GPU box -- both must reproduce them byte for byte."""
import json
import os

import numpy as np
import pytest

from termdaw_amd import workloads as W

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")


def _cases():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(G, "make_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _check(backend):
    m = _cases()
    want = json.load(open(os.path.join(G, "digests.json")))
    pcm, _ = W.config1(seconds=0.25).render(backend)
    assert np.array_equal(pcm, np.load(os.path.join(G, "config1_0p25s.pcm.npy")))
    for name, mk in m.CASES.items():
        proj, scan = mk()
        pcm, _ = proj.render(backend, scan=scan)
        assert pcm.shape[0] == want[name]["frames"], name
        assert m.digest(pcm) == want[name]["sha256"], name


def test_oracle_reproduces_golden(oracle):
    _check(oracle)


@pytest.mark.gpu
def test_gpu_reproduces_golden(gpu_api):
    _check(gpu_api)
