#!/usr/bin/env python3
"""Regenerates the committed fixtures from the CPU oracle:  python tests/golden/make_golden.py

The reference ships no golden audio and cannot be built here, so these vectors pin the
*oracle's* output (itself pinned by the reference's ADSR tests, the hand KATs and the numpy twin); they
make every later change of oracle or engine visible.  Inputs are regenerated from integer-only seeded
generators (termdaw_amd/workloads.py), outputs are stored as data:
  config1_0p25s.pcm.npy   full int16 PCM of the README project, 0.25 s (12 blocks)
  digests.json            sha256 of the int16 PCM of the larger cases
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import binding as oracle   # noqa: E402
from termdaw_amd import workloads as W   # noqa: E402

CASES = {
    "config1_3s": lambda: (W.config1(), False),
    # (config 1's first block already holds the global peak, so its scanned render equals the un-scanned one
    # byte for byte and would pin nothing: the scanned cases are projects whose peak comes later)
    "config2_2s": lambda: (W.config2(seconds=2.0), False),
    "config2_2s_scanned": lambda: (W.config2(seconds=2.0), True),
    "config4_1s": lambda: (W.config4(seconds=1.0), False),
    "drum_project_4s": lambda: (W.drum_project(), False),
    "drum_project_4s_scanned": lambda: (W.drum_project(), True),
}


def digest(pcm):
    return hashlib.sha256(np.ascontiguousarray(pcm, dtype="<i2").tobytes()).hexdigest()


if __name__ == "__main__":
    p = W.config1(seconds=0.25)
    pcm, _ = p.render(oracle)
    np.save(os.path.join(HERE, "config1_0p25s.pcm.npy"), pcm)
    out = {}
    for name, mk in CASES.items():
        proj, scan = mk()
        pcm, _ = proj.render(oracle, scan=scan)
        out[name] = {"frames": int(pcm.shape[0]), "sha256": digest(pcm)}
    for name in out:   # a scanned fixture must differ from its un-scanned twin, or it pins nothing
        if name.endswith("_scanned"):
            assert out[name]["sha256"] != out[name[:-len("_scanned")]]["sha256"], name
    json.dump(out, open(os.path.join(HERE, "digests.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1))
