"""The reference's own unit tests (src/adsr.rs:116-205: adsr_0 .. adsr_4, 51 assertions, tol 1e-3) replayed
against the oracle's restatement of apply_adsr / apply_ads / apply_r / apply_r_rt.  These are the only
golden vectors the reference ships for the render path; they pin adsr.rs:41-114."""
import ctypes as C

import numpy as np
import pytest

TOL = 0.001


def hit_conf(a, d, dv, s, sv, r):   # AdsrConf::hit_conf adsr.rs:15-30 via build_adsr_conf(6)
    return [a, d, dv, s, sv, r]


# (function, conf, args, expected) -- values from adsr.rs:120-204
C0 = hit_conf(1.0, 1.0, 0.5, 1.0, 0.25, 1.0)
C2 = hit_conf(1.0, 1.0, 0.5, 2.0, 0.25, 1.0)
DIP = [1.0, 1.0, 0.0, 0.5, 0.0, 0.5, 0.0, 1.0, 1.0]
VECTORS = (
    # adsr_0
    [("adsr", C0, (t,), e) for t, e in [(0.0, 0.0), (0.5, 0.5), (1.0, 1.0), (1.5, 0.75), (2.0, 0.5), (2.5, 0.375),
                                        (3.0, 0.25), (3.5, 0.125), (4.0, 0.0), (8.0, 0.0)]]
    # adsr_1
    + [("adsr", C0, (0.0,), 0.0)]
    + [("ads", C0, (t,), e) for t, e in [(0.5, 0.5), (1.0, 1.0), (1.5, 0.75), (2.0, 0.5), (2.5, 0.375), (3.0, 0.25), (7.0, 0.25)]]
    + [("r", C0, (t, 0.25), e) for t, e in [(0.0, 0.25), (0.5, 0.125), (1.0, 0.0), (9.0, 0.0)]]
    # adsr_2
    + [("adsr", C2, (0.0,), 0.0)]
    + [("ads", C2, (t,), e) for t, e in [(0.5, 0.5), (1.0, 1.0), (1.5, 0.75), (2.0, 0.5), (3.0, 0.375)]]
    + [("r", C2, (t, 0.375), e) for t, e in [(0.0, 0.375), (0.5, 0.1875), (1.0, 0.0), (9.0, 0.0)]]
    # adsr_3
    + [("adsr", C2, (0.0,), 0.0)]
    + [("ads", C2, (t,), e) for t, e in [(0.5, 0.5), (1.0, 1.0), (1.5, 0.75), (2.0, 0.5), (3.0, 0.375)]]
    + [("r_rt", C2, (t, 3.0), e) for t, e in [(0.0, 0.375), (0.5, 0.1875), (1.0, 0.0), (9.0, 0.0)]]
    # adsr_4
    + [("adsr", DIP, (t,), e) for t, e in [(0.0, 1.0), (0.5, 0.5), (1.0, 0.0), (1.5, 0.0), (2.0, 0.0), (2.5, 0.5),
                                           (3.0, 1.0), (4.0, 1.0), (8.0, 1.0)]]
)


def _conf9(lib, arr):
    a = np.asarray(arr, np.float32)
    out = np.zeros(9, np.float32)
    fp = C.POINTER(C.c_float)
    assert lib.orc_build_adsr_conf(a.ctypes.data_as(fp), a.size, out.ctypes.data_as(fp))
    return out


def test_vector_count():
    assert len(VECTORS) == 51


@pytest.mark.parametrize("i", range(len(VECTORS)))
def test_reference_adsr_vectors(oracle, i):
    lib = oracle.lib()
    fn, conf, args, expect = VECTORS[i]
    c = _conf9(lib, conf)
    p = c.ctypes.data_as(C.POINTER(C.c_float))
    f = {"adsr": lib.orc_apply_adsr, "ads": lib.orc_apply_ads, "r": lib.orc_apply_r, "r_rt": lib.orc_apply_r_rt}[fn]
    assert abs(expect - f(p, *args)) < TOL


def test_build_adsr_conf_lengths(oracle):
    lib = oracle.lib()
    fp = C.POINTER(C.c_float)
    out = np.zeros(9, np.float32)
    assert lib.orc_build_adsr_conf(None, 0, out.ctypes.data_as(fp)) and not out.any()   # adsr.rs:95-96 default
    six = _conf9(lib, [0.01, 0.1, 0.8, 5.0, 0.2, 0.5])
    assert list(six) == [np.float32(x) for x in [0.0, 0.01, 1.0, 0.1, 0.8, 5.0, 0.2, 0.5, 0.0]]   # hit form
    bad = np.zeros(5, np.float32)
    assert not lib.orc_build_adsr_conf(bad.ctypes.data_as(fp), 5, out.ctypes.data_as(fp))          # adsr.rs:111-113
