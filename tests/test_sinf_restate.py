"""kernels.hip sin_glibc (engine option "sine_mode" 1: the oscillators of debug_sine / synth carry glibc's sinf bit for bit,
/root/reference/src/extensions.rs:450,501 `f32::sin`) is a restatement of glibc's published double-precision algorithm.
tools/sinf_restate.c is the same sequence of operations on the host: here it is compared with the host's own sinf over every
finite float.  (The device side of it: tests/test_gpu_sine_exact.py.)"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cpu_has_fma():
    try:
        return " fma " in open("/proc/cpuinfo").read()
    except OSError:
        return False


@pytest.mark.skipif(shutil.which("gcc") is None or not _cpu_has_fma(), reason="needs gcc and a CPU with FMA (glibc's sinf variant there)")
def test_the_restated_sinf_is_the_hosts_sinf_on_every_finite_float(tmp_path):
    exe = str(tmp_path / "sinf_restate")
    subprocess.check_call(["gcc", "-O2", "-mfma", "-ffp-contract=off", os.path.join(ROOT, "tools", "sinf_restate.c"), "-o", exe, "-lm", "-lpthread"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600).stdout
    assert "all finite floats: 0 differ from glibc sinf" in out, out
