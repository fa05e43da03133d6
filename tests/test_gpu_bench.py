"""bench.py as the driver invokes it (GPU box)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus_2_invoked_plainly_starts_its_own_ranks(gpu_api):
    """`python bench.py --gpus 2` without a launcher: the parent starts two ranks through torch.distributed.run before it
    touches the GPU itself, relays rank 0's ONE JSON line and exits 0.  On the 1-GPU box both ranks share device 0 and the
    peak table is exchanged over gloo (TD_BENCH_ONE_DEVICE / TD_BENCH_BACKEND: the self-test switches of bench.py)."""
    env = dict(os.environ, TD_BENCH_ONE_DEVICE="1", TD_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--no-extras",
                        "--no-cpu-baseline", "--seconds", "6"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["steps"] == 5
    assert out["scaling"] == "weak" and out["value"] > 0
    assert out["peak_table_entries"] == 2
    # what explains a scaling curve: every rank's own view of the timed region
    rk = out["ranks"]
    assert rk["n"] == 2
    for k in ("dt_ms_min", "dt_ms_max", "render_ms_min", "render_ms_max", "exchange_ms", "exchange_ms_min", "start_skew_us"):
        assert k in rk and rk[k] >= 0.0, (k, rk)
    assert rk["dt_ms_min"] <= rk["dt_ms_max"] and rk["render_ms_min"] <= rk["render_ms_max"] <= rk["dt_ms_max"]
    # ms_per_step x steps is the slowest rank's region (max over ranks), within rounding
    assert abs(out["ms_per_step"] * out["steps"] - rk["dt_ms_max"]) <= 0.02 * rk["dt_ms_max"] + 0.01
    assert rk["start_skew_us"] < 1e6


def test_bench_world_size_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0
