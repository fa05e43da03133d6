"""The job's one collective behind the C ABI (include/termdaw_amd.h td_comm_* / td_batch_exchange_peaks; csrc/comm.cpp).

BASELINE config 5: independent projects over the GPUs of a node, "RCCL over xGMI only for the final peak all-reduce".  The
reference renders one project per process (State::render's loop, /root/reference/src/state.rs:563-575); a batch driver that runs
the loop on every GPU ends with this exchange.  Here, on the 1-GPU box: RCCL itself with one rank (ncclGetUniqueId ->
ncclCommInitRank -> ncclAllReduce(ncclMax, ncclFloat32) on the batch's stream), the host-callback kind with a second rank played
by the callback, and the error returns.  Two real ranks on one device go through the same entry over gloo:
tests/test_batch_gloo.py::test_two_rank_batch_with_the_hip_engine."""
import numpy as np
import pytest

from termdaw_amd import batch as tb
from termdaw_amd import workloads as W

pytestmark = pytest.mark.gpu


def _shard(api, per_rank, world, rank):
    mk = lambda pid: W.config2(seconds=0.5, n_src=6, seed_offset=64 * pid, base_len=3000)   # noqa: E731
    return tb.build_shard(api, mk, tb.shard(per_rank * world, world, rank), {"output_f32": 0})


def test_one_rank_over_rccl(gpu_api, oracle):
    """RCCL, dlopen'ed by the library, with a job of one rank: the collective runs on the engine's stream and leaves this rank's
    own peaks -- the oracle's normalization values -- in the table."""
    per_rank = 3
    b, first = _shard(gpu_api, per_rank, 1, 0)
    comm = gpu_api.Comm(gpu_api.comm_unique_id(), 0, 1)
    assert comm.backend() == "rccl-native" and "rccl" in gpu_api.comm_library()
    want = []
    for pid in range(per_rank):
        p = W.config2(seconds=0.5, n_src=6, seed_offset=64 * pid, base_len=3000)
        sb, fb, g = p.build(oracle)
        g.render_all(sb, fb, p.cs, 16, want_f32=False)
        want.append(np.float32(g.get_normalization_value("sum")))
    for _ in range(3):     # (the same communicator serves every step)
        b.rewind()
        b.render_all_async(first.cs, 16)
        b.exchange_peaks(comm, per_rank)          # enqueued behind the renders: no host wait in between
        b.sync()
        assert np.array_equal(b.peak_table(), np.array(want, np.float32))
    # a table wider than the batch: the other entries stay 0
    b.exchange_peaks(comm, per_rank + 2)
    t = b.peak_table()
    assert len(t) == per_rank + 2 and np.array_equal(t[:per_rank], np.array(want, np.float32)) and not t[per_rank:].any()


def test_the_host_callback_kind_with_a_second_rank_played_by_the_callback(gpu_api):
    """td_comm_init_host: the host's own all-reduce on the library's page-locked mirror.  Rank 1 of 2 here; "rank 0" is the
    callback, which contributes its own entries: the table in device memory afterwards is the element-wise maximum."""
    per_rank, world, rank = 2, 2, 1
    b, first = _shard(gpu_api, per_rank, world, rank)
    other = np.array([0.25, 0.0, 7.5, 0.0], np.float32)     # rank 0's entries sit at 0 + i * world
    seen = []

    def allreduce_max(table):
        seen.append(table.copy())
        np.maximum(table, other, out=table)
    comm = gpu_api.Comm.over_host(allreduce_max, rank, world)
    assert comm.backend() == "host-callback"
    b.rewind()
    b.render_all(first.cs, 16)
    mine = b.peaks()
    b.exchange_peaks(comm, per_rank)
    b.sync()
    t = b.peak_table()
    assert len(seen) == 1 and np.array_equal(seen[0], np.array([0.0, mine[0], 0.0, mine[1]], np.float32))
    assert np.array_equal(t, np.array([0.25, mine[0], 7.5, mine[1]], np.float32))
    # a callback that fails is an error return, not a crash

    def broken(table):
        raise RuntimeError("no network")
    bad = gpu_api.Comm.over_host(broken, rank, world)
    with pytest.raises(gpu_api.TermdawError):
        b.exchange_peaks(bad, per_rank)


def test_error_returns(gpu_api):
    b, first = _shard(gpu_api, 2, 1, 0)
    with pytest.raises(gpu_api.TermdawError):
        b.exchange_peaks(None, 1)                 # per_rank smaller than the batch
    with pytest.raises((gpu_api.TermdawError, ValueError)):
        gpu_api.Comm(b"short", 0, 1)
    with pytest.raises(gpu_api.TermdawError):
        gpu_api.Comm(gpu_api.comm_unique_id(), 3, 2)   # rank outside the job
    b.rewind()
    b.render_all(first.cs, 16)
    b.exchange_peaks(None, 2)                     # a job of one rank without a communicator: the table alone
    b.sync()
    assert np.array_equal(b.peak_table(), b.peaks())
