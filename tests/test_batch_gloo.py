"""N > 1 path: two gloo ranks shard a batch of projects (p mod world) and exchange the per-project peak table
with one all-reduce(max).  Without a GPU the renderer stand-in is the CPU oracle and what is under test is
termdaw_amd.batch's sharding and exchange; with a GPU (-m gpu) both ranks render their shard with the HIP
engine on device 0 through the same api.Batch + PeakExchange objects bench.py uses over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from termdaw_amd import batch
from termdaw_amd import workloads as W

N_PROJECTS = 5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _peak_of(pid):
    from oracle import binding as oracle
    p = W.config2(seconds=0.1, n_src=4, seed_offset=64 * pid, base_len=3000)
    sb, fb, g = p.build(oracle)
    g.render_all(sb, fb, p.cs, 16, want_f32=False)
    return g.get_normalization_value("sum")


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = batch.shard(N_PROJECTS, world, rank)
    table = batch.exchange_peaks({pid: _peak_of(pid) for pid in mine}, N_PROJECTS, dist)
    out[rank] = (mine, table.tolist())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_is_a_partition():
    for world in (1, 2, 3, 8):
        got = sorted(p for r in range(world) for p in batch.shard(13, world, r))
        assert got == list(range(13))
    with pytest.raises(ValueError):
        batch.shard(4, 2, 2)


def test_two_rank_peak_exchange():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert out[0][0] == [0, 2, 4] and out[1][0] == [1, 3]
    want = [_peak_of(pid) for pid in range(N_PROJECTS)]
    assert out[0][1] == out[1][1] == [float(np.float32(x)) for x in want]
    assert all(x > 0 for x in want) and len(set(want)) == N_PROJECTS


def _hip_worker(rank, world, port, out):
    import torch   # noqa: F401  (before the engine: one HIP runtime per process)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from termdaw_amd import api
    api.set_device(0)
    per_rank = 3
    mk = lambda pid: W.config2(seconds=0.5, n_src=6, seed_offset=64 * pid, base_len=3000)   # noqa: E731
    b, first = batch.build_shard(api, mk, batch.shard(per_rank * world, world, rank), {"output_f32": 0})
    ex = batch.PeakExchange(b, per_rank, rank, world, dist, on_device=False)
    for _ in range(2):
        b.rewind()
        b.render_all(first.cs, 16)
        table = ex()
    import hashlib
    digests = [hashlib.sha256(b.read_pcm(i, first.cs).tobytes()).hexdigest() for i in range(per_rank)]
    out[rank] = (table.tolist(), digests)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_batch_with_the_hip_engine():
    import hashlib
    from oracle import binding as oracle
    world, per_rank = 2, 3
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_hip_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    want_peak, want_digest = [], []
    for pid in range(per_rank * world):
        p = W.config2(seconds=0.5, n_src=6, seed_offset=64 * pid, base_len=3000)
        sb, fb, g = p.build(oracle)
        pcm, _ = g.render_all(sb, fb, p.cs, 16, want_f32=False)
        want_peak.append(float(np.float32(g.get_normalization_value("sum"))))
        want_digest.append(hashlib.sha256(pcm.tobytes()).hexdigest())
    assert out[0][0] == out[1][0] == want_peak
    for rank in range(world):
        assert out[rank][1] == [want_digest[pid] for pid in batch.shard(per_rank * world, world, rank)]
