"""N > 1 path on CPU: two gloo ranks shard a batch of projects (p mod world) and exchange the per-project
peak table with one all-reduce(max).  The renderer stand-in is the CPU oracle (the HIP engine needs a GPU);
what is under test is termdaw_amd.batch -- the sharding and the exchange bench.py uses on RCCL."""
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
