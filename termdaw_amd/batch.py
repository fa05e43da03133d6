"""Multi-GPU batch rendering: independent projects shard across the GPUs of a node (BASELINE config 5).

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU
tests).  The render path has no data-path collective: project p is rendered entirely on GPU p mod G.
The only exchange is the per-project peak table -- one all-reduce(max) of n_projects floats, each rank
contributing its own entries and zeros elsewhere.
"""
import numpy as np


def shard(n_projects, world, rank):
    """Project ids rendered by `rank`: round-robin p mod world == rank."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return list(range(rank, n_projects, world))


def exchange_peaks(local_peaks, n_projects, dist=None, device="cpu", buf=None):
    """local_peaks: {project_id: peak >= 0}.  Returns the full table (np.float32[n_projects]) after one
    all-reduce(max); with dist None (single process) it is just the local table -- no device round trip.
    buf: optional preallocated float32 tensor of n_projects elements on `device` (reused across calls)."""
    table = np.zeros(n_projects, dtype=np.float32)
    for pid, pk in local_peaks.items():
        if not (0 <= pid < n_projects):
            raise IndexError("project id %d outside the table" % pid)
        table[pid] = float(pk)
    if dist is None or not dist.is_initialized():
        return table
    import torch
    t = buf if buf is not None else torch.empty(n_projects, dtype=torch.float32, device=device)
    t.copy_(torch.from_numpy(table))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.cpu().numpy()
