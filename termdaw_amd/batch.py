"""Multi-GPU batch rendering: independent projects shard across the GPUs of a node (BASELINE config 5).

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU
tests).  The render path has no data-path collective: project p is rendered entirely on GPU p mod G, the
projects of one GPU form ONE `api.Batch` (same-kind launches of different projects share a grid).  The only
exchange is the per-project peak table -- one all-reduce(max) of n_projects floats, each rank contributing its
own entries and zeros elsewhere.  The reference has no counterpart: it renders one
project per process (State::render, state.rs:477-577); this is the loop a batch driver would run it in.
"""
import numpy as np


def shard(n_projects, world, rank):
    """Project ids rendered by `rank`: round-robin p mod world == rank."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return list(range(rank, n_projects, world))


def build_shard(backend, make_project, ids, options=None):
    """Builds the projects `ids` (global project ids; make_project(pid) -> workloads.ProjectScript) on the current device
    and returns (batch, first_project).  `backend` is termdaw_amd.api; `options`: engine options applied to every
    graph ({"fuse_sources": 0, ...})."""
    batch = backend.Batch()
    first = None
    for pid in ids:
        p = make_project(pid)
        sb, fb, g = p.build(backend)
        for k, v in (options or {}).items():
            g.set_option(k, v)
        batch.add(sb, fb, g)
        if first is None:
            first = p
        else:
            p.assets.clear()   # (the samples live in HBM now)
    return batch, first


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


class PeakExchange:
    """The per-project peak table of the whole job, kept on the device: the engine writes this rank's entries straight
    into the tensor the collective runs on (td_batch_peak_table_device: own entries at rank + i * world, zeros
    elsewhere), then ONE all-reduce(max) -- RCCL on device memory, no host round trip.  `exchange()` returns when the
    table is complete in device memory; `host()` copies it out (the caller's report, not part of the exchange).  With a
    host-side backend (gloo, CPU tests of the N > 1 path) the collective itself needs the table on the host."""

    def __init__(self, batch, per_rank, rank, world, dist=None, on_device=True):
        import torch
        self.batch, self.rank, self.world, self.dist, self.on_device = batch, rank, world, dist, on_device
        self.n_total = per_rank * world
        self.table = torch.zeros(self.n_total, dtype=torch.float32, device="cuda")
        self._host = None
        torch.cuda.synchronize()

    def exchange(self):
        import torch
        self.batch.peak_table_device(self.table.data_ptr(), self.n_total, first=self.rank, stride=self.world)
        self.batch.sync()                 # the engine's stream has written this rank's entries
        self._host = None
        if self.dist is not None and self.dist.is_initialized():
            if self.on_device:
                self.dist.all_reduce(self.table, op=self.dist.ReduceOp.MAX)
                torch.cuda.synchronize()  # the reduced table stands in device memory
            else:
                t = self.table.cpu()
                self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
                self._host = t.numpy()

    def is_collective(self):
        """True when exchange() runs an all-reduce over more than one rank (it then also is a barrier between them)."""
        return self.dist is not None and self.dist.is_initialized() and self.dist.get_world_size() > 1

    def host(self):
        return self._host if self._host is not None else self.table.cpu().numpy()

    def __call__(self):
        self.exchange()
        return self.host()
