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
    """The per-project peak table of the whole job, exchanged by the LIBRARY (td_batch_exchange_peaks, include/termdaw_amd.h): the
    engine writes this rank's entries into its own device table (own entries at rank + i * world, zeros elsewhere) and runs ONE
    ncclAllReduce(ncclMax, ncclFloat32) on the batch's stream right behind the renders -- RCCL on device memory, no host
    synchronisation in between, no torch in the data path.  `dist` (torch.distributed) only carries the 128-byte RCCL id from
    rank 0 to the others -- what a Rust host would do over a socket -- and, with a host-side backend (gloo: the CPU-hosted tests
    of the N > 1 path, two ranks on one GPU), serves as the host's own all-reduce behind td_comm_init_host.
    `exchange()` returns when the table is complete in device memory; `host()` copies it out (the report, not the exchange)."""

    def __init__(self, batch, per_rank, rank, world, dist=None, on_device=True, comm=None):
        """comm: an api.Comm of the same job to share (one communicator serves every batch of a process)."""
        from . import api
        self.batch, self.per_rank, self.rank, self.world, self.dist, self.on_device = batch, per_rank, rank, world, dist, on_device
        self.n_total = per_rank * world
        self.comm = None
        live = dist is not None and dist.is_initialized()
        self.fallback = None       # why the library's own RCCL communicator is not in use (None: it is, or none is needed)
        if comm is not None:
            self.comm = comm
        elif live and on_device:
            # rank 0 makes the id, torch.distributed carries its 128 bytes (a Rust host: a socket); every rank joins.  The ranks
            # then agree on the outcome -- an all-reduce(min) of "my td_comm_init worked" -- so that either ALL of them run the
            # library's collective or all of them say loudly that they do not (`fallback`, bench.py's exchange_backend) and run
            # the same reduction through torch.distributed on the library's device table instead of hanging in a half-made job.
            import torch
            err = ""
            try:
                box = [api.comm_unique_id() if rank == 0 else None]
            except api.TermdawError as e:
                box, err = [None], str(e)
            dist.broadcast_object_list(box, src=0)
            if box[0] is not None:
                try:
                    self.comm = api.Comm(box[0], rank, world)   # ncclCommInitRank: returns when every rank has joined
                except api.TermdawError as e:
                    err = str(e)
            ok = torch.tensor([1 if self.comm is not None else 0], dtype=torch.int32, device="cuda")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                self.comm = None
                self.fallback = "td_comm_init failed on a rank (%s): torch.distributed all_reduce on the library's table" % (err or "another rank")
        elif live:
            import torch

            def allreduce_max(table):                          # (np.float32 view of the library's page-locked mirror)
                t = torch.from_numpy(table)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            self.comm = api.Comm.over_host(allreduce_max, rank, world)

    def backend(self):
        if self.fallback:
            return "torch-rccl (FALLBACK: %s)" % self.fallback
        return self.comm.backend() if self.comm is not None else "none"

    def exchange(self):
        if self.fallback:   # (never on the test boxes: see __init__) the round-5 form: the engine fills torch's tensor, torch reduces it
            import torch
            if getattr(self, "_t", None) is None:
                self._t = torch.zeros(self.n_total, dtype=torch.float32, device="cuda")
            self.batch.peak_table_device(self._t.data_ptr(), self.n_total, first=self.rank, stride=self.world)
            self.batch.sync()
            self.dist.all_reduce(self._t, op=self.dist.ReduceOp.MAX)
            torch.cuda.synchronize()
            return
        self.batch.exchange_peaks(self.comm, self.per_rank)   # table kernel + collective, enqueued on the batch's stream
        self.batch.sync()                                      # the reduced table stands in device memory

    def is_collective(self):
        """True when exchange() runs an all-reduce over more than one rank (it then also is a barrier between them)."""
        return (self.comm is not None or self.fallback is not None) and self.world > 1

    def host(self):
        if self.fallback:
            return self._t.cpu().numpy()
        return self.batch.peak_table(self.n_total)

    def __call__(self):
        self.exchange()
        return self.host()
