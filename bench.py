#!/usr/bin/env python3
"""bench.py -- offline render throughput of the HIP engine on BASELINE config 2 (+ the other configs beside it).

A "step" is one complete fresh render of every project of this GPU's batch (default: ONE project = BASELINE
config 2: 64 sampleloop vertices -> one normalize, 60 s @ 48 kHz, bl 1024 = 2,880,512 stereo frames) from sample
PCM resident in HBM to the 16-bit PCM buffer in HBM: reset normalize vertices, rewind the FlowwBank, compile +
launch every vertex kernel.  `--projects-per-gpu P` puts P such projects (seeds offset by 64 per project id, BASELINE
config 5's per-GPU share at P = 64) into the batch; same-kind launches of different projects share one grid.  At
N > 1 project p of the job lives on rank p mod N (no data-path collective) and the ranks exchange only the
per-project peak table with ONE RCCL all-reduce(max) inside the timed region.

    python bench.py --gpus N --steps K --warmup W [--projects-per-gpu P]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  Besides the contract fields:
  roofline      the dominant kernel (by HIP-event time measured live on the engine's stream during the timed steps)
                against a ceiling MEASURED in this process on this device (tools/ubench/ceilings.hip)
  rooflines     the same for every kernel family of the timed region
  cpu_baseline  the CPU oracle (oracle/, a C++ restatement of the reference's block-serial algorithm -- NOT the
                Rust reference, which cannot be built here) single-threaded on the same project
  config5       (default run) 64 projects per GPU through the same batch path, aggregate Msamples/s
  scanned       (N = 1) the reference's normalize-then-render workflow on the same project
  configs       (N = 1) BASELINE configs 1, 3, 4: ms per render, dominant kernel, its bound and fraction
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_GBS = 6290.0      # ... 6.29 TB/s measured float4 copy
L2_PEAK_GBS = 34500.0      # ... "L2 (per XCD)": 4 MiB per XCD, ~34.5 TB/s aggregate
L2_GATHER_GBS = 17800.0    # ... "Indexed rows: gather": rows every workgroup shares (the XCD's L2), 16.8-18.8 TB/s chip-wide
MALL_GATHER_GBS = 8600.0   # ... the same from a 38 MB table (Infinity Cache): 8.6 TB/s
SIMDS = 256 * 4            # CUs x SIMDs
CLOCK_HZ = 2.4e9           # max clock (guide); a wave64 VALU instruction occupies its SIMD-32 for 2 cycles

SECONDS = 60.0
N_SRC = 64
PROF_EVERY = 8
PREWARM_S = 0.25           # untimed, before the W warm-up steps: steady device clocks (see time_batch)
PROFILE_TAG = "r06"        # profiles/<tag>_*: the rocprofv3 passes `traffic_profiled` / `valu_profiled` come from
PROFILE_TAG_PREV = "r05"   # ... or, until this round's passes are committed, the previous round's (the entry says which file)


def algorithmic_bytes_per_frame(k, fused, packed):
    """Bytes each kernel has to move per output frame.  Edge-buffer mode is SURVEY.md
    8(d)'s model: an 8-byte stereo f32 per edge read and per vertex write.  With source inlining the
    sample_loop vertices have no edge buffers: k_sum gathers each source's sample frame itself -- 4 B in the
    packed 16-bit form, 8 B as f32 -- and writes the raw sum once."""
    per_src = (4.0 if packed else 8.0) if fused else 8.0
    return {
        "k_sample_loop": 8.0 * k,         # per source: the 8 B edge write (the sample reads loop over <= 0.9 MB and are cache hits:
                                          # PMC shows 0.31 GB fetched against 1.47 GB written per launch), k sources per launch
        "k_sum": per_src * k + 8.0,       # Normalize pass A: k edge (or inlined sample) reads + raw sum write
        "k_scale": 8.0 + 4.0,             # Normalize pass B with fused int16 quantise: raw sum in, PCM out (no f32 write-back)
    }


def k_sum_roofline_note(single_pass, sample_mb, frames, projects):
    """The text beside the dominant kernel's roofline entry (both Normalize forms; tests/test_host_logic.py formats both)."""
    return ("source inlining: k_sum's algorithmic bytes are its own gathers, %s -- k looping samples "
            "read in place in their packed 16-bit form + %s.  The gathers re-read a %.0f MB "
            "sample set ~37x per launch, so they are cache hits, not HBM traffic: `peak` is the rate at which "
            "tools/ubench/ceilings.hip performs the SAME gathers and an 8 B/frame write with the arithmetic "
            "removed (best of three issue depths, same lengths, same grid, this process, this device)%s; "
            "`hbm_compulsory_frac` prices the bytes that must cross HBM once (packed samples + the output write) "
            "against 8 TB/s") % (
                "(4k+4) B/frame" if single_pass else "(4k+8) B/frame",
                ("one int16 PCM write (the launch also finds the running peak through in-launch "
                 "granules, scales and quantises: single_pass_normalize)") if single_pass else "one raw-sum write",
                sample_mb,
                (" -- frac = ceiling_ms / avg_ms; the ceiling kernel writes 4 B/frame more than this launch "
                 "(%.1f us at 8 TB/s) and does none of its peak hand-off" % (4.0 * frames * projects / 8e12 * 1e6)) if single_pass else "")


def ubench():
    """tools/ubench/libtd_ubench.so (built by __graft_entry__.build(); rebuilt here if missing)."""
    d = os.path.join(ROOT, "tools", "ubench")
    so = os.path.join(d, "libtd_ubench.so")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(d, "ceilings.hip")):
        subprocess.check_call(["make", "-C", d, "-s"])
    L = ctypes.CDLL(so)
    L.td_ubench_gather.restype = ctypes.c_float
    L.td_ubench_gather.argtypes = [ctypes.POINTER(ctypes.c_uint32), ctypes.c_int, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.td_ubench_stream.restype = ctypes.c_float
    L.td_ubench_stream.argtypes = [ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.td_ubench_valu_chain_ns.restype = ctypes.c_float
    L.td_ubench_valu_chain_ns.argtypes = []
    L.td_ubench_fma_issue_ns.restype = ctypes.c_float
    L.td_ubench_fma_issue_ns.argtypes = [ctypes.c_int]
    return L


def cpu_model():
    """The host CPU's model string (/proc/cpuinfo), or "unknown"."""
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(project, frames, runs=5):
    """Oracle ("port" of the reference algorithm) on ONE host core, the process pinned to it for the measurement
    (`taskset -c <core>`): whole config-2 project per run, median."""
    from oracle import binding as oracle
    pinned = None
    before = None
    try:
        before = os.sched_getaffinity(0)
        try:
            pinned = int(ctypes.CDLL(None).sched_getcpu())   # the core this process is on now (core 0 also serves the box's interrupts)
        except Exception:   # noqa: BLE001
            pinned = -1
        if pinned not in before:
            pinned = min(before)
        os.sched_setaffinity(0, {pinned})
    except (AttributeError, OSError):
        before = None
    times = []
    try:
        for r in range(runs + 1):
            built = project.build(oracle)
            t0 = time.perf_counter()
            project.render(oracle, built=built, want_f32=False)
            dt = time.perf_counter() - t0
            if r:
                times.append(dt)
    finally:
        if before is not None:
            try:
                os.sched_setaffinity(0, before)
            except OSError:
                pass
    med = float(np.median(times))
    return {
        "value": round(frames / med / 1e6, 4),
        "unit": "Msamples/s",
        "cores": 1,
        "kind": "port",
        "sample": "full workload (%d frames) x %d runs, median" % (frames, runs),
        "pinned_to_core": pinned,
        "cpu": cpu_model(),
        "host_cores": os.cpu_count(),
        "seconds_per_render": round(med, 4),
    }


def cpu_worker(seed_offset, seconds):
    """One config-5 project (config 2, seeds offset) on this process' core; prints its render time."""
    from termdaw_amd import workloads
    from oracle import binding as oracle
    p = workloads.config2(seconds=seconds, n_src=N_SRC, seed_offset=seed_offset)
    built = p.build(oracle)
    p.render(oracle, built=built, want_f32=False)        # warm
    built = p.build(oracle)
    t0 = time.perf_counter()
    p.render(oracle, built=built, want_f32=False)
    print("CPU_WORKER %.6f %d" % (time.perf_counter() - t0, p.cs * p.bl), flush=True)


def cpu_all_cores(seconds, workers):
    """Context for config 5: one project per host core, all at once, each through the single-threaded
    oracle in its own process (started before anything here touches the GPU runtime in THAT process)."""
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(64 * i), "--seconds", str(seconds)],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for i in range(workers)]
    rate = 0.0
    done = 0
    slowest = 0.0
    deadline = time.time() + 150.0   # a box that cannot run them side by side is not worth waiting for
    for p in procs:
        try:
            out, _ = p.communicate(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            for q in procs:
                if q.poll() is None:
                    q.kill()
            return None
        for line in out.splitlines():
            if line.startswith("CPU_WORKER"):
                _, sec, fr = line.split()
                rate += float(fr) / float(sec)
                slowest = max(slowest, float(sec))
                done += 1
    if not done:
        return None
    return {"value": round(rate / 1e6, 2), "unit": "Msamples/s", "cores": done, "kind": "port",
            "sample": "%d processes x 1 project each (config 2, seeds offset by 64 per process), all concurrently; sum of the "
                      "per-process rates; slowest render %.3f s" % (done, slowest)}


def launch_ranks(n):
    """`python bench.py --gpus N ...` without a launcher: run the same command line under torch.distributed.run
    (--nproc-per-node N on 127.0.0.1, a free port), print the ONE JSON line rank 0 produced, return the exit code."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for out in proc.stdout:
        out = out.strip()
        if out.startswith("{") and out.endswith("}"):
            try:
                json.loads(out)
                line = out
                continue
            except ValueError:
                pass
        if out:
            sys.stderr.write(out + "\n")
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        sys.stderr.write("bench.py: the ranks exited 0 without a JSON line\n")
        rc = 1
    return rc


def profiled(name, with_tag=False):
    """A committed rocprofv3 summary of THIS round (profiles/<tag>_<name>.json; the previous round's while this round's has
    not been made yet), or None."""
    for tag in (PROFILE_TAG, PROFILE_TAG_PREV):
        path = os.path.join(ROOT, "profiles", "%s_%s.json" % (tag, name))
        try:
            d = json.load(open(path))
            return (d, tag) if with_tag else d
        except Exception:   # noqa: BLE001
            continue
    return (None, PROFILE_TAG) if with_tag else None


def build_batch(api, workloads, rank, world, n_per_gpu, seconds, no_fuse, no_pack):
    """This rank's share of the job: global project ids rank, rank + world, ... (p mod world == rank)."""
    from termdaw_amd import batch as tb
    ids = tb.shard(n_per_gpu * world, world, rank)
    # the path's product is the integer PCM (what State::render hands to the WAV writer, state.rs:517-532); the f32
    # copy of the output vertex that the engine can keep for inspection is switched off (with it: `with_f32_copy`)
    opts = {"fuse_sources": 0 if no_fuse else 1, "packed_samples": 0 if no_pack else 1, "output_f32": 0}
    return tb.build_shard(api, lambda pid: workloads.config2(seconds=seconds, n_src=N_SRC, seed_offset=64 * pid), ids, opts)


def time_batch(batch, cs, steps, warmup, barrier, exchange):
    """W untimed + K timed steps bracketed by barriers; the peak exchange sits inside the timed region."""
    def step():
        batch.rewind()                       # fresh-after-refresh state (state.rs:467) for every project
        batch.render_all_async(cs, 16)
    # bring the device to its steady clocks first: the first ~50 ms of work after an idle period run measurably slower
    # (0.079 vs 0.073 ms per step measured), which is a property of the power state, not of the path
    t_pre = time.perf_counter()
    prewarm = 0
    while time.perf_counter() - t_pre < PREWARM_S:
        step()
        prewarm += 1
        if prewarm % 16 == 0:
            batch.sync()
    batch.sync()
    for _ in range(warmup):
        step()
    batch.sync()
    batch.host_times(reset=True)             # (the first steps allocate edge buffers: not steady-state host work)
    exchange.exchange()                      # warm the exchange path too (lazy initialisation is not render work)
    barrier()
    # HIP events around every launch of every PROF_EVERY-th step, on the engine's stream (the events cost a
    # few microseconds per launch -- a tenth of a single-project step if every render carried them)
    # (a short timed region -- the driver's K = 20 -- carries ONE sampled step, a middle one, not K / 8 + 1: each costs ~15 us of the
    # region; the sampling counter starts at this call, so `every` = K with K // 2 untimed-warm steps' worth of offset is not
    # available -- the engine samples submissions 0, every, 2 every, ...: every = K makes it the first step of the region)
    # Where a step launches ONE family only (the headline: one launch per step) the region carries no per-launch event pair --
    # one costs ~15 us, a quarter of a step -- but two marks on the engine's stream around the K submissions; the family's
    # average duration is the marked time over its launches.  Steps of several families (--no-fuse) keep sampled steps.
    batch.set_profiling(1)                   # (one sampled step outside the region: which families a step launches, how often)
    step()
    batch.sync()
    fam0 = batch.kernel_times()
    batch.set_profiling(0)
    by_marks = len(fam0) == 1
    every = 0 if by_marks else (PROF_EVERY if steps >= 64 else max(PROF_EVERY, steps))
    batch.set_profiling(every)
    barrier()
    start_wall = time.time()                 # (right behind the opening barrier: max - min over the ranks = start skew)
    t0 = time.perf_counter()
    batch.mark(0)
    for _ in range(steps):
        step()
    batch.mark(1)
    # The closing barrier.  With more than one rank it IS the path's only exchange -- td_batch_exchange_peaks: one
    # ncclAllReduce(max) of the peak table on device memory, queued by the library on the engine's own stream right behind the K
    # steps (no host synchronisation in between); no rank's stream gets past it before every rank has contributed, i.e. has
    # finished its K steps; exchange() returns when the stream has drained -- followed by torch.cuda.synchronize(): a second
    # collective (dist.barrier) behind it would time the same rendezvous twice.  A single rank without a process group has no
    # collective and takes the plain barrier.
    exchange.exchange()
    if exchange.is_collective():
        import torch
        torch.cuda.synchronize()
    else:
        barrier()
    t_end = time.perf_counter()
    dt = t_end - t0
    time_batch.last_ranks = {"dt_ms": dt * 1e3, "start_wall": start_wall}
    peaks = exchange.host()                  # (the report's copy of the table: outside the timed region, like the PCM it stays in HBM)
    ktimes = batch.kernel_times()
    batch.set_profiling(0)
    marked = batch.marked_ms()
    # this rank's own K steps: the time between the two marks on the engine's stream (HIP events: no host synchronisation of its
    # own inside the region); what is left of the region is the closing exchange -- the collective plus the wait for the slowest rank
    time_batch.last_ranks["render_ms"] = marked if marked > 0 else dt * 1e3
    time_batch.last_ranks["exchange_ms"] = max(0.0, dt * 1e3 - time_batch.last_ranks["render_ms"])
    if by_marks and marked > 0:
        (fam, (_, per_step)), = fam0.items()
        ktimes = {fam: (marked, steps * per_step)}
    return dt, ktimes, peaks, every, marked


def time_project(p, api, opts, reps):
    """One project rendered in a pipelined loop (fresh-after-refresh state each time): ms per render, kernel families."""
    sb, fb, g = p.build(api)
    for k, v in opts.items():
        g.set_option(k, v)

    def render():
        g.reset_normalize_vertices()
        fb.set_time(0)
        g.set_time(0)
        g.render_all_async(sb, fb, p.cs, 16)
    # steady device clocks first, as for the headline (PREWARM_S): the first tens of ms after an idle period run slower
    t_pre, n_pre = time.perf_counter(), 0
    while n_pre < 3 or time.perf_counter() - t_pre < 0.15:
        render()
        n_pre += 1
        if n_pre % 16 == 0:
            g.sync()
    g.sync()
    per = (time.perf_counter() - t_pre) / n_pre
    reps = max(reps, min(400, int(0.06 / max(per, 1e-6))))   # (a short render: enough of them for >= 60 ms of timed work)
    g.host_times(reset=True)     # (the first renders allocate buffers and arenas)
    t0 = time.perf_counter()
    for _ in range(reps):
        render()
    g.sync()
    ms = (time.perf_counter() - t0) / reps * 1e3
    g.set_profiling(1)
    for _ in range(2):
        render()
    g.sync()
    kt = g.kernel_times()
    g.set_profiling(0)
    hosts = g.host_times()
    kernels = sorted(((k, v[0] / 2.0, v[1] // 2, v[0] / max(v[1], 1)) for k, v in kt.items()), key=lambda r: -r[1])
    return ms, kernels, hosts, (sb, fb, g)


def time_config_batch(mk, api, P, opts, steps):
    """P identical-shape projects (variants 0 .. P-1) through td_batch_*: ms per step, per project."""
    from termdaw_amd import batch as tb
    b, first = tb.build_shard(api, lambda pid: mk(pid), list(range(P)), opts)

    def step():
        b.rewind()
        b.render_all_async(first.cs, 16)
    t_pre, n_pre = time.perf_counter(), 0
    while n_pre < 2 or time.perf_counter() - t_pre < 0.1:   # (steady clocks, as above)
        step()
        n_pre += 1
    b.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    b.sync()
    ms = (time.perf_counter() - t0) / steps * 1e3
    frames = first.cs * first.bl
    del b
    return {"projects": P, "steps": steps, "ms_per_step": round(ms, 4), "ms_per_project": round(ms / P, 4),
            "Msamples_per_s": round(frames * P / ms / 1e3, 1)}


def other_configs(api, workloads, ub, chain_ns):
    """BASELINE configs 1, 3, 4 (N = 1): ms per render in a pipelined loop, kernel breakdown, dominant kernel with its
    bound.  Configs 3 and 4 hold band-pass vertices and are reported in BOTH band modes -- "exact" (the default and the
    parity mode: bit-identical to the reference's serial recurrence) and "scan" (engine option band_mode 1, tolerance
    class <= 1e-6 RMS) -- each entry says which; each also as a batch of 8 and 32 such projects through td_batch_*.
    Bounds: config 1 launch latency; k_synth / k_band_scan f32 VALU issue (instruction count from the committed PMC
    pass); k_band_spec the dependent-chain latency of its warm-up walk (ns per dependent VALU measured in this process)."""
    out = []
    valu = profiled("valu") or {}
    issue_ns = float(ub.td_ubench_fma_issue_ns(SIMDS // 4)) if ub is not None else -1.0
    plan = (("config1", workloads.config1, None, 50, ()),
            ("config3", workloads.config3, "exact", 10, (8, 32)), ("config3", workloads.config3, "scan", 10, (8, 32)),
            ("config3", workloads.config3, "guard", 10, ()), ("config3", workloads.config3, "exact+sine", 6, ()),
            ("config4", workloads.config4, "exact", 4, (8, 32)), ("config4", workloads.config4, "scan", 10, (8, 32)),
            ("config4", workloads.config4, "guard", 10, (32,)))
    for name, mk, mode, reps, batches in plan:
        p = mk()
        frames = p.cs * p.bl
        opts = {} if mode is None else {"band_mode": {"scan": 1, "guard": 2}.get(mode, 0)}
        if mode == "exact+sine":   # what a bare td_graph does when nothing is set: every kind the reference's bytes (glibc's sinf on the device)
            opts["sine_mode"] = 1
        if mode == "guard":        # the front-end's two defaults: the scan AND the fast sine kinds under the guard (k_sine_probe in the render)
            opts["sine_mode"] = 2
        ms, kernels, hosts, built = time_project(p, api, opts, reps)
        g = built[2]
        dom = kernels[0]
        launches = sum(k[2] for k in kernels)
        entry = {"config": name, "frames": frames, "vertices": sum(len(v) for k, v in p.calls.items() if k.startswith("add_")),
                 "ms_per_render": round(ms, 4), "Msamples_per_s": round(frames / ms / 1e3, 1), "launches_per_render": int(launches),
                 "kernel_ms_per_render": round(sum(k[1] for k in kernels), 4),
                 "host_ms_per_render": {k: round(v / max(hosts["chunks"], 1), 4) for k, v in hosts.items() if k != "chunks"},
                 "kernels": [{"kernel": k[0], "ms_per_render": round(k[1], 4), "launches": int(k[2]), "avg_ms": round(k[3], 5)} for k in kernels],
                 "dominant": dom[0]}
        if mode == "guard":   # (the front-end's default: the scan under the guard -- what the guard saw of the timed renders)
            entry["guard"] = g.band_guard_stats()
        if mode is not None:
            entry["band_mode"] = mode
            entry["band_mode_note"] = ("exact: speculative-segment kernels, bit-identical to the reference's serial recurrence (default, parity mode)"
                                       if mode == "exact" else
                                       "exact+sine: the exact band-pass kernels AND engine option sine_mode 1 -- the oscillators evaluate glibc's sinf operation "
                                       "for operation (kernels.hip sin_glibc): the whole render bit for bit the oracle's (tests/test_gpu_sine_exact.py); "
                                       "what a bare td_graph does with nothing set" if mode == "exact+sine" else
                                       "guard: the front-end's defaults -- the scan kernels (band_mode 2) and the fast sine kinds (sine_mode 2) under the guard: every "
                                       "render estimates / measures its own deviation and is redone with the exact kernels and glibc's sinf when over 2e-7 RMS "
                                       "(tests/test_gpu_band_guard.py, tests/test_gpu_sine_guard.py)" if mode == "guard" else
                                       "scan: blocked affine scan (k_band_scan / k_band_chain), tolerance class: <= 1e-6 RMS and +-1 LSB against the "
                                       "oracle (tests/test_gpu_band_scan.py; measured RMS per chain depth: profiles/%s_scan_rms.txt)" % PROFILE_TAG)
        vkey = name if mode in (None, "exact") else name + "_scan"   # (the guarded launches are priced with the scan entry's counters)
        if mode == "exact+sine":
            vkey = "-"   # (no counters committed for the exact-sine form of k_synth: no bound claimed)
        insts = (valu.get(vkey, {}).get(dom[0]) or {}).get("SQ_INSTS_VALU")
        if name == "config1":
            floor = launches * 1.45e-3    # MI355X_MICROARCH.md price list, row "boundary": dependent kernel boundary 1.45 us
            entry["bound"] = {"kind": "launch latency", "floor_ms": round(floor, 5), "frac": round(floor / ms, 4),
                              "note": "%d dependent launches x 1.45 us (guide: same-stream kernel boundary); the kernels themselves move "
                                      "%.1f MB" % (launches, frames * (2 * 4 + 8 + 20) / 1e6)}
        elif dom[0] in ("k_synth", "k_sources", "k_band_scan"):
            if insts and issue_ns > 0:
                floor = insts * issue_ns / SIMDS * 1e-6
                # ... priced by instruction class instead of at the fastest one: the average issue time of the kernel's loop
                # instructions from its ISA (tools/isa_mix.py -> profiles/<tag>_isa_mix.json: packed f32, compares / selects /
                # min / max / conversions 1.75 ns, f64 1.85, transcendentals 3.4 against 1.06 for plain f32 add / mul / fma)
                mix = (profiled("isa_mix") or {}).get(((valu.get(vkey, {}).get(dom[0]) or {}).get("rocprof_kernel") or "").replace("tdk::", ""))
                fastest = {"floor_ms": round(floor, 4), "frac": round(floor / dom[3], 4)}
                if mix:
                    floor = insts * mix["avg_ns_per_valu"] / SIMDS * 1e-6
                entry["bound"] = {"kind": "f32 VALU issue", "floor_ms": round(floor, 4), "frac": round(floor / dom[3], 4),
                                  "priced": ("by instruction class: %.3f ns per VALU instruction (profiles/%s_isa_mix.json)" % (mix["avg_ns_per_valu"], PROFILE_TAG)) if mix
                                            else "every instruction at the fastest class",
                                  "at_fastest_class": fastest,
                                  "SQ_INSTS_VALU_profiled": insts, "profile": "profiles/%s_valu.json" % PROFILE_TAG,
                                  "ns_per_fma_per_simd_measured": round(issue_ns, 3),
                                  "note": "wave-level VALU instructions per launch (PMC, committed profile) x the issue time of the FASTEST class "
                                          "(v_fma_f32, 8 waves per SIMD, measured in this process) / 1024 SIMDs; compares, selects, min / max, "
                                          "conversions, packed and f64 ops issue slower, so the true floor is higher"}
            else:
                entry["bound"] = {"kind": "f32 VALU issue", "floor_ms": None, "frac": None, "note": "no committed PMC pass for this round"}
        elif dom[0] == "k_band_spec" and chain_ns and chain_ns > 0:
            st = g.band_stats()
            gmin = 1.0 - float(np.exp(np.float32(-2.0 * np.pi * 20.0 / 48000.0)))
            wq2 = (int(30.0 / gmin + 64.0) + 31) // 32 * 32      # the medium warm-up (engine option band_medium, default 30 / gamma)
            wq2 = (wq2 + 255) // 256 * 256
            steps = wq2 + 256
            floor = steps * 3 * chain_ns * 1e-6
            entry["bound"] = {"kind": "dependent-chain latency", "floor_ms": round(floor, 4), "frac": round(floor / dom[3], 4),
                              "steps_per_launch": steps, "ns_per_dependent_valu_measured": round(chain_ns, 3),
                              "band_stats_last_render": st,
                              "note": "medium warm-up (30/gamma from the block-response guess, 20 Hz stage) + one 256-frame segment, 3 dependent "
                                      "VALU per step, one wave alone on its SIMD (tools/ubench/ceilings.hip td_ubench_valu_chain_ns); the "
                                      "rest of the launch is the segment's output phase and the first workgroup's longer walk"}
        del built, g
        if batches:
            entry["batch"] = []
            for P in batches:
                try:
                    bsteps = 3 if (name == "config4" and mode == "exact") else 6
                    e = time_config_batch(lambda pid: mk(variant=pid), api, P, opts, bsteps)
                    e["per_project_over_single"] = round(e["ms_per_project"] / ms, 4)
                    entry["batch"].append(e)
                except Exception as ex:   # noqa: BLE001
                    entry["batch"].append({"projects": P, "error": str(ex)})
        out.append(entry)
    return out


def cold_tables(api, workloads, seconds, n_projects=16, reps=8):
    """The headline region over DISTINCT projects: n_projects config-2 projects (seeds offset by 64 per project: 16 x 20 MB of
    packed tables, more than the 256 MB Infinity Cache holds beside everything else) on ONE stream, each step renders the next
    one -- the same one launch per render as the headline, whose single project's tables never leave the cache."""
    from termdaw_amd import batch as tb
    opts = {"fuse_sources": 1, "packed_samples": 1, "output_f32": 0}
    b, first = tb.build_shard(api, lambda pid: workloads.config2(seconds=seconds, n_src=N_SRC, seed_offset=64 * (pid + 1)), list(range(n_projects)), opts)
    cs = first.cs

    def one(i):
        sb, fb, g = b.projects[i % n_projects]
        g.reset_normalize_vertices()
        fb.set_time(0)
        g.render_all_async(sb, fb, cs, 16)
    t_pre = time.perf_counter()
    i = 0
    while i < 2 * n_projects or time.perf_counter() - t_pre < PREWARM_S:
        one(i)
        i += 1
    b.sync()
    n = reps * n_projects
    t0 = time.perf_counter()
    for i in range(n):
        one(i)
    b.sync()
    ms = (time.perf_counter() - t0) / n * 1e3
    frames = cs * first.bl
    del b
    return {"projects": n_projects, "renders": n, "ms_per_render": round(ms, 5), "Msamples_per_s": round(frames / ms / 1e3, 1)}


def edge_buffer_mode(api, workloads, seconds, frames, copy_gbs, reps=5, copy_detail=None):
    """SURVEY 8(d)'s sum+normalize check on the SAME project with the edge-buffer model (engine option fuse_sources 0: one
    HBM buffer per source vertex, the Normalize reads 64 of them): k_sum + k_scale move 8k+8 + 8+8 = 536 B per frame
    (+ the 4 B PCM), which IS HBM traffic there (L2 hit 2 %, profiles/*_pmc_summary.json "nofuse")."""
    p = workloads.config2(seconds=seconds, n_src=N_SRC)
    ms, kernels, _, built = time_project(p, api, {"fuse_sources": 0, "output_f32": 1}, reps)
    kd = {k[0]: k[3] for k in kernels}
    del built
    if "k_sum" not in kd or "k_scale" not in kd:
        return None
    nbytes = 536 * frames
    gbs = nbytes / ((kd["k_sum"] + kd["k_scale"]) * 1e-3) / 1e9
    return {"k_sum_ms": round(kd["k_sum"], 5), "k_scale_ms": round(kd["k_scale"], 5), "k_sample_loop_ms": round(kd.get("k_sample_loop", 0.0), 5),
            "ms_per_render": round(ms, 4), "bytes": int(nbytes), "bytes_per_frame": 536, "GB/s": round(gbs, 1),
            "frac_hbm_peak": round(gbs / HBM_PEAK_GBS, 4), "frac_measured_copy": round(gbs / copy_gbs, 4),
            "measured_copy_GBs": round(copy_gbs, 1), "copy_GBs": copy_detail, "renders": reps,
            "k_sum_alone_GBs": round(520 * frames / (kd["k_sum"] * 1e-3) / 1e9, 1),
            "note": "run after the timed region, outside it: the headline project rebuilt with engine options fuse_sources 0 "
                    "(SURVEY 8(d)'s edge-buffer model) and output_f32 1 (pass B writes the f32 frames back, as the 8+8 of the model "
                    "says); HIP-event time per launch, %d renders; north_star asks >= 40 %% of the measured copy rate here" % reps}


def pinned_d2h_gbs(nbytes, reps=3):
    """The box's device -> page-locked host copy rate for one transfer of nbytes (torch, one stream), GB/s."""
    import torch
    n = max(1, nbytes // 4)
    dev = torch.empty(n, dtype=torch.int32, device="cuda").zero_()
    host = torch.empty(n, dtype=torch.int32, pin_memory=True)
    host.copy_(dev, non_blocking=True)
    torch.cuda.synchronize()
    best = 0.0
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        host.copy_(dev, non_blocking=True)
        e1.record()
        torch.cuda.synchronize()
        best = max(best, n * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del dev, host
    return best


def end_to_end(b64, cs, frames, render_only_ms_per_project):
    """SURVEY 8(d) "report separately with D2H + file write": the config-5 share (64 projects) through
    td_batch_render_to_files -- render in groups, a copy stream taking finished PCM to page-locked host memory under the
    later renders, host threads writing the WAV files (tmpfs) -- outside the timed region, never `value`."""
    import shutil
    import tempfile
    P = len(b64)
    nbytes = frames * 4 * P
    out = {"projects": P, "bytes": nbytes, "render_only_ms_per_project": round(render_only_ms_per_project, 5)}
    try:
        out["pinned_d2h_GBs_measured"] = round(pinned_d2h_gbs(nbytes // 8), 2)
    except Exception as e:   # noqa: BLE001
        out["pinned_d2h_GBs_measured"] = None
        out["pinned_error"] = str(e)[:80]
    group = 8
    cores = os.cpu_count() or 2

    def run(paths, reps, writers=32, fresh=False):
        best = None
        for _ in range(reps):
            for f in (paths or ()) if fresh else ():
                if os.path.exists(f):
                    os.unlink(f)       # (a NEW file per render, as a batch render to fresh output paths writes)
            b64.rewind()
            t = b64.render_to_files(cs, 16, 48000, paths, group=group, writers=writers)
            if best is None or t["wall_ms"] < best["wall_ms"]:
                best = t
        return best
    run(None, 1)   # (allocates the page-locked buffer)
    t = run(None, 3)
    gbs = t["bytes"] / (t["copy_span_ms"] * 1e-3) / 1e9
    out["render_d2h"] = {"ms_per_project": round(t["wall_ms"] / P, 5), "Msamples_per_s": round(frames * P / t["wall_ms"] / 1e3, 1),
                         "d2h_GBs": round(gbs, 2), "d2h_busy_GBs": round(t["bytes"] / (t["copy_busy_ms"] * 1e-3) / 1e9, 2),
                         "gpu_render_span_ms": round(t["gpu_render_span_ms"], 3), "copy_span_ms": round(t["copy_span_ms"], 3),
                         "wall_ms": round(t["wall_ms"], 3)}
    if out.get("pinned_d2h_GBs_measured"):
        out["render_d2h"]["d2h_frac_of_pinned_rate"] = round(gbs / out["pinned_d2h_GBs_measured"], 4)
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    d = tempfile.mkdtemp(prefix="td_e2e_", dir=base)
    try:
        paths = [os.path.join(d, "p%02d.wav" % i) for i in range(P)]
        run(paths, 1)
        # The write step is the host's and differs from box to box of the pool (profiles/NOTES.md#e2e): 32 threads overwriting
        # the previous call's files are the best form on some, 96 threads writing new files on others.  All four forms are
        # run; the line carries the best and what every form took.
        forms, best, bw, bf = {}, None, 0, False
        for w in sorted({max(1, min(32, cores // 2)), max(1, min(96, cores // 2))}):
            for fresh in (False, True):
                t = run(paths, 2, writers=w, fresh=fresh)
                forms["%d %s" % (w, "new" if fresh else "overwrite")] = round(t["wall_ms"] / P, 4)
                if best is None or t["wall_ms"] < best["wall_ms"]:
                    best, bw, bf = t, w, fresh
        t = best
        out["render_d2h_wav"] = {"ms_per_project": round(t["wall_ms"] / P, 5), "Msamples_per_s": round(frames * P / t["wall_ms"] / 1e3, 1),
                                 "write_span_ms": round(t["write_span_ms"], 3), "write_GBs": round(t["bytes"] / (max(t["write_span_ms"], 1e-6) * 1e-3) / 1e9, 2),
                                 "wall_ms": round(t["wall_ms"], 3), "writers": bw, "files": "new" if bf else "overwritten", "group": group,
                                 "dir": "tmpfs" if base else "tmp", "ms_per_project_by_form": forms}
        out["file_bytes_each"] = os.path.getsize(paths[0])
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return out


NOTES = "profiles/NOTES.md"   # the prose that used to ride in the line, by note id


def compact(o):
    """The ONE line the driver stores: every headline number, no prose (notes live in profiles/NOTES.md under the ids
    given here); `--full` prints the long form instead."""
    c = {k: o[k] for k in ("metric", "value", "unit", "n_gpus", "n_ranks_seen", "ranks", "exchange_backend", "steps", "warmup", "ms_per_step",
                           "higher_is_better", "scaling", "vs_baseline", "dtype", "data") if k in o}
    cfg = o["config"]
    c["config"] = {"workload": cfg["workload"], "frames_per_project": cfg["frames_per_project"], "projects_per_gpu": cfg["projects_per_gpu"],
                   "vertices_per_project": cfg["vertices_per_project"], "normalize": cfg["normalize_id"], "output": "int16 PCM in HBM",
                   "parallelism": cfg["parallelism_id"]}
    r = o.get("roofline")
    if r:
        rr = {"bound": r.get("bound"), "kernel": r.get("rocprof_kernel"), "avg_ms": r.get("avg_ms"), "achieved": r.get("achieved"),
              "peak": r.get("peak"), "unit": r.get("unit"), "frac": r.get("frac"), "traffic": r.get("traffic")}
        tp = r.get("traffic_profiled") or {}
        if tp:   # (flat: the driver's parser keeps the scalars of this object)
            rr["traffic_avg_us_rocprof"] = tp.get("avg_us_rocprof")
        for k in ("bytes_per_frame", "frac_of_ubench_ceiling", "ubench_ceiling_GBs", "hbm_compulsory_bytes", "hbm_compulsory_frac",
                  "wasted_vs_compulsory", "traffic_source", "l2_hit_profiled", "frac_l2_stream", "peak_l2_stream_GBs", "frac_gather_rows",
                  "peak_gather_rows_GBs", "valu_issue_ms", "frac_valu_issue",
                  "frac_of_measured_copy", "measured_copy_GBs"):
            if k in r:
                rr[k] = r[k]
        cg = r.get("copy_GBs") or {}
        if cg:
            rr["copy_own_kernel_GBs"], rr["copy_torch_GBs"] = cg.get("own_float4_kernel"), cg.get("torch_copy_")
        ct = r.get("cold_tables") or {}
        if "ms_per_render" in ct:   # (flat: a consumer that keeps only the scalars of this object keeps these)
            rr["cold_ms_per_render"] = ct["ms_per_render"]
            rr["cold_projects"] = ct.get("projects")
            rr["warm_ms_per_render"] = ct.get("warm_ms_per_render")
        rr["note"] = NOTES + "#roofline"
        c["roofline"] = rr
    if "cpu_baseline" in o:
        c["cpu_baseline"] = o["cpu_baseline"]
        c["gpu_over_cpu_1thread"] = o.get("gpu_over_cpu_1thread")
    e = o.get("edge_buffer_mode")
    if e:
        c["edge_buffer_mode"] = e if "error" in e else {k: e[k] for k in ("k_sum_ms", "k_scale_ms", "bytes_per_frame", "GB/s", "frac_hbm_peak", "frac_measured_copy",
                                                                           "measured_copy_GBs", "copy_GBs", "ms_per_render") if k in e}
        c["edge_buffer_mode"]["note"] = NOTES + "#edge_buffer_mode"
    c5 = o.get("config5")
    if c5:
        c["config5"] = {k: c5[k] for k in ("value", "unit", "projects_per_gpu", "projects", "steps", "ms_per_step", "ms_per_project", "ranks") if k in c5}
    sc = o.get("scanned")
    if sc:
        c["scanned"] = {k: sc[k] for k in ("ms_per_render", "Msamples_per_s", "scan_ms", "ms_per_render_with_f32_copy") if k in sc}
    if "with_f32_copy" in o:
        c["with_f32_copy_ms"] = o["with_f32_copy"]["ms_per_render"]
    if "pcie_inclusive" in o:
        c["pcie_inclusive"] = {k: o["pcie_inclusive"][k] for k in ("ms_per_render", "Msamples_per_s")}
    if "e2e" in o:
        e = o["e2e"]
        c["e2e"] = {k: e[k] for k in ("projects", "render_only_ms_per_project", "pinned_d2h_GBs_measured", "file_bytes_each", "error") if k in e}
        if "render_d2h" in e:
            c["e2e"]["render_d2h"] = {k: e["render_d2h"][k] for k in ("ms_per_project", "d2h_GBs", "d2h_frac_of_pinned_rate") if k in e["render_d2h"]}
        if "render_d2h_wav" in e:
            c["e2e"]["render_d2h_wav"] = {k: e["render_d2h_wav"][k] for k in ("ms_per_project", "writers", "files", "dir", "ms_per_project_by_form") if k in e["render_d2h_wav"]}
    if "cpu_baseline_all_cores" in o:
        a = o["cpu_baseline_all_cores"]
        c["cpu_all_cores"] = {"value": a["value"], "cores": a["cores"], "gpu_over": o.get("gpu_over_cpu_all_cores")}
    cf = o.get("configs")
    if isinstance(cf, list):
        rows = []
        for x in cf:
            row = {"config": x["config"], "ms": x["ms_per_render"], "Msps": x["Msamples_per_s"], "launches": x["launches_per_render"]}
            if "band_mode" in x:
                row["band_mode"] = x["band_mode"]
            if x.get("band_mode") == "exact+sine":   # (the line's budget: time only)
                rows.append(row)
                continue
            if "guard" in x:
                row["guard"] = {"redos": x["guard"]["redos"], "est": float("%.3g" % x["guard"]["max_est"])}
            ks = x.get("kernels") or []
            row["kernels"] = {k["kernel"]: k["ms_per_render"] for k in ks[:2]}
            b = x.get("bound") or {}
            if b:
                row["bound"] = {"kind": b.get("kind"), "floor_ms": b.get("floor_ms"), "frac": b.get("frac")}
            if x.get("batch"):
                row["batch_ms_per_project"] = {str(e2["projects"]): e2.get("ms_per_project", e2.get("error")) for e2 in x["batch"]}
            rows.append(row)
        c["configs"] = rows
        c["configs_note"] = NOTES + "#configs"
    elif cf:
        c["configs"] = cf
    if len(o.get("rooflines", [])) > 1:   # (one family per step: `roofline` is all there is to say)
        c["rooflines"] = [{k: r2.get(k) for k in ("kernel", "avg_ms", "launches", "achieved", "peak", "frac", "bytes_per_frame") if k in r2} for r2 in o.get("rooflines", [])]
    for k in ("kernel_timing_every", "marked_ms", "host_ms_per_step", "peak_table_entries", "device_bytes", "vertex_frames_per_s"):
        if k in o:
            c[k] = o[k]
    c["peak_table"] = o.get("peak_table", [])[:4]
    return c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cpu-worker", type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--projects-per-gpu", type=int, default=1, help="projects in this GPU's batch (config 5: 64)")
    ap.add_argument("--seconds", type=float, default=SECONDS, help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline only: no config5 / scanned / configs / ceilings of other kernels")
    ap.add_argument("--no-fuse", action="store_true", help="edge-buffer model: one HBM buffer per source vertex (no source inlining)")
    ap.add_argument("--no-pack", action="store_true", help="inlined sources gather the f32 sample form (8 B/frame) instead of the packed 16-bit one")
    ap.add_argument("--full", action="store_true", help="print the long form of the line (every note string and per-kernel breakdown) instead of the compact one")
    args = ap.parse_args()
    if args.cpu_worker is not None:
        cpu_worker(args.cpu_worker, args.seconds)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Invoked plainly with --gpus N: start the N ranks ourselves -- fresh child processes through torch.distributed.run,
        # BEFORE anything in this process touches HIP / torch.cuda (a process that has initialised the GPU must never exec
        # or fork GPU work) -- relay rank 0's JSON line and propagate the exit code.
        sys.exit(launch_ranks(args.gpus))

    # Everything any library writes to stdout while the bench runs (RCCL prints a version banner through the C runtime)
    # goes to stderr: stdout carries the ONE JSON line and nothing else.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import torch   # (before the engine: both then share ONE HIP runtime in this process)
    import torch.distributed as dist
    from termdaw_amd import api, workloads

    if api.device_count() < 1 or not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP render path has no CPU fallback")
    if os.environ.get("TD_BENCH_ONE_DEVICE") == "1":   # self-test on a 1-GPU box: every rank on device 0 (use with gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    api.set_device(local_rank)
    use_dist = world > 1 or os.environ.get("TD_BENCH_FORCE_DIST") == "1"   # the latter: 1-rank RCCL self-test
    backend = os.environ.get("TD_BENCH_BACKEND", "nccl")   # "gloo" only for the 1-GPU self-test above
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    def make_exchange(batch, n_per_gpu):
        """Per-project peak table of the whole job (n_per_gpu x world floats), see termdaw_amd.batch.PeakExchange."""
        from termdaw_amd import batch as tb
        ex = tb.PeakExchange(batch, n_per_gpu, rank, world, dist if use_dist else None, on_device=(backend == "nccl"), comm=make_exchange.comm)
        make_exchange.comm = ex.comm          # (one communicator for every batch of this process)
        return ex
    make_exchange.comm = None

    def gather_ranks(mine):
        """Every rank's own view of the timed region (outside it): what explains a scaling curve -- per-rank region and
        render times, the time each spent in the closing exchange (the wait for the slowest rank included), start skew."""
        vals = [mine["dt_ms"], mine["render_ms"], mine["exchange_ms"], mine["start_wall"]]
        rows = [vals]
        if use_dist:
            t = torch.tensor(vals, dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            outs = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(outs, t)
            rows = [[float(x) for x in o.cpu().tolist()] for o in outs]
        a = np.array(rows, dtype=np.float64)
        return {"n": int(a.shape[0]),
                "dt_ms_min": round(float(a[:, 0].min()), 4), "dt_ms_max": round(float(a[:, 0].max()), 4),
                "render_ms_min": round(float(a[:, 1].min()), 4), "render_ms_max": round(float(a[:, 1].max()), 4),
                "exchange_ms": round(float(a[0, 2]), 4), "exchange_ms_min": round(float(a[:, 2].min()), 4),
                "start_skew_us": round(float((a[:, 3].max() - a[:, 3].min()) * 1e6), 1)}

    def reduce_max(dt):
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---- the timed region: this rank's batch of config-2 projects ----
    P = max(1, args.projects_per_gpu)
    batch, project = build_batch(api, workloads, rank, world, P, args.seconds, args.no_fuse, args.no_pack)
    cs, bl = project.cs, project.bl
    frames = cs * bl
    main_exchange = make_exchange(batch, P)
    exchange_backend = main_exchange.backend()
    dt, ktimes, peaks, prof_every, marked_ms = time_batch(batch, cs, args.steps, args.warmup, barrier, main_exchange)
    ranks = gather_ranks(time_batch.last_ranks)
    dt = reduce_max(dt)
    device_bytes = sum(g.device_bytes() for _, _, g in batch.projects)
    host = batch.host_times()

    # ---- config 5 proper beside it: 64 projects per GPU through the same path (skipped when the main region already is that) ----
    c5 = None
    want_c5 = not args.no_extras and P != 64 and not args.no_fuse and not args.no_pack
    if want_c5:
        # every rank must take the same branch (the section holds collectives): 64 resident projects need ~6.5 GB
        ok = torch.tensor([1 if torch.cuda.mem_get_info()[0] > (10 << 30) else 0], dtype=torch.int32,
                          device="cuda" if backend == "nccl" else "cpu")
        if use_dist:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        want_c5 = bool(int(ok.item()))
    if want_c5:
        c5_steps = max(2, min(10, args.steps))
        b64, _ = build_batch(api, workloads, rank, world, 64, args.seconds, False, False)
        dt5, kt5, pk5, _, _ = time_batch(b64, cs, c5_steps, 2, barrier, make_exchange(b64, 64))
        ranks5 = gather_ranks(time_batch.last_ranks)
        dt5 = reduce_max(dt5)
        c5 = {"projects_per_gpu": 64, "projects": 64 * world, "steps": c5_steps, "ms_per_step": round(dt5 / c5_steps * 1e3, 4),
              "ms_per_project": round(dt5 / c5_steps / 64 * 1e3, 5),
              "value": round(frames * 64 * world * c5_steps / dt5 / 1e6, 2), "unit": "Msamples/s",
              "kernels": {k: round(v[0] / max(v[1], 1), 5) for k, v in kt5.items()},
              "host_ms_per_step": {k: round(v / max(b64.host_times(reset=False)["steps"], 1), 4) for k, v in b64.host_times(reset=False).items() if k != "steps"},
              "peak_table_entries": int(len(pk5)), "peak_table_min_max": [round(float(np.min(pk5)), 6), round(float(np.max(pk5)), 6)],
              "ranks": ranks5,
              "note": "BASELINE config 5's per-GPU share: 64 independent config-2 projects (seed offset 64 x project id) resident per GPU, "
                      "one batch submission per step (launches of the 64 projects merged into one grid per kernel family), ONE "
                      "all-reduce(max) of the %d-entry peak table in the timed region" % (64 * world)}
        e2e = None
        if rank == 0 and world == 1:
            try:
                e2e = end_to_end(b64, cs, frames, dt5 / c5_steps / 64 * 1e3)
            except Exception as e:   # noqa: BLE001
                e2e = {"error": str(e)[:200]}
        del b64

    result_line = None
    if rank == 0:
        total_frames = frames * P * args.steps * world
        value = total_frames / dt / 1e6
        fused, packed = not args.no_fuse, not args.no_pack
        abf = algorithmic_bytes_per_frame(N_SRC, fused, packed)
        single_pass = fused and "k_scale" not in ktimes
        if single_pass:
            # SumDesc modes 4 / 5 (engine option single_pass_normalize): the running peak is found inside the summing launch
            # through granules, the frames leave the registers as int16 PCM -- no raw-sum write, no pass B (mode 4 keeps a
            # k_norm_fix check launch for grids that are not resident at once; mode 5, the resident grid, is this ONE launch)
            abf["k_sum"] = (4.0 if packed else 8.0) * N_SRC + 4.0
        survey_abf = algorithmic_bytes_per_frame(N_SRC, False, False)   # SURVEY 8(d) edge-buffer figure
        ub = None
        try:
            ub = ubench()
        except Exception as e:   # noqa: BLE001
            sys.stderr.write("bench.py: no ubench library (%s): ceilings omitted\n" % e)

        # the HBM denominator measured on this box (1 GiB each way: beyond the Infinity Cache) -- the repo's own
        # float4 copy kernel (tools/ubench/ceilings.hip k_stream: 16 B per lane in, 16 B out) and, beside it, torch's copy_
        own_copy_gbs = None
        if ub is not None:
            try:
                fr = (1 << 30) // 8
                ms_c = float(ub.td_ubench_stream(fr, 1, 1, 10))
                if ms_c > 0:
                    own_copy_gbs = 2.0 * fr * 8 / (ms_c * 1e-3) / 1e9
            except Exception:   # noqa: BLE001
                own_copy_gbs = None
        copy_gbs = HBM_COPY_GBS
        try:
            n = 256 << 20
            src = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
            dst = torch.empty_like(src)
            for _ in range(3):
                dst.copy_(src)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                dst.copy_(src)
            e1.record()
            torch.cuda.synchronize()
            copy_gbs = 10 * 2 * n * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9   # read + write bytes
            del src, dst
        except Exception:   # noqa: BLE001
            copy_gbs = HBM_COPY_GBS
        torch_copy_gbs = copy_gbs
        if own_copy_gbs and own_copy_gbs > copy_gbs:
            copy_gbs = own_copy_gbs      # the denominator is the faster of the two copies this box shows

        # ---- measured ceilings for the kernels of the timed region ----
        lens = np.array([project.assets["s%02d" % k].pcm.shape[0] for k in range(N_SRC)], dtype=np.uint32)
        nq = 4 if frames >= 2600 * 1024 else (2 if frames >= 1800 * 1024 else 1)
        gather_ms = gather_l2_ms = stream_ms = None
        if ub is not None and fused and packed:
            lp = lens.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
            gx = (frames + 1024 * nq - 1) // (1024 * nq)
            per = max(1, (256 * 4) // gx)                     # the engine's batch slicing (launch_sum, kernels.hip)
            it = 20 if P == 1 else 5
            gather_ms = min(float(ub.td_ubench_gather(lp, N_SRC, frames, nq, it, P, per)), float(ub.td_ubench_gather(lp, N_SRC, frames, nq, it, P, 0)))
            small = np.full(N_SRC, 1021, dtype=np.uint32)     # 64 x 4 KB: resident in every XCD's L2 (and mostly in L1)
            sp = small.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
            gather_l2_ms = min(float(ub.td_ubench_gather(sp, N_SRC, frames, nq, it, P, per)), float(ub.td_ubench_gather(sp, N_SRC, frames, nq, it, P, 0)))
        if ub is not None:
            stream_ms = float(ub.td_ubench_stream(frames, 1, 1, 20))
        tp, tp_tag = profiled("pmc_summary", with_tag=True)
        tp = tp or {}
        mode_key = "nofuse" if args.no_fuse else ("fused_f32" if args.no_pack else "fused")

        kernels = []
        for name, (ms, launches) in sorted(ktimes.items(), key=lambda kv: -kv[1][0]):
            avg_ms = ms / max(launches, 1)                    # one launch covers the P projects of the batch
            bytes_per_launch = abf.get(name, 0.0) * frames * P
            gbs = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            prof = (tp.get(mode_key) or {}).get(name) or {}
            row = {"kernel": name, "avg_ms": round(avg_ms, 5), "launches": int(launches), "projects_per_launch": P,
                   "algorithmic_bytes_per_launch": int(bytes_per_launch), "achieved": round(gbs, 1), "unit": "GB/s",
                   # fabric-side (L2-miss) bytes per launch from the committed rocprofv3 --pmc passes of this command -- the counters
                   # cannot be read from inside a process that is being timed; `traffic_profiled` says which file and how it was made
                   "traffic": prof.get("hbm_side_bytes_per_launch"),
                   "traffic_profiled": None if "hbm_side_bytes_per_launch" not in prof else {
                       "bytes_per_launch": prof["hbm_side_bytes_per_launch"], "l2_hit_rate": prof.get("l2_hit_rate"),
                       "avg_us_rocprof": prof.get("avg_us"),
                       "profile": "profiles/%s_pmc_summary.json" % tp_tag,
                       "note": "separate rocprofv3 --pmc passes of this command (FETCH_SIZE x 2 per the gfx950 half-count rule + "
                               "WRITE_SIZE); fabric-side bytes: Infinity-Cache hits are included, so HBM bytes are at most this"}}
            if name == "k_sum" and fused and packed and gather_ms and gather_ms > 0:
                ceil_gbs = bytes_per_launch / (gather_ms * 1e-3) / 1e9
                compulsory = (int(lens.sum()) * 4 + (4 if single_pass else 8) * frames) * P
                row.update({"bound": "cache-gather ceiling (measured)", "peak": round(ceil_gbs, 1), "frac": round(gbs / ceil_gbs, 4),
                            "ceiling_ms": round(gather_ms, 5),
                            "ceiling_l2_resident_GBs": None if not gather_l2_ms else round(bytes_per_launch / (gather_l2_ms * 1e-3) / 1e9, 1),
                            "frac_of_l2_resident_ceiling": None if not gather_l2_ms else round(gather_l2_ms / avg_ms, 4),
                            "hbm_compulsory_bytes": compulsory,
                            "hbm_compulsory_frac": round(compulsory / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                            "bytes_per_frame": abf[name],
                            "note": k_sum_roofline_note(single_pass, lens.sum() * 4 / 1e6, frames, P)})
            else:
                row.update({"bound": "hbm", "peak": HBM_PEAK_GBS, "frac": round(gbs / HBM_PEAK_GBS, 4),
                            "frac_of_measured_copy": round(gbs / copy_gbs, 4), "bytes_per_frame": abf.get(name)})
                if name == "k_scale" and stream_ms and stream_ms > 0:
                    row["stream_ceiling_ms"] = round(stream_ms, 5)
                    row["note"] = ("Normalize pass B: 8 B raw sum in + 4 B PCM out per frame; stream_ceiling_ms = a bare 8 B in / 8 B out "
                                   "float4 stream over the same %d frames (tools/ubench/ceilings.hip), i.e. launch ramp + tail included" % frames)
            kernels.append(row)
        dom = kernels[0] if kernels else None
        rocprof_kernel = None
        if dom:
            if dom["kernel"] == "k_sum":
                rocprof_kernel = (("tdk::k_sum16w<4, true>" if nq == 4 else "tdk::k_sum16w<2, true>" if nq == 2 else "tdk::k_sum<3>") if packed
                                  else ("tdk::k_sum16w<2, false>" if frames >= 1800 * 1024 else "tdk::k_sum<2>")) if fused else "tdk::k_sum<1>"
            else:
                rocprof_kernel = "tdk::" + dom["kernel"]
        roofline = None
        if dom:
            roofline = {k: dom.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_profiled",
                                                 "bytes_per_frame", "ceiling_ms", "ceiling_l2_resident_GBs", "frac_of_l2_resident_ceiling",
                                                 "hbm_compulsory_bytes", "hbm_compulsory_frac", "frac_of_measured_copy", "note") if k in dom}
            roofline["avg_ms"] = dom["avg_ms"]
            roofline["rocprof_kernel"] = rocprof_kernel   # the engine times launch FAMILIES; this instantiation is the rocprofv3 name
            if "hbm_compulsory_bytes" in dom:
                # the same launch against the HBM roofline in its plain form: the bytes that must cross HBM once per launch
                # (every sample table once + the output write) over the launch time, against 8 TB/s
                hb = dom["hbm_compulsory_bytes"] / (dom["avg_ms"] * 1e-3) / 1e9
                roofline["hbm_form"] = {"bound": "hbm", "achieved": round(hb, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": round(hb / HBM_PEAK_GBS, 4), "bytes": dom["hbm_compulsory_bytes"],
                                        "note": "compulsory HBM bytes only: the launch is bound by its cache-served gathers (`bound` above), not by HBM"}
            roofline["measured_copy_GBs"] = round(copy_gbs, 1)
            roofline["copy_GBs"] = {"own_float4_kernel": None if not own_copy_gbs else round(own_copy_gbs, 1), "torch_copy_": round(torch_copy_gbs, 1),
                                    "guide_float4": HBM_COPY_GBS, "bytes_each_way": 1 << 30}
            vp = ((profiled("valu") or {}).get("config2") or {}).get(dom["kernel"])
            if vp:
                roofline["valu_profiled"] = dict(vp, profile="profiles/%s_valu.json" % PROFILE_TAG)
                if fused and packed and dom["kernel"] == "k_sum" and vp.get("SQ_INSTS_VALU"):
                    # what the launch costs in VALU issue alone: wave-level instructions (profiled) x the measured issue time of
                    # their class -- SDWA conversions and packed f32, 1.75 ns per wave and SIMD (tools/ubench/issue_rate.hip) --
                    # over the chip's 1 024 SIMDs: the second thing the launch is limited by, beside the cache hierarchy
                    roofline["valu_issue_ms"] = round(vp["SQ_INSTS_VALU"] * 1.75e-9 / 1024.0 * 1e3, 5)
                    roofline["frac_valu_issue"] = round(roofline["valu_issue_ms"] / dom["avg_ms"], 4)
            if dom.get("traffic") and dom.get("hbm_compulsory_bytes"):
                roofline["wasted_vs_compulsory"] = round(dom["traffic"] / dom["hbm_compulsory_bytes"], 2)
                roofline["traffic_source"] = "profiles/%s_pmc_summary.json" % tp_tag   # (rocprofv3 --pmc FETCH_SIZE x 2 [gfx950 half-count] + WRITE_SIZE, per launch: NOTES.md#roofline)
            if fused and packed and dom["kernel"] == "k_sum":
                # two hardware-denominated readings beside the measured ceiling (guide figures only, no ubench):
                #  frac_of_l2_peak: the gathered + written bytes against the aggregate L2 bandwidth;
                #  l2_mall_split: the bytes split by the PROFILED L2 hit rate -- hits at the guide's L2 gather rate, the rest at
                #  its Infinity-Cache gather rate, one after the other -- as a time floor for this launch
                gathered = dom["algorithmic_bytes_per_launch"]
                roofline["frac_of_l2_peak"] = round(dom["achieved"] / L2_PEAK_GBS, 4)
                roofline["l2_peak_GBs"] = L2_PEAK_GBS
                hit = ((dom.get("traffic_profiled") or {}).get("l2_hit_rate"))
                if hit is not None:
                    floor_ms = (gathered * hit / (L2_GATHER_GBS * 1e9) + gathered * (1.0 - hit) / (MALL_GATHER_GBS * 1e9)) * 1e3
                    roofline["l2_mall_split"] = {"l2_hit_rate_profiled": hit, "l2_gather_GBs": L2_GATHER_GBS, "mall_gather_GBs": MALL_GATHER_GBS,
                                                 "floor_ms": round(floor_ms, 5), "frac": round(floor_ms / dom["avg_ms"], 4),
                                                 "note": "bytes x hit rate / 17.8 TB/s + bytes x (1 - hit rate) / 8.6 TB/s (MI355X_MICROARCH.md, "
                                                         "'Indexed rows: gather'), hit rate from profiles/%s_pmc_summary.json" % tp_tag}
                    # Two guide-priced floors for THIS launch's split, both printed, the HARDER one as `frac` (round-5 review: the
                    # gathers are contiguous 64 B per lane -- a wave reads 4 KB in a row -- so the hits are priced at the L2's STREAM
                    # rate, 34.5 TB/s, not at the rate of random 1 152-byte rows; the misses at the Infinity-Cache rate either way):
                    #   frac_l2_stream   = (bytes x hit / 34.5 TB/s + bytes x (1 - hit) / 8.6 TB/s) / measured
                    #   frac_gather_rows = (bytes x hit / 17.8 TB/s + bytes x (1 - hit) / 8.6 TB/s) / measured   (round 5's `frac`)
                    # No dependency on this repo's own ceiling kernel, which stays beside them (frac_of_ubench_ceiling).
                    stream_floor_ms = (gathered * hit / (L2_PEAK_GBS * 1e9) + gathered * (1.0 - hit) / (MALL_GATHER_GBS * 1e9)) * 1e3
                    roofline["frac_of_ubench_ceiling"] = roofline.get("frac")
                    roofline["ubench_ceiling_GBs"] = roofline.get("peak")
                    roofline["frac_gather_rows"] = round(floor_ms / dom["avg_ms"], 4)
                    roofline["peak_gather_rows_GBs"] = round(gathered / (floor_ms * 1e-3) / 1e9, 1)
                    roofline["frac_l2_stream"] = round(stream_floor_ms / dom["avg_ms"], 4)
                    roofline["peak_l2_stream_GBs"] = round(gathered / (stream_floor_ms * 1e-3) / 1e9, 1)
                    roofline["l2_hit_profiled"] = hit
                    roofline["bound"] = "l2 stream + infinity cache"   # (hits at 34.5 TB/s, misses at 8.6 TB/s: NOTES.md#roofline)
                    roofline["peak"] = roofline["peak_l2_stream_GBs"]
                    roofline["frac"] = roofline["frac_l2_stream"]
                else:
                    roofline["bound"] = "l2+infinity-cache gather (ceiling measured in-process: tools/ubench)"
            if fused and dom["kernel"] == "k_sum":
                roofline["survey_model"] = {
                    "bytes_per_frame": survey_abf["k_sum"],
                    "note": "SURVEY 8(d)'s edge-buffer figure (8k+8) for the same launch time -- not a bandwidth: the k source edge "
                            "buffers do not exist in this mode (--no-fuse runs that model, where the kernel is HBM-bound)",
                    "equivalent_GBs": round(survey_abf["k_sum"] * frames * P / (dom["avg_ms"] * 1e-3) / 1e9, 1)}
        out = {
            "metric": "offline render Msamples/sec (stereo 48 kHz) + % HBM roofline, 64-vertex graph",
            "value": round(value, 2),
            "unit": "Msamples/s",
            "n_gpus": world,
            "n_ranks_seen": dist.get_world_size() if use_dist else 1,   # what the process group itself reports
            "ranks": ranks,   # per-rank view of the timed region: dt / render min-max, exchange_ms (rank 0), start skew (profiles/NOTES.md#ranks)
            "exchange_backend": exchange_backend,      # "rccl-native": td_batch_exchange_peaks -- ncclAllReduce(max) on the engine's own stream, device memory; "host-callback": the gloo self-test; "none": one rank without a process group
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE config 2: 64 sampleloop -> 1 normalize, %g s @48 kHz, bl 1024, 16-bit PCM out; %d project(s) "
                                   "per GPU per step (project p of the job on rank p mod N, seed offset 64*p)" % (args.seconds, P),
                       "frames_per_project": frames, "projects_per_gpu": P, "vertices_per_project": N_SRC + 1,
                       "source_inlining": not args.no_fuse, "packed_samples": (not args.no_fuse) and (not args.no_pack),
                       "normalize_id": (("one launch" if "k_norm_fix" not in ktimes else "one launch + check launch") if single_pass else "two passes"),
                       "parallelism_id": "project p on rank p mod N; one RCCL all-reduce(max) of the %d-entry peak table" % (P * world),
                       "normalize": ("single pass, ONE launch: sum, running peak through in-launch granules, scale and quantise (k_sum16w mode 5)"
                                     if "k_norm_fix" not in ktimes else
                                     "single pass: running peak through in-launch granules (k_sum16w mode 4) + k_norm_fix check") if single_pass
                                    else "two passes: k_sum (sum + block peaks), k_scale",
                       "output": "int16 PCM in HBM (engine option output_f32 0: no f32 copy of the output vertex is kept)",
                       "parallelism": "projects sharded across GPUs; RCCL all-reduce(max) of the %d-entry peak table only" % (P * world)},
            "roofline": roofline,
            "rooflines": kernels,
            "kernel_timing": ("HIP events around each launch of every %dth step of the timed region, engine stream" % prof_every) if prof_every else
                             "two HIP-event marks on the engine's stream around the K submissions of the timed region (a step launches one family only): average = marked time / launches",
            "marked_ms": round(marked_ms, 5),
            "kernel_timing_every": prof_every,
            "prewarm": "%.2f s of untimed steps before the W warm-up steps (steady device clocks)" % PREWARM_S,
            "host_ms_per_step": {k: round(v / max(host["steps"], 1), 5) for k, v in host.items() if k != "steps"},
            "peak_table": [round(float(x), 6) for x in peaks[:16]],
            "peak_table_entries": int(len(peaks)),
            "device_bytes": int(device_bytes),
            # vertex-frames/s = frames x vertices reached from the output
            "vertex_frames_per_s": round(value * 1e6 * (N_SRC + 1), 0),
        }
        if c5:
            out["config5"] = c5
            if e2e:
                out["e2e"] = e2e
        if world == 1 and not args.no_extras:
            sb0, fb0, g0 = batch.projects[0]
            # the reference's recommended workflow: `normalize` (scan_exact, state.rs:473) then `render`; the scan is paid
            # once per project edit, the render each time -- both are reported
            tsc = []
            for _ in range(3):
                t0 = time.perf_counter()
                g0.true_normalize_scan(sb0, fb0, cs)
                tsc.append(time.perf_counter() - t0)
            reps = max(10, min(100, args.steps))

            def loop(fresh):
                for _ in range(3):
                    if fresh:
                        g0.reset_normalize_vertices()
                    fb0.set_time(0)
                    g0.render_all_async(sb0, fb0, cs, 16)
                g0.sync()
                t0 = time.perf_counter()
                for _ in range(reps):
                    if fresh:
                        g0.reset_normalize_vertices()
                    fb0.set_time(0)
                    g0.render_all_async(sb0, fb0, cs, 16)
                g0.sync()
                return (time.perf_counter() - t0) / reps
            ts = loop(False)
            g0.set_profiling(1)
            loop(False)
            kts = g0.kernel_times()
            g0.set_profiling(0)
            g0.set_option("output_f32", 1)
            ts_f32 = loop(False)
            out["scanned"] = {"ms_per_render": round(ts * 1e3, 4), "Msamples_per_s": round(frames / ts / 1e6, 1),
                              "ms_per_render_with_f32_copy": round(ts_f32 * 1e3, 4),
                              "scan_ms": round(sorted(tsc)[1] * 1e3, 4),
                              "kernels": {k: round(v[0] / max(v[1], 1), 5) for k, v in kts.items()},
                              "note": "renders after true_normalize_scan (graph.rs:222-237), pipelined like the timed region: the peak is "
                                      "known, so the summing kernel scales, pans and quantises out of registers (speculative single pass) "
                                      "and k_norm_fix only checks that no block exceeded it"}
            ts_fresh_f32 = loop(True)
            g0.set_option("output_f32", 0)
            out["with_f32_copy"] = {"ms_per_render": round(ts_fresh_f32 * 1e3, 4), "Msamples_per_s": round(frames / ts_fresh_f32 / 1e6, 1),
                                    "note": "the timed region's fresh (un-scanned) render with engine option output_f32 1 (the library default): "
                                            "pass B also writes the scaled f32 frames of the output vertex back (read by td_graph_read_f32, never by "
                                            "the WAV sink) -- round 1's headline was measured this way"}
            # outside the timed region, for reference only (never `value`): one render plus the copy of its
            # 16-bit PCM to host memory (pageable numpy array), median of 5
            ts = []
            for _ in range(5):
                step_t0 = time.perf_counter()
                g0.reset_normalize_vertices()
                fb0.set_time(0)
                g0.render_all(sb0, fb0, cs, 16, want_f32=False, want_pcm=True)
                ts.append(time.perf_counter() - step_t0)
            ts.sort()
            out["pcie_inclusive"] = {"ms_per_render": round(ts[2] * 1e3, 4), "Msamples_per_s": round(frames / ts[2] / 1e6, 1),
                                     "note": "render + D2H of the PCM into pageable host memory; not part of `value`"}
            if roofline is not None and not args.no_fuse and not args.no_pack:
                try:
                    ct = cold_tables(api, workloads, args.seconds)
                    ct["warm_ms_per_render"] = out["ms_per_step"] / P
                    roofline["cold_tables"] = ct
                except Exception as e:   # noqa: BLE001
                    roofline["cold_tables"] = {"error": str(e)[:120]}
            chain_ns = float(ub.td_ubench_valu_chain_ns()) if ub is not None else None
            try:
                out["edge_buffer_mode"] = edge_buffer_mode(api, workloads, args.seconds, frames, copy_gbs, copy_detail=(roofline or {}).get("copy_GBs"))
            except Exception as e:   # noqa: BLE001
                out["edge_buffer_mode"] = {"error": str(e)}
            try:
                out["configs"] = other_configs(api, workloads, ub, chain_ns)
            except Exception as e:   # noqa: BLE001
                out["configs"] = {"error": str(e)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(project, frames)
            out["gpu_over_cpu_1thread"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
            if not args.no_extras:
                try:
                    ncores = len(os.sched_getaffinity(0))
                except AttributeError:
                    ncores = os.cpu_count() or 1
                allc = cpu_all_cores(args.seconds, min(ncores, 256))
                if allc:
                    out["cpu_baseline_all_cores"] = allc
                    out["gpu_over_cpu_all_cores"] = round(out["value"] / allc["value"], 1)
        result_line = json.dumps(out if args.full else compact(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    try:
        ctypes.CDLL(None).fflush(None)   # (what the C runtime still buffers belongs to stderr too)
    except Exception:   # noqa: BLE001
        pass
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    os.close(real_stdout)
    if result_line is not None:
        print(result_line, flush=True)


if __name__ == "__main__":
    main()
