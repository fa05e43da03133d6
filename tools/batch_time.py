"""A batch of P config-3 / config-4 projects through td_batch_* (GPU box): ms per step, kernel families, host phases.
    python tools/batch_time.py c4 8 [band_mode]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from termdaw_amd import api, workloads as W, batch as tb

if __name__ == "__main__":
    which, P = sys.argv[1], int(sys.argv[2])
    mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    seconds = float(sys.argv[4]) if len(sys.argv) > 4 else 60.0
    mk = {"c3": W.config3, "c4": W.config4}[which]
    b, first = tb.build_shard(api, lambda pid: mk(seconds=seconds, variant=pid), list(range(P)),
                                 dict({"band_mode": mode}, **{k: int(v) for k, v in (kv.split("=") for kv in filter(None, os.environ.get("TD_OPTS", "").split(",")))}))
    def step():
        b.rewind()
        b.render_all_async(first.cs, 16)
    for _ in range(2):
        step()
    b.sync()
    b.host_times(reset=True)
    t0 = time.perf_counter()
    n = 4
    for _ in range(n):
        step()
    b.sync()
    ms = (time.perf_counter() - t0) / n * 1e3
    host = b.host_times()
    b.set_profiling(1)
    step(); b.sync()
    kt = b.kernel_times()
    b.set_profiling(0)
    print("%s x%d band_mode %d: %.3f ms/step = %.3f ms/project; host %s" % (which, P, mode, ms, ms / P, {k: round(v / max(host["steps"], 1), 3) for k, v in host.items() if k != "steps"}))
    print("  " + "  ".join("%s %.3f ms x%d" % (k, v[0], v[1]) for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0])))
