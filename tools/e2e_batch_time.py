"""td_batch_render_to_files on the config-5 share (64 config-2 projects -> 64 WAV files on tmpfs), a few repetitions (GPU box):
    python tools/e2e_batch_time.py [projects] [group] [writers]"""
import os, shutil, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from termdaw_amd import api, workloads as W, batch as tb

P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
group = int(sys.argv[2]) if len(sys.argv) > 2 else 8
writers = int(sys.argv[3]) if len(sys.argv) > 3 else 32
b, first = tb.build_shard(api, lambda pid: W.config2(seed_offset=64 * pid), list(range(P)), {"fuse_sources": 1, "packed_samples": 1, "output_f32": 0})
d = tempfile.mkdtemp(dir="/dev/shm")
try:
    paths = [os.path.join(d, "p%03d.wav" % i) for i in range(P)]
    for rep in range(6):
        for f in paths:
            if rep % 2 == 0 and os.path.exists(f):
                os.unlink(f)           # (even repetitions write NEW files, odd ones overwrite the previous repetition's)
        b.rewind()
        t = b.render_to_files(first.cs, 16, 48000, paths, group=group, writers=writers)
        print("%s files: wall %.2f ms = %.4f ms/project  copy span %.2f  write span %.2f  render span %.2f" %
              ("new" if rep % 2 == 0 else "old", t["wall_ms"], t["wall_ms"] / P, t["copy_span_ms"], t["write_span_ms"], t["gpu_render_span_ms"]))
finally:
    shutil.rmtree(d, ignore_errors=True)
