"""Shader / memory clocks while one config renders in a loop (GPU box):  python tools/clock_watch.py c4 [band_mode] [seconds]
Renders in a child process; this process samples `rocm-smi --showclocks` once a second."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    from termdaw_amd import api, workloads as W
    which, mode, secs = sys.argv[2], int(sys.argv[3]), float(sys.argv[4])
    p = {"c2": W.config2, "c3": W.config3, "c4": W.config4}[which]()
    sb, fb, g = p.build(api)
    g.set_option("band_mode", mode)
    t0, n = time.time(), 0
    while time.time() - t0 < secs:
        for _ in range(16):
            g.reset_normalize_vertices(); fb.set_time(0); g.set_time(0)
            g.render_all_async(sb, fb, p.cs, 16)
            n += 1
        g.sync()
    print("%s band_mode %d: %d renders, %.3f ms each" % (which, mode, n, (time.time() - t0) / n * 1e3))
    sys.exit(0)
which = sys.argv[1] if len(sys.argv) > 1 else "c4"
mode = sys.argv[2] if len(sys.argv) > 2 else "1"
secs = sys.argv[3] if len(sys.argv) > 3 else "8"
child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", which, mode, secs])
time.sleep(3.0)
while child.poll() is None:
    out = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True).stdout
    print(" | ".join(l.split(":", 1)[-1].strip() for l in out.splitlines() if "sclk" in l or "mclk" in l or "fclk" in l))
    time.sleep(1.0)
sys.exit(child.returncode)
