"""A/B two builds of the HIP library on bench config 2, alternating rounds on one device (rule 24).
Usage: python tools/ab_lib.py libA.so libB.so [--no-fuse]"""
import json, os, subprocess, sys
a, b = sys.argv[1], sys.argv[2]
extra = sys.argv[3:]
res = {a: [], b: []}
for r in range(3):
    for lib in (a, b):
        out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--steps", "100", "--warmup", "10"] + extra,
                             env=dict(os.environ, TD_LIB=os.path.abspath(lib)), capture_output=True, text=True).stdout
        d = json.loads(out.strip().splitlines()[-1])
        res[lib].append((d["value"], {k["kernel"]: k["avg_ms"] for k in d["kernels"]}))
for lib, v in res.items():
    print(os.path.basename(lib), [round(x[0]) for x in v], v[-1][1])
