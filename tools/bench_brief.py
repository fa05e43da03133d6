"""Prints the interesting parts of a bench.py JSON line (file argument)."""
import json, sys
d = json.load(open(sys.argv[1]))
print("value", d["value"], "ms_per_step", d["ms_per_step"], "host", d.get("host_ms_per_step"))
for r in d.get("rooflines", []):
    print("  ", {k: v for k, v in r.items() if k in ("kernel", "avg_ms", "achieved", "peak", "frac", "ceiling_ms", "frac_of_l2_resident_ceiling", "hbm_compulsory_frac", "stream_ceiling_ms", "projects_per_launch")})
for k in ("config5", "scanned", "pcie_inclusive"):
    if d.get(k):
        print(k, {a: b for a, b in d[k].items() if a != "note"})
for c in d.get("configs", []) if isinstance(d.get("configs"), list) else []:
    print(c["config"], c["ms_per_render"], "ms", c["launches_per_render"], "launches", "host", c["host_ms_per_render"], "bound", {a: b for a, b in (c.get("bound") or {}).items() if a != "note"})
    print("    ", [(k["kernel"], k["ms_per_render"], k["launches"]) for k in c["kernels"]])
if d.get("cpu_baseline"):
    print("cpu", d["cpu_baseline"]["value"], "x", d.get("gpu_over_cpu_1thread"))
