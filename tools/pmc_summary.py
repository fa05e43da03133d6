"""Summarise rocprofv3 outputs under gpurun_out/ into profiles/<tag>_pmc_summary.json (traffic + kernel stats per bench
mode) and profiles/<tag>_valu.json (VALU counters per config), plus a copy of every kernel_stats.csv.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts a wide coalesced read stream at half its
bytes (MI355X_MICROARCH.md, HBM section) -> doubled here ("fetch_bytes_corrected").
Rows are keyed by the engine's launch FAMILY (k_sum16w<4, true> -> k_sum: the instantiation with the most time wins)."""
import collections, csv, glob, json, os, re, shutil, sys

def family(s):   # "void tdk::k_sum16w<4, true>(args)" -> ("k_sum", "tdk::k_sum16w<4, true>")
    full = re.sub(r"^void ", "", s.split("(")[0]).strip()
    base = re.sub(r"<.*$", "", full).replace("tdk::", "")
    if base.startswith("k_sum"): base = "k_sum"
    if base == "k_band_chain": base = "k_band_scan"   # (the engine's launch family: a chain is a k_band_scan launch with several stages)
    if base == "k_synth_affine": base = "k_synth"     # (the affine form of the same family)
    if base == "k_norm1": base = "k_sum"              # (the narrow single-pass Normalize is launched as the summing family)
    return base, full

def newest(pattern):
    """Only the most recent file matching the pattern (a directory may hold the output of several runs)."""
    fs = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return fs[-1:]

def counters(d):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in newest(d + "/**/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "tdk::" not in r["Kernel_Name"]: continue
            out[family(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out

def stats(d):
    out = {}
    for f in newest(d + "/**/*kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            if "tdk::" not in r["Name"]: continue
            out[family(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "total_us": float(r["TotalDurationNs"]) / 1e3,
                                      "pct": float(r["Percentage"])}
    return out

def best_per_family(rows, weight):
    """rows: {(fam, full): value}; keeps, per family, the instantiation with the largest weight(value)."""
    best = {}
    for (fam, full), v in rows.items():
        if fam not in best or weight(v) > weight(best[fam][1]):
            best[fam] = (full, v)
    return best

if __name__ == "__main__":
    tag = sys.argv[1]
    res = {}
    for mode in ("fused", "nofuse"):
        st = best_per_family(stats("gpurun_out/prof_%s_%s" % (tag, mode)), lambda v: v["total_us"])
        rows = {}
        for fam, (full, s) in st.items():
            row = dict(s, rocprof_kernel=full)
            for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum_TCC_MISS_sum"):
                for (f2, full2), v in counters("gpurun_out/pmc_%s_%s_%s" % (tag, mode, c)).items():
                    if full2 != full: continue
                    for cn, vals in v.items():
                        row[cn] = sum(vals) / len(vals)
            if "FETCH_SIZE" in row:
                row["fetch_bytes_corrected"] = int(row["FETCH_SIZE"] * 1024 * 2)
                row["write_bytes"] = int(row.get("WRITE_SIZE", 0) * 1024)
                row["hbm_side_bytes_per_launch"] = row["fetch_bytes_corrected"] + row["write_bytes"]
            if "TCC_HIT_sum" in row:
                row["l2_hit_rate"] = round(row["TCC_HIT_sum"] / max(row["TCC_HIT_sum"] + row["TCC_MISS_sum"], 1), 4)
            rows[fam] = row
        res[mode] = rows
    res["_note"] = ("per launch, averaged over the launches of one rocprofv3 pass each; hbm_side_bytes_per_launch = FETCH_SIZE*1024*2 (gfx950 "
                    "half-count correction) + WRITE_SIZE*1024: fabric-side (L2-miss) bytes, Infinity-Cache hits included")
    os.makedirs("profiles", exist_ok=True)
    json.dump(res, open("profiles/%s_pmc_summary.json" % tag, "w"), indent=1, sort_keys=True)
    valu = {}
    for cfg, d in (("config2", "gpurun_out/pmc_%s_fused_VALU" % tag), ("config2_nofuse", "gpurun_out/pmc_%s_nofuse_VALU" % tag),
                   ("config3", "gpurun_out/pmc_%s_c3_VALU" % tag), ("config4", "gpurun_out/pmc_%s_c4_VALU" % tag),
                   ("config3_scan", "gpurun_out/pmc_%s_c3scan_VALU" % tag), ("config4_scan", "gpurun_out/pmc_%s_c4scan_VALU" % tag),
                   ("config3_guard", "gpurun_out/pmc_%s_c3guard_VALU" % tag), ("config4_guard", "gpurun_out/pmc_%s_c4guard_VALU" % tag),
                   # engine option one_grid_sources 0: every source family its own launch (k_synth / k_sampsyn / k_adsr_env by themselves)
                   ("config3_separate", "gpurun_out/pmc_%s_c3sep_VALU" % tag), ("config4_separate", "gpurun_out/pmc_%s_c4sep_VALU" % tag)):
        rows = {}
        for (fam, full), v in counters(d).items():
            row = {cn: sum(vals) / len(vals) for cn, vals in v.items()}
            row["launches_seen"] = max(len(vals) for vals in v.values())
            row["rocprof_kernel"] = full
            if row.get("SQ_BUSY_CYCLES"):
                # SQ_* cycle counters tick in quad-cycles summed over the shader engines; the ratio of the two is unit-free
                row["valu_active_over_busy"] = round(row.get("SQ_ACTIVE_INST_VALU", 0.0) / row["SQ_BUSY_CYCLES"], 4)
            if fam not in rows or row.get("SQ_INSTS_VALU", 0) * row["launches_seen"] > rows[fam].get("SQ_INSTS_VALU", 0) * rows[fam]["launches_seen"]:
                rows[fam] = row
        if rows: valu[cfg] = rows
    valu["_note"] = ("rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES, one pass per config; "
                     "per-launch averages; SQ_INSTS_VALU = wave-level VALU instructions issued")
    json.dump(valu, open("profiles/%s_valu.json" % tag, "w"), indent=1, sort_keys=True)
    for d in glob.glob("gpurun_out/prof_%s_*/" % tag):
        mode = re.search(r"prof_%s_([^/]+)" % tag, d).group(1)
        for f in newest(d + "/**/*kernel_stats.csv"):
            shutil.copy(f, "profiles/%s_%s_kernel_stats.csv" % (tag, mode))
    print(json.dumps({"pmc": res, "valu": valu}, indent=1)[:6000])
