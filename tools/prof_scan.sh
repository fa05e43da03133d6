# k_band_scan under rocprofv3 (GPU box): kernel trace + VALU / wait counters of config 4 in scan mode
R=/root/repo
cd /tmp && export TMPDIR=/tmp
export TD_OPTS=${TD_OPTS:-band_mode=1}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_scan_c4 -- python3 $R/tools/time_configs.py c4 > $R/gpurun_out/prof_scan_c4.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d $R/gpurun_out/pmc_scan_c4_VALU -- python3 $R/tools/time_configs.py c4 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc_scan_c4_MEM -- python3 $R/tools/time_configs.py c4 > $R/gpurun_out/pmc_scan_c4_MEM.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT --output-format csv -d $R/gpurun_out/pmc_scan_c4_F64 -- python3 $R/tools/time_configs.py c4 > $R/gpurun_out/pmc_scan_c4_F64.log 2>&1
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $R/gpurun_out/sq_counters.txt
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in ("prof_scan_c4",):
    for f in glob.glob("gpurun_out/%s/**/*kernel_stats.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            print(r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, "us")
for d in ("pmc_scan_c4_VALU", "pmc_scan_c4_MEM", "pmc_scan_c4_F64"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_band_scan" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(d, k, {c: sum(x) / len(x) for c, x in v.items()})
PY
