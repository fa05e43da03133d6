# instruction mix / occupancy counters of ONE kernel (GPU box):  bash tools/prof_kernel.sh <config of tools/time_configs.py> <kernel name substring>
# (TD_OPTS is passed through to time_configs.py, e.g. TD_OPTS=band_mode=1)
CFG=${1:-c4}; KER=${2:-k_sampsyn}
R=/root/repo
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32" "SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VALU_ADD_F64" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rm -rf $R/gpurun_out/pmc_k_$tag
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_k_$tag -- python3 $R/tools/time_configs.py $CFG > /dev/null 2>&1
done
cd $R
KER=$KER python3 - <<'PY'
import csv, glob, collections, os
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_k_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if os.environ["KER"] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    m = sum(v) / len(v)
    print("%-32s %14.0f   per 64 frames of 2880512: %.2f" % (k, m, m / (2880512 / 64.0)))
PY
