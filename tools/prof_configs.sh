# The configs-3 / 4 part of tools/prof_round.sh on its own (kernel stats + VALU pass, exact and scan band modes):  bash tools/prof_configs.sh <tag>
TAG=${1:-r04}
R=/root/repo
cd /tmp && export TMPDIR=/tmp
VALU="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES"
for mode in "" scan; do
  [ "$mode" = scan ] && export TD_OPTS=band_mode=1
  for c in c3 c4; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_${c}${mode} -- python3 $R/tools/time_configs.py $c > $R/gpurun_out/prof_${TAG}_${c}${mode}.log 2>&1
    rocprofv3 --pmc $VALU --output-format csv -d $R/gpurun_out/pmc_${TAG}_${c}${mode}_VALU -- python3 $R/tools/time_configs.py $c > /dev/null 2>&1
  done
done
unset TD_OPTS
