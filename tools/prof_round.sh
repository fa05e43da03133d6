# Round profile set (run on the MI355X box through gpurun):  bash tools/prof_round.sh <tag>
#   gpurun_out/prof_<tag>_<mode>/   rocprofv3 --kernel-trace --stats of bench.py (fused = default engine, nofuse = edge-buffer model)
#   gpurun_out/pmc_<tag>_<mode>_*/  separate --pmc passes: FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum | VALU counters
#   gpurun_out/pmc_<tag>_c3_VALU, _c4_VALU   the same VALU pass for configs 3 and 4 (tools/time_configs.py); ..scan: band_mode 1; ..sep: one_grid_sources 0
# then tools/pmc_summary.py <tag> writes profiles/<tag>_pmc_summary.json and profiles/<tag>_valu.json.
TAG=${1:-r06}
R=/root/repo
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-extras"
VALU="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES"
for mode in fused nofuse; do
  flag=""; [ $mode = nofuse ] && flag="--no-fuse"
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_$mode -- python3 $R/bench.py --steps 20 --warmup 5 $B $flag > $R/gpurun_out/prof_${TAG}_$mode.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_${TAG}_${mode}_FETCH_SIZE -- python3 $R/bench.py --steps 3 --warmup 1 $B $flag > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_${TAG}_${mode}_WRITE_SIZE -- python3 $R/bench.py --steps 3 --warmup 1 $B $flag > /dev/null 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmc_${TAG}_${mode}_TCC_HIT_sum_TCC_MISS_sum -- python3 $R/bench.py --steps 3 --warmup 1 $B $flag > /dev/null 2>&1
  rocprofv3 --pmc $VALU --output-format csv -d $R/gpurun_out/pmc_${TAG}_${mode}_VALU -- python3 $R/bench.py --steps 3 --warmup 1 $B $flag > /dev/null 2>&1
done
# the batch (config 5 share) launch of the sum kernel
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_batch64 -- python3 $R/bench.py --steps 5 --warmup 2 $B --projects-per-gpu 64 > $R/gpurun_out/prof_${TAG}_batch64.log 2>&1
for c in c3 c4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_$c -- python3 $R/tools/time_configs.py $c > $R/gpurun_out/prof_${TAG}_$c.log 2>&1
  rocprofv3 --pmc $VALU --output-format csv -d $R/gpurun_out/pmc_${TAG}_${c}_VALU -- python3 $R/tools/time_configs.py $c > /dev/null 2>&1
done
# the same two configs with the tolerance-class band-pass (engine option band_mode 1: k_band_scan / k_band_chain)
export TD_OPTS=band_mode=1
for c in c3 c4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_${c}scan -- python3 $R/tools/time_configs.py $c > $R/gpurun_out/prof_${TAG}_${c}scan.log 2>&1
  rocprofv3 --pmc $VALU --output-format csv -d $R/gpurun_out/pmc_${TAG}_${c}scan_VALU -- python3 $R/tools/time_configs.py $c > /dev/null 2>&1
done
# ... under the guard (band_mode 2 + sine_mode 2, the front-end's defaults: k_band_chain<.., true>, k_sine_probe, the verdict in the launch)
export TD_OPTS=band_mode=2,sine_mode=2
for c in c3 c4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_${c}guard -- python3 $R/tools/time_configs.py $c > $R/gpurun_out/prof_${TAG}_${c}guard.log 2>&1
  rocprofv3 --pmc $VALU --output-format csv -d $R/gpurun_out/pmc_${TAG}_${c}guard_VALU -- python3 $R/tools/time_configs.py $c > /dev/null 2>&1
done
# ... and with every source family launched on its own (engine option one_grid_sources 0): k_synth / k_sampsyn / k_adsr_env by themselves
export TD_OPTS=debug.one_grid_sources=0
for c in c3 c4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_${c}sep -- python3 $R/tools/time_configs.py $c > $R/gpurun_out/prof_${TAG}_${c}sep.log 2>&1
  rocprofv3 --pmc $VALU --output-format csv -d $R/gpurun_out/pmc_${TAG}_${c}sep_VALU -- python3 $R/tools/time_configs.py $c > /dev/null 2>&1
done
unset TD_OPTS
cd $R
python tools/pmc_summary.py $TAG
