TAG=${1:-e}
R=/root/repo
cd /tmp && export TMPDIR=/tmp
for mode in fused nofuse; do
  flag=""; [ $mode = nofuse ] && flag="--no-fuse"
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_$mode -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline $flag > $R/gpurun_out/prof_${TAG}_$mode.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_${TAG}_${mode}_FETCH_SIZE -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $flag > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_${TAG}_${mode}_WRITE_SIZE -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $flag > /dev/null 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmc_${TAG}_${mode}_TCC_HIT_sum_TCC_MISS_sum -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $flag > /dev/null 2>&1
done
cd $R
python bench.py > gpurun_out/bench_${TAG}_fused.json 2> gpurun_out/bench_${TAG}_fused.err
python bench.py --no-fuse > gpurun_out/bench_${TAG}_nofuse.json 2>/dev/null
python bench.py --no-pack --no-cpu-baseline > gpurun_out/bench_${TAG}_fused_f32.json 2>/dev/null
find gpurun_out -name "*.csv" | head -40
python tools/time_configs.py c1 c2 c3 drum synth c4 > gpurun_out/configs_${TAG}.txt 2>/dev/null
python tools/hosttime_all.py > gpurun_out/hosttime_${TAG}.txt 2>/dev/null
