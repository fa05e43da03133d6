R=/root/repo
cd /tmp && export TMPDIR=/tmp
for mode in fused nofuse; do
  flag=""; [ $mode = nofuse ] && flag="--no-fuse"
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_e_$mode -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline $flag > $R/gpurun_out/prof_e_$mode.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_e_${mode}_FETCH_SIZE -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $flag > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_e_${mode}_WRITE_SIZE -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $flag > /dev/null 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmc_e_${mode}_TCC_HIT_sum_TCC_MISS_sum -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $flag > /dev/null 2>&1
done
cd $R
python bench.py > gpurun_out/bench_e_fused.json 2> gpurun_out/bench_e_fused.err
python bench.py --no-fuse > gpurun_out/bench_e_nofuse.json 2>/dev/null
python bench.py --no-pack --no-cpu-baseline > gpurun_out/bench_e_fused_f32.json 2>/dev/null
find gpurun_out -name "*.csv" | head -40
