# The end-of-round run on the GPU box (gpurun -- bash tools/round_run.sh): the guard soak over the three seed ranges, the profile set
# (whose PMC summary bench.py prices its roofline line with: before the bench lines), the bench lines.
mkdir -p gpurun_out
( time python tools/fuzz_soak.py 0 20000 guard --jobs 40 ) > gpurun_out/r06_soak_a.txt 2>&1
( time python tools/fuzz_soak.py 100000 140000 guard --jobs 40 ) > gpurun_out/r06_soak_b.txt 2>&1
( time python tools/fuzz_soak.py 300000 330000 guard --jobs 40 ) > gpurun_out/r06_soak_c.txt 2>&1
bash tools/prof_round.sh r06 > gpurun_out/r06_prof_round.log 2>&1
bash tools/bench_round.sh r06 > gpurun_out/r06_bench_round.log 2>&1
tail -3 gpurun_out/r06_soak_a.txt; tail -6 gpurun_out/r06_bench_round.log
