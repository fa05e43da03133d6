#!/bin/bash
# The round's bench lines (GPU box): bash tools/bench_round.sh <tag>  ->  gpurun_out/<tag>_bench{_k20,,_full,_nofuse,_batch64}.json
T=${1:-r06}
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_k20.json 2> gpurun_out/${T}_b1.err
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_b2.err
python bench.py --full --steps 200 --warmup 20 > gpurun_out/${T}_bench_full.json 2> gpurun_out/${T}_b3.err
python bench.py --no-fuse --no-extras --steps 200 --warmup 20 > gpurun_out/${T}_bench_nofuse.json 2> gpurun_out/${T}_b4.err
python bench.py --projects-per-gpu 64 --steps 20 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/${T}_bench_batch64.json 2> gpurun_out/${T}_b5.err
for f in _k20 "" _full _nofuse _batch64; do
  python3 -c "import json,sys; j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], j['value'], j['ms_per_step'], j['roofline']['frac'])" gpurun_out/${T}_bench$f.json
done
