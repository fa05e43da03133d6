cd $GRAFT_REPO_ROOT
for i in $(seq 14); do
  for s in 6832 13751 9275; do
    ( python tools/fuzz_soak.py $s $((s+1)) guard > gpurun_out/repro_${s}_$i.txt 2>&1 ) &
  done
done
wait
grep -L "mismatching: \[\]" gpurun_out/repro_*.txt | head -20
echo ---
cat $(grep -L "mismatching: \[\]" gpurun_out/repro_*.txt | head -3) | tail -12
