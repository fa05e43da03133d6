#!/bin/bash
# bash tools/ubench/h2d_check.sh <processes> <seconds> [coherent]   (GPU box; see h2d_check.hip)
n=${1:-40}; secs=${2:-60}; mode=$3
out=gpurun_out/h2d_check_${n}${mode:+_$mode}
mkdir -p $out
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/ubench/h2d_check.hip -o $out/h2d_check || exit 2
pids=()
for i in $(seq 1 $n); do
    timeout $((secs + 120)) $out/h2d_check $secs $i $mode > $out/p$i.txt 2>&1 &
    pids+=($!)
done
rcs=0
for p in "${pids[@]}"; do wait $p; rc=$?; [ $rc -gt 1 ] && rcs=$((rcs + 1)); done
echo "== $n processes, $secs s${mode:+, $mode}: $(cat $out/p*.txt | grep -c '^seed') finished, $rcs died"
cat $out/p*.txt | grep '^seed' | awk '{u += $3; k += $5; r += $11} END {print "uploads", u, "seen wrong by the kernel", k, "by the read-back", r}'
cat $out/p*.txt | grep -h 'KERNEL\|READBACK\|fault\|error' | head -12
rm -f $out/h2d_check
