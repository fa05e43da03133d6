#!/bin/bash
# Builds variants of the HIP library (kernels.hip compiled with the given -D sets) into ab/lib_<name>.so for A/B timing on the
# GPU box with TD_LIB (tools/time_configs.py, tools/ab_lib.py).   usage: tools/ab_variants.sh name1:"-DA=1 -DB=0" name2:"..." ...
set -e
cd "$(dirname "$0")/../termdaw_amd"
make -j8 >/dev/null
mkdir -p ../ab
F="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero"
for spec in "$@"; do
  name="${spec%%:*}"; defs="${spec#*:}"
  ( /opt/rocm/bin/hipcc $F $defs -c csrc/kernels.hip -o ../ab/kernels_$name.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../ab/lib_$name.so ../ab/kernels_$name.o build/engine.o build/compile.o build/devmem.o build/comm.o build/project.o build/lua_subset.o build/wav.o build/midi.o -ldl ) &
done
wait
ls -la ../ab/*.so
