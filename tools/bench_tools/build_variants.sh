#!/bin/bash
# builds libfmi_hip variants with -DFMI_EXP=<mask> (timing experiments on the GEMM core) into tools/bench_tools/_build/
set -e
cd "$(dirname "$0")"
CS=../../face_mask_inpaint_amd/csrc
mkdir -p _build
for v in "$@"; do
  for f in gemm conv; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result $EXP_FLAGS -DFMI_EXP=$v -c $CS/$f.hip -o _build/${f}_$v.o &
  done
  wait
  others=$(ls $CS/*.o | grep -v "/gemm.o\|/conv.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC _build/gemm_$v.o _build/conv_$v.o $others -o _build/libfmi_exp_$v.so
done
ls -la _build/*.so
