#!/bin/bash
# ab.sh <script.py> <variant> [<variant> ...]: runs the timing script with tools/bench_tools/_build/libfmi_<variant>.so in turn, twice round, on ONE box
# (box-to-box differences of +-5 % hide most kernel changes); "tree" = the in-tree library
cd "$(dirname "$0")/../.."
script=$1; shift
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = tree ]; then lib=face_mask_inpaint_amd/csrc/libfmi_hip.so; else lib=tools/bench_tools/_build/libfmi_$v.so; fi
    echo -n "$v: "; FMI_LIB_PATH=$lib timeout -k 10 300 python $script 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
