#!/bin/bash
# build_flags.sh NAME "<compiler flags>": libfmi_hip variant with gemm.hip / conv.hip / attention.hip rebuilt under the given flags -> _build/libfmi_NAME.so
set -e
cd "$(dirname "$0")"
CS=../../face_mask_inpaint_amd/csrc
mkdir -p _build
name=$1; shift
FILES=${FILES:-"gemm conv attention"}
for f in $FILES; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result "$@" -c $CS/$f.hip -o _build/${f}_$name.o &
done
wait
objs=""; others=$(ls $CS/*.o)
for f in $FILES; do objs="$objs _build/${f}_$name.o"; others=$(echo "$others" | grep -v "/$f.o"); done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $others -o _build/libfmi_$name.so
ls -la _build/libfmi_$name.so
