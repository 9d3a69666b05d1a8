#!/bin/bash
# usage: isa.sh <file.hip stem> <mangled-kernel-regex>  -> writes /tmp/st/k.s, prints loop summary + resource usage
cd /root/repo
mkdir -p /tmp/st
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result $EXP_FLAGS -c face_mask_inpaint_amd/csrc/$1.hip -o /tmp/st/$1.o -save-temps=obj 2>&1 | grep -E "error" -A5
S=/tmp/st/$1-hip-amdgcn-amd-amdhsa-gfx950.s
name=$(grep -E "^_Z.*:" $S | grep -E "$2" | head -1 | sed 's/:.*//')
echo "kernel: $name"
awk -v n="$name:" '$1==n{f=1} f{print} f&&/s_endpgm/{exit}' $S > /tmp/st/k.s
wc -l /tmp/st/k.s
grep -A14 "\.name: *$name\$" $S | grep -E "vgpr_count|sgpr_count|group_segment|private_segment"
grep -n "Loop Header\|s_barrier\|global_load_lds\|s_waitcnt vmcnt" /tmp/st/k.s | head -${3:-40}
echo "branches: $(grep -c s_cbranch /tmp/st/k.s)  mfma: $(grep -c v_mfma /tmp/st/k.s)"
