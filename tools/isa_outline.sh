#!/bin/bash
# Outline of one kernel's ISA: loads, vmcnt waits, barriers, LDS writes, branches, with MFMA / exp counts in between.
# usage: tools/isa_outline.sh <file.hip> <kernel-symbol-regex> [max-lines]
src=$1; sym=$2; n=${3:-200}
tmp=/tmp/isa_outline.s
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -mllvm -amdgpu-mfma-vgpr-form -Iinclude -Immgl_amd/csrc -S --cuda-device-only "$src" -o $tmp 2>/dev/null
S=$(grep -n "^_Z.*:" $tmp | grep -E "$sym" | head -1 | cut -d: -f1)
E=$(awk -v a=$S 'NR>a && /s_endpgm/ {e=NR} NR>a && /^\.Lfunc_end/ {print NR; exit}' $tmp)
echo "kernel at lines $S-$E: $(sed -n ${S}p $tmp | cut -c1-100)"
awk -v a=$S -v b=$E 'NR>=a && NR<=b' $tmp | grep -n "s_waitcnt vmcnt\|s_barrier\|global_load\|buffer_load\|global_store\|buffer_store\|ds_write\|s_cbranch\|^.LBB\|v_exp\|v_mfma\|scratch_" | awk '{ if ($2 ~ /v_mfma/) m++; else if ($2 ~ /v_exp/) e++; else if ($2 ~ /ds_write/) w++; else { if (m||e||w) { printf("      [%d mfma, %d exp, %d ds_write]\n", m, e, w); m=0; e=0; w=0 }; print $1, $2, $3, $4, $5 } }' | head -$n
