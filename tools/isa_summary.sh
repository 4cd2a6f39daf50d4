#!/bin/bash
# Instruction skeleton (loads, waits, MFMAs, barriers, branches; runs collapsed) of one kernel of a built object:
#   tools/isa_summary.sh eilev_amd/csrc/build/gemm.o 'gemm_rows32_kernelILi2ELi10ELi3E'
L=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin $1 $T/fat.bin && $L/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/dev.elf
$L/llvm-objdump -d $T/dev.elf | awk -v k="$2" '$0 ~ k && />:$/ {on=1; next} on && /^$/ {exit} on {print}' > $T/k.s
grep -E "s_waitcnt|v_mfma|s_barrier|s_cbranch|global_load|buffer_load|global_store|buffer_store|ds_read|ds_write|scratch_|s_endpgm" $T/k.s | sed -E 's/^\s+//; s/\s+\/\/.*$//' |
  awk '{op=$1; arg=""; if (op=="s_waitcnt") arg=$2" "$3; key=op" "arg; if (key==prev) c++; else { if (prev!="") print c"x", prev; prev=key; c=1 } } END {print c"x", prev}'
rm -rf $T
