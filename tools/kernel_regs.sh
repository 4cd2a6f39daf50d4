#!/bin/bash
# Register / scratch / LDS usage of the kernels in a built object or library:  tools/kernel_regs.sh <file.o|.so> [name-filter]
L=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$L/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$1 --output=$T/dev.elf 2>/dev/null || \
  { objcopy -O binary --only-section=.hip_fatbin $1 $T/fat.bin && $L/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/dev.elf; }
$L/llvm-readelf --notes $T/dev.elf | awk -v f="${2:-}" '
/\.agpr_count:/ {a=$2} /\.group_segment_fixed_size:/ {l=$2} /\.name:/ {n=$2} /\.private_segment_fixed_size:/ {p=$2}
/\.sgpr_count:/ {s=$2} /\.vgpr_count:/ {v=$2} /\.vgpr_spill_count:/ {sp=$2; if (f=="" || index(n,f)) printf "%-90s vgpr %3d agpr %3d sgpr %3d spill %3d scratch %5d lds %6d\n", substr(n,1,90), v, a, s, sp, p, l}'
rm -rf $T
