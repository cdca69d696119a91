#!/bin/bash
# register / LDS / spill figures of every kernel in a built object: tools/kregs.sh acc_genomics_amd/csrc/build/smem_kernel.o
L=/opt/rocm/lib/llvm/bin
t=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin "$1" $t/fat.bin
$L/clang-offload-bundler --unbundle --type=o --input=$t/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$t/dev.co
$L/llvm-readelf --notes $t/dev.co | grep -E "^ +\.name:|\.vgpr_count|\.vgpr_spill_count|\.agpr_count|group_segment_fixed_size|\.sgpr_count" | sed 's/^ *//' | paste - - - - - - | sed 's/\t/ /g'
rm -rf $t
