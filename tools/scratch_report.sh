#!/bin/bash
# Kernels of the built library that use scratch memory (register spills), from the code-object metadata of every csrc/*.o:
#   tools/scratch_report.sh            -> one line per kernel with private_segment_fixed_size > 0 (nothing = no spills)
L=/opt/rocm/lib/llvm/bin
D=$(dirname "$0")/../ubisoft-laforge-daft-exprt_amd/build/obj
T=$(mktemp -d)
for o in $D/*.o; do
  f=$(basename $o .o)
  $L/llvm-objcopy -O binary --only-section=.hip_fatbin $o $T/$f.fat 2>/dev/null || continue
  tgt=$($L/clang-offload-bundler --type=o --input=$T/$f.fat --list 2>/dev/null | grep gfx950) || continue
  $L/clang-offload-bundler --type=o --targets=$tgt --input=$T/$f.fat --output=$T/$f.co --unbundle
  $L/llvm-readelf --notes $T/$f.co | grep -E "^ +\.name:|private_segment_fixed_size|\.vgpr_count|agpr_count" | paste - - - - |
    awk '$0 !~ /private_segment_fixed_size: +0/' | sed 's/ \+/ /g' | cut -c1-240
done
rm -rf $T
