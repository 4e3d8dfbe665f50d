#!/bin/bash
# registers, LDS and scratch of every kernel in a HIP object or shared library: tools/kernel_resources.sh file.o [name filter]
set -e
f=$1; pat=${2:-.}
tmp=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$tmp/fat.bin "$f" 2>/dev/null || cp "$f" $tmp/fat.bin
t=$(/opt/rocm/lib/llvm/bin/clang-offload-bundler --list --type=o --input=$tmp/fat.bin | grep gfx950 | head -1)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$tmp/fat.bin --targets=$t --output=$tmp/k.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/k.co | awk '
/\.agpr_count:/ {a=$2} /\.group_segment_fixed_size:/ {l=$2} /\.private_segment_fixed_size:/ {p=$2} /\.sgpr_count:/ {s=$2}
/\.vgpr_count:/ {v=$2} /\.vgpr_spill_count:/ {sp=$2} /\.name:/ {n=$2}
/\.wavefront_size:/ {printf "%-110s vgpr %3d agpr %3d sgpr %3d lds %6d scratch %4d spill %3d\n", n, v, a, s, l, p, sp}' | c++filt | grep -E "$pat" || true
rm -rf $tmp
