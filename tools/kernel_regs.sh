#!/bin/bash
# register / LDS / scratch use of the kernels in one translation unit, read from the code object hipcc emits for gfx950
# usage: bash tools/kernel_regs.sh so3.hip [name filter]
set -e
SRC=$1; FILT=${2:-.}
D=$(cd $(dirname $0)/../gpr_calculator_amd/csrc && pwd)
T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 -I$D/../../include -I$D --cuda-device-only -c $D/$SRC -o $T/dev.o ${EXTRA_FLAGS}
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/dev.o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/gfx950.o
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/gfx950.o | awk '
/\.name:/ {name=$2} /\.vgpr_count:/ {v=$2} /\.agpr_count:/ {a=$2} /\.sgpr_count:/ {s=$2} /\.group_segment_fixed_size:/ {l=$2} /\.private_segment_fixed_size:/ {p=$2}
/\.vgpr_spill_count:/ {sp=$2} /\.wavefront_size:/ {printf "%-90s vgpr %3s agpr %3s sgpr %3s lds %6s scratch %5s spill %s\n", substr(name,1,90), v, a, s, l, p, sp}' | grep -E "$FILT" || true
rm -rf $T
