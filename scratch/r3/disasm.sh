#!/bin/bash
# disasm.sh <family-number> '<demangled kernel regex>' [out]  -- ISA of one kernel out of build/vecsim_family_<n>.o
set -e
B=/opt/rocm/lib/llvm/bin
T=${TMPDIR:-/tmp}/vs_disasm; mkdir -p $T
OBJ=${VS_OBJ_DIR:-/root/repo/simurlacra_amd/csrc/build}/vecsim_family_$1.o
$B/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin $OBJ
$B/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co
SYM=$($B/llvm-readelf -s -W $T/dev.co | awk '$4=="FUNC"{print $8}' | sort -u | while read m; do echo "$m $(echo $m | c++filt)"; done | grep -E "$2" | head -1 | cut -d' ' -f1)
echo "symbol: $SYM" >&2
$B/llvm-objdump -d --no-show-raw-insn --disassemble-symbols=$SYM $T/dev.co > ${3:-$T/k.s}
wc -l ${3:-$T/k.s} >&2
