#!/bin/bash
# disassembles the gfx950 code object of a kernel object file: tools/disasm.sh <lib/obj/k_*.o> <out.s>
set -e
LLVM=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$LLVM/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin "$1"
$LLVM/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/co.o
$LLVM/llvm-objdump -d --no-show-raw-insn $T/co.o > "$2"
rm -rf $T
