#!/bin/bash
# Which device functions of a HIP library hold packed-FP32 arithmetic (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32)?
#   bash tools/probe_packed_fp32/scan_code_object.sh /path/to/lib.so [scratch-dir]
# Pulls the gfx950 code object out of the library's .hip_fatbin section and counts the instructions per function.
# Used on torch's librccl.so to decide how ddp.py forms the gradient average (profiles/r03_rccl_packed_fp32_functions.txt).
set -e
LIB=$1; T=${2:-$(mktemp -d)}; LLVM=/opt/rocm/lib/llvm/bin
objcopy -O binary --only-section=.hip_fatbin "$LIB" "$T/fatbin.bin"
$LLVM/clang-offload-bundler --type=o --unbundle --input="$T/fatbin.bin" --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output="$T/gfx950.co"
$LLVM/llvm-objdump -d --no-show-raw-insn "$T/gfx950.co" 2>/dev/null \
  | grep -E "^[0-9a-f]+ <|v_pk_(add|mul|fma)_f32" \
  | awk '/^[0-9a-f]+ </{f=$2; next} {c[f]++} END{for(k in c) print c[k], k}' | sort -rn
