#!/bin/bash
# ThreadSanitizer over the library's host orchestration (frame pipeline with its helper thread) on the functional model.
set -e
R="$(cd "$(dirname "$0")/../.." && pwd)"
CXX=/opt/rocm/lib/llvm/bin/clang++
OUT=/tmp/ks_tsan_pipeline
FLAGS="-std=c++17 -O1 -g -fPIC -ffp-contract=off -fsanitize=thread -Wno-unused-value -Wno-unknown-attributes -DKS_EMU_BUILD -I $R/tools/emu"
cd "$R/kimera_semantics_amd/csrc"
$CXX -x c++ $FLAGS -c -o /tmp/ks_tsan_lib.o ks_hip.hip
$CXX $FLAGS -c -o /tmp/ks_tsan_lds.o "$R/tools/emu/emu_lds.cpp"
$CXX $FLAGS -c -o /tmp/ks_tsan_main.o "$R/tools/emu/tsan_pipeline.cpp"
$CXX -fsanitize=thread -o $OUT /tmp/ks_tsan_main.o /tmp/ks_tsan_lib.o /tmp/ks_tsan_lds.o -lpthread -ldl
export TSAN_OPTIONS="${TSAN_OPTIONS:-halt_on_error=0 second_deadlock_stack=1 history_size=4}"
for args in "0 4 12" "1 4 12" "0 8 14" "0 2 8" "1 8 12"; do
  echo "== tsan_pipeline $args"
  setarch "$(uname -m)" -R $OUT $args 2>&1 | tail -${TSAN_TAIL:-40}
done
