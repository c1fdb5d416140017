#!/bin/bash
# Builds tools/emu/_build/libks_hip_emu.so: the WHOLE library (ks_hip.hip and every kernel header) compiled for the
# host against the stand-in <hip/hip_runtime.h> of this directory.  Same C ABI; KS_HIP_LIB=<this file> makes the Python
# binding drive it.  Test tooling (tools/emu/README.md) — not a product path, not a fallback: nothing loads it by default.
set -e
R="$(cd "$(dirname "$0")/../.." && pwd)"
CXX=/opt/rocm/lib/llvm/bin/clang++
OUT="${1:-$R/tools/emu/_build/libks_hip_emu.so}"
mkdir -p "$(dirname "$OUT")"
FLAGS="-std=c++17 -O1 -fPIC -ffp-contract=off -Wno-unused-value -Wno-unknown-attributes -DKS_EMU_BUILD -I $R/tools/emu"
cd "$R/kimera_semantics_amd/csrc"
$CXX -x c++ $FLAGS -c -o /tmp/ks_hip_emu.$$.o ks_hip.hip
$CXX $FLAGS -c -o /tmp/ks_emu_lds.$$.o "$R/tools/emu/emu_lds.cpp"
$CXX -shared -o "$OUT" /tmp/ks_hip_emu.$$.o /tmp/ks_emu_lds.$$.o -lpthread -ldl
rm -f /tmp/ks_hip_emu.$$.o /tmp/ks_emu_lds.$$.o
echo "built $OUT"
