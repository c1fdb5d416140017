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
# EMU_SAN=address: AddressSanitizer over host AND device code (device memory is heap memory here: an out-of-bounds
# access of a kernel is reported like any other).  Run with
#   LD_PRELOAD=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0
if [ -n "$EMU_SAN" ]; then FLAGS="$FLAGS -g -fsanitize=$EMU_SAN -shared-libsan -fno-omit-frame-pointer"; fi
cd "$R/kimera_semantics_amd/csrc"
# One build at a time per output (pytest-xdist workers all ask for it), nothing to do when the library is newer than every
# source it is made of, and the output appears ATOMICALLY (a reader never maps a half-written file).
exec 9>"$OUT.lock"
flock 9
if [ -z "$EMU_SAN" ] && [ -f "$OUT" ] && [ -z "$(find "$R/kimera_semantics_amd/csrc" "$R/tools/emu" "$R/include" -maxdepth 2 -newer "$OUT" \( -name '*.h' -o -name '*.hip' -o -name '*.cpp' -o -name '*.sh' \) -print -quit)" ]; then
  echo "up to date: $OUT"
  exit 0
fi
$CXX -x c++ $FLAGS -c -o /tmp/ks_hip_emu.$$.o ks_hip.hip
$CXX $FLAGS -c -o /tmp/ks_emu_lds.$$.o "$R/tools/emu/emu_lds.cpp"
$CXX -shared ${EMU_SAN:+-fsanitize=$EMU_SAN -shared-libsan} -o "$OUT.$$.tmp" /tmp/ks_hip_emu.$$.o /tmp/ks_emu_lds.$$.o -lpthread -ldl
mv -f "$OUT.$$.tmp" "$OUT"
rm -f /tmp/ks_hip_emu.$$.o /tmp/ks_emu_lds.$$.o
echo "built $OUT"
