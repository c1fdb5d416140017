#!/bin/sh
# Compile-check of the drop-in boundary against the REAL Kimera-Semantics headers and factory.
#
#   * kimera_semantics_amd/host/hip_semantic_tsdf_integrator.cpp is built with -DKS_USE_REAL_KIMERA: its bases are the
#     reference's own vxb::TsdfIntegratorBase / kimera::SemanticIntegratorBase, its types the reference's own
#     (kimera_semantics/include/kimera_semantics/*.h, read in place under $1 = /root/reference);
#   * the reference's factory (src/semantic_tsdf_integrator_factory.cpp + its header) gets integration/factory.patch
#     ("fast_hip" / "merged_hip" names and enum values) — patched copies go to a temporary directory OUTSIDE the
#     repository, the reference sources themselves are compiled where they lie;
#   * everything links into integration/_build/libkimera_semantics_hip.so together with the reference's own CPU
#     integrators, and adapter_demo is built against THAT factory (integration/_build/adapter_demo_real), so a
#     -m gpu test can drive   SemanticTsdfIntegratorFactory::create("fast_hip", ...) -> virtual -> C ABI -> HIP.
# Voxblox / Eigen / glog are not in this image: their header stand-ins are oracle/ref_shim (CPU arithmetic of the
# Voxblox base classes, which the reference's CPU integrators need) and kimera_semantics_amd/compat (types).
set -e
REF="${1:-/root/reference}"
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/.." && pwd)"
OUT="$HERE/_build"
TMP="$(mktemp -d /tmp/ks_real_kimera.XXXXXX)"
trap 'rm -rf "$TMP"' EXIT
mkdir -p "$OUT" "$TMP/include/kimera_semantics" "$TMP/src"
K="$REF/kimera_semantics"
# each half of the two-file diff is applied to its own file (patch -o writes the result outside the reference tree)
python3 - "$HERE/factory.patch" "$K" "$TMP" <<'PY'
import subprocess, sys, re, os
patch, K, tmp = sys.argv[1:4]
text = open(patch).read()
parts = re.split(r'(?m)^(?=diff -ruN )', text)
for part in parts:
    if not part.strip():
        continue
    rel = re.search(r'^\+\+\+ b/kimera_semantics/(\S+)', part, re.M).group(1)
    dst = os.path.join(tmp, rel if rel.startswith('include/') else rel)
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    subprocess.run(['patch', '-s', '-o', dst, os.path.join(K, rel)], input=part.encode(), check=True)
PY
INC="-I$TMP/include -I$ROOT/oracle/ref_shim -I$ROOT/kimera_semantics_amd/compat -I$K/include -I$ROOT -I$ROOT/include -I$ROOT/kimera_semantics_amd/host"
FLAGS="-O2 -std=c++17 -ffp-contract=off -fPIC -pthread -w -DKS_USE_REAL_KIMERA"
${CXX:-g++} $FLAGS $INC -shared -o "$OUT/libkimera_semantics_hip.so" \
  "$TMP/src/semantic_tsdf_integrator_factory.cpp" \
  "$K/src/semantic_integrator_base.cpp" "$K/src/semantic_tsdf_integrator_fast.cpp" "$K/src/semantic_tsdf_integrator_merged.cpp" \
  "$K/src/color.cpp" "$K/src/csv_iterator.cpp" \
  "$ROOT/kimera_semantics_amd/host/hip_semantic_tsdf_integrator.cpp" \
  -L"$ROOT/kimera_semantics_amd" -lks_hip -Wl,-rpath,'$ORIGIN/../../kimera_semantics_amd' -Wl,-rpath,/opt/rocm/lib
${CXX:-g++} $FLAGS -DKS_DEMO_REAL_FACTORY $INC -o "$OUT/adapter_demo_real" "$ROOT/kimera_semantics_amd/host/adapter_demo.cpp" \
  -L"$OUT" -lkimera_semantics_hip -L"$ROOT/kimera_semantics_amd" -lks_hip \
  -Wl,-rpath,'$ORIGIN' -Wl,-rpath,'$ORIGIN/../../kimera_semantics_amd' -Wl,-rpath,/opt/rocm/lib
echo "[integration] built $OUT/libkimera_semantics_hip.so + adapter_demo_real against $K/include (factory patched)"
