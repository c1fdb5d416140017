#!/bin/sh
# Compiles the REAL reference sources (never copied: read in place from $1 = /root/reference)
# against the header shims into oracle/_ref/libks_ref.so.  Shim search order: oracle/ref_shim
# (CPU Voxblox arithmetic, test-only) before kimera_semantics_amd/compat (types, Eigen, glog).
set -e
REF="${1:-/root/reference}"
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="$ROOT/oracle/_ref"
mkdir -p "$OUT"
SRC="$REF/kimera_semantics/src"
${CXX:-g++} -O2 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -shared -pthread -w \
  -I"$HERE" -I"$ROOT/kimera_semantics_amd/compat" -I"$REF/kimera_semantics/include" \
  "$SRC/semantic_integrator_base.cpp" "$SRC/semantic_tsdf_integrator_fast.cpp" \
  "$SRC/semantic_tsdf_integrator_merged.cpp" "$SRC/semantic_tsdf_integrator_factory.cpp" \
  "$SRC/color.cpp" "$SRC/csv_iterator.cpp" "$HERE/ref_driver.cpp" \
  -o "$OUT/libks_ref.so"
echo "[oracle] built $OUT/libks_ref.so from $SRC"
