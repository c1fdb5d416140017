#!/bin/sh
# Compile-check of integration/server.patch (the server side): the reference's SemanticTsdfServer
# (kimera_semantics_ros/{include,src}/…/semantic_tsdf_server.{h,cpp}) with the patch applied — on-demand layer sync for a
# GPU-resident integrator: kOnDemand right after the factory call, syncLayers() at the top of updateMesh / generateMesh /
# publishPointclouds / saveMap — is COMPILED against the real Kimera-Semantics headers, the adapter's header and
# stand-ins for <ros/ros.h> / <voxblox_ros/*.h> (integration/stubs: ROS and voxblox_ros are not in this image; the
# virtuals of voxblox::TsdfServer are declared there as upstream declares them).  Patched copies live in a temporary
# directory outside the repository; nothing is linked or run.
set -e
REF="${1:-/root/reference}"
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/.." && pwd)"
TMP="$(mktemp -d /tmp/ks_server_patch.XXXXXX)"
trap 'rm -rf "$TMP"' EXIT
mkdir -p "$TMP/kimera_semantics_ros"
cp -r "$REF/kimera_semantics_ros/include" "$REF/kimera_semantics_ros/src" "$TMP/kimera_semantics_ros/"
(cd "$TMP" && patch -s -p1 < "$HERE/server.patch")
# the factory header with integration/factory.patch (the names "fast_hip" / "merged_hip")
mkdir -p "$TMP/include/kimera_semantics"
python3 - "$HERE/factory.patch" "$REF/kimera_semantics" "$TMP" <<'PY'
import subprocess, sys, re, os
patch, K, tmp = sys.argv[1:4]
for part in re.split(r'(?m)^(?=diff -ruN )', open(patch).read()):
    if not part.strip():
        continue
    rel = re.search(r'^\+\+\+ b/kimera_semantics/(\S+)', part, re.M).group(1)
    if rel.startswith('include/'):
        dst = os.path.join(tmp, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        subprocess.run(['patch', '-s', '-o', dst, os.path.join(K, rel)], input=part.encode(), check=True)
PY
INC="-I$TMP/kimera_semantics_ros/include -I$TMP/include -I$HERE/stubs -I$ROOT/oracle/ref_shim -I$ROOT/kimera_semantics_amd/compat -I$REF/kimera_semantics/include -I$ROOT/include -I$ROOT/kimera_semantics_amd/host"
${CXX:-g++} -O0 -std=c++17 -fPIC -Wall -Wno-unused-variable -Wno-deprecated-declarations -DKS_USE_REAL_KIMERA $INC -c -o "$TMP/semantic_tsdf_server.o" "$TMP/kimera_semantics_ros/src/semantic_tsdf_server.cpp"
nm -C "$TMP/semantic_tsdf_server.o" | grep -q "kimera::SemanticTsdfServer::syncDeviceMap" && echo "[integration] server.patch applies to $REF/kimera_semantics_ros and the patched SemanticTsdfServer compiles"
