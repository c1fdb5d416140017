cd $GRAFT_REPO_ROOT
O=gpurun_out/call5; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_apply_runs_gpu.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -4 > $O/pytest_runs.txt; tail -2 $O/pytest_runs.txt
timeout 600 python -m pytest tests/test_hip_vs_ref_gpu.py tests/test_parity_gpu.py -m gpu -q -x -n 4 -k "full_size_c4 or merged_bit_exact or close_up or long or xlong or sensor" 2>&1 | tail -5 > $O/pytest_c4.txt; tail -2 $O/pytest_c4.txt
python - <<'PY' > $O/trace_probe.py
PY
cat > /tmp/probe_ring.py <<'PY'
import sys; sys.path.insert(0, '.')
import bench
from kimera_semantics_amd import binding as B
name = sys.argv[1]; turns = int(sys.argv[2])
wl = bench.WORKLOADS[name]
frames = bench.make_frames(wl, range(6))
h = B.HipIntegrator(B.default_config(max_tiles=1 << 16 if name.startswith("C4") else 1 << 13, max_points=wl["w"] * wl["h"], **bench.integ_cfg(wl)))
for t in range(turns):
    for f in frames:
        h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
print(name, h.update_stats())
h.close()
PY
cp /tmp/probe_ring.py $O/probe_ring.py
cd /tmp && export TMPDIR=/tmp
for W in C4-merged C3; do
  rm -rf $GRAFT_REPO_ROOT/$O/tr_$W
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr_$W -o run -- python $GRAFT_REPO_ROOT/$O/probe_ring.py $W 3 > $GRAFT_REPO_ROOT/$O/tr_$W.log 2>&1
done
cd $GRAFT_REPO_ROOT
for W in C4-merged C3; do
W=$W O=$O python - <<'PY'
import csv, glob, os
w = os.environ["W"]; o = os.environ["O"]
f = glob.glob(f"{o}/tr_{w}/**/run_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_points_" in r["Kernel_Name"]]
out = open(f"{o}/last_frame_{w}.txt", "w")
for start, end in ((idx[-1], len(rows)),):
    t0 = int(rows[start]["Start_Timestamp"])
    for r in rows[start:end]:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")[-48:]
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        out.write(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  {name}\n")
out.close()
print(open(f"{o}/last_frame_{w}.txt").read())
PY
done > $O/last_frames.txt
grep -h "k_apply\|k_find_long\|k_xl\|^C" $O/last_frames.txt $O/tr_*.log | grep -v amdgpu | head -40
find $O -name "*.csv" -size +2M -delete
bash tools/sq_pass.sh C4-merged 3 sq_c4_merged_runs3 > $O/sq.log 2>&1
