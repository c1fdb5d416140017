# Per-dispatch durations of the stage-B chain (k_test / k_mark per phase) of the headline config, unpipelined so that the
# chain of ONE frame is seen on its own.  gpurun, repo root:  sh tools/phase_trace.sh
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/phase_trace
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o run -- python $R/bench.py --steps 4 --warmup 1 --no-pipeline --no-cpu-baseline --no-secondary --no-oracle-count > $O/log.txt 2>&1
cd $R
python - <<'PY'
import csv, glob, os, collections
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/phase_trace/**/run_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last frame: find the last k_set_params
idx = [i for i, r in enumerate(rows) if "k_set_params" in r["Kernel_Name"]]
start = idx[-6]   # a frame in the middle of the timed part
end = idx[-5]
t0 = int(rows[start]["Start_Timestamp"])
out = open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/phase_trace/one_frame.txt", "w")
for r in rows[start - 12:end]:
    name = r["Kernel_Name"].split("(")[0][-40:]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.write(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  grid {r.get('Grid_Size', r.get('Grid_Size_X', '?')):>9}  {name}\n")
out.close()
print(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/phase_trace/one_frame.txt").read())
PY
