# One unpipelined frame of a bench workload, kernel by kernel (rocprofv3 --kernel-trace of tools/probe.py).
#   gpurun:  sh tools/frame_trace.sh C4-merged
set -x
W=${1:-C4-merged}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/frame_trace_$W
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o run -- python $R/tools/probe.py $W 3 > $O/log.txt 2>&1
cd $R
W=$W python - <<'PY'
import csv, glob, os
root = os.environ["GRAFT_REPO_ROOT"]; w = os.environ["W"]
f = glob.glob(root + f"/gpurun_out/frame_trace_{w}/**/run_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_points_" in r["Kernel_Name"]]
start, end = idx[-1], len(rows)
t0 = int(rows[start]["Start_Timestamp"])
out = open(root + f"/gpurun_out/frame_trace_{w}/one_frame.txt", "w")
for r in rows[start:end]:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[-48:]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.write(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  {name}\n")
out.close()
print(open(root + f"/gpurun_out/frame_trace_{w}/one_frame.txt").read())
PY
find $O -name "*.csv" -size +2M -delete
