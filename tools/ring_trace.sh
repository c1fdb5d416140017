# The last frame of tools/probe_ring.py (a steady-state frame), kernel by kernel.   sh tools/ring_trace.sh <workload> <out dir> [turns]
W=${1:-C4-merged}; O=$2; T=${3:-3}
R=$GRAFT_REPO_ROOT
rm -rf $R/$O/tr_$W; mkdir -p $R/$O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tr_$W -o run -- python $R/tools/probe_ring.py $W $T > $R/$O/tr_$W.log 2>&1
cd $R
W=$W O=$O python - <<'PY'
import csv, glob, os
w = os.environ["W"]; o = os.environ["O"]
f = glob.glob(f"{o}/tr_{w}/**/run_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_points_" in r["Kernel_Name"]]
out = open(f"{o}/last_frame_{w}.txt", "w")
start, end = idx[-1], len(rows)
t0 = int(rows[start]["Start_Timestamp"])
tend = 0
for r in rows[start:end]:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[-48:]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    tend = max(tend, e)
    out.write(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  {name}\n")
out.write(f"# frame: {(tend - t0) / 1e3:.1f} us from the first kernel's start to the last kernel's end\n")
out.close()
PY
grep -v amdgpu $R/$O/tr_$W.log | tail -1
find $R/$O/tr_$W -name "*.csv" -size +2M -delete
