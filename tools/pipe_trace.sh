# A pipelined stretch of one bench workload, kernel by kernel with the hardware queue each ran on.   sh tools/pipe_trace.sh <workload> <out dir> <tag> [env...]
W=${1:-C3}; O=$2; TAG=$3; shift 3
R=$GRAFT_REPO_ROOT
rm -rf $R/$O/pt_$TAG; mkdir -p $R/$O
cd /tmp && export TMPDIR=/tmp
env KS_PROBE_PIPE=8 "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/pt_$TAG -o run -- python $R/tools/probe_ring.py $W 8 > $R/$O/pt_$TAG.log 2>&1
cd $R
TAG=$TAG O=$O python - <<'PY'
import csv, glob, os
tag = os.environ["TAG"]; o = os.environ["O"]
f = glob.glob(f"{o}/pt_{tag}/**/run_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_points_" in r["Kernel_Name"]]
start = idx[-16]; end = idx[-12]
t0 = int(rows[start]["Start_Timestamp"])
qcol = "Queue_Id" if "Queue_Id" in rows[0] else None
scol = "Stream_Id" if "Stream_Id" in rows[0] else None
with open(f"{o}/pipe_{tag}.txt", "w") as out:
    out.write("# columns: start, duration, queue, stream, kernel   (four steady-state frames, pipelined)\n")
    for r in rows[start:end]:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")[-44:]
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        out.write(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  q{r[qcol] if qcol else '?':>3} s{r[scol] if scol else '?':>3}  {name}\n")
    ts = [int(rows[i]["Start_Timestamp"]) for i in idx[-24:-4]]
    out.write("# k_points_* to k_points_*: " + " ".join(f"{(b - a) / 1e3:.0f}" for a, b in zip(ts, ts[1:])) + " us\n")
PY
grep -v amdgpu $R/$O/pt_$TAG.log | tail -1
tail -1 $R/$O/pipe_$TAG.txt
find $R/$O/pt_$TAG -name "*.csv" -size +2M -delete
