# Round 4, third GPU call: kernel trace of the event-driven exact mode (where does a frame go?)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04c
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for CASE in "pipe=0,frames=12" "pipe=8,frames=40"; do
  T=$(echo $CASE | cut -c1-6 | tr -d '=,')
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$T -o run -- python $R/tools/exact_tune.py C2 "$CASE" > $O/$T.log 2>&1
  grep -v amdgpu.ids $O/$T.log | tail -2
  python - <<PY
import csv, glob, collections
f = glob.glob("$O/$T/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
out = open("$O/${T}_kernel_stats.txt", "w")
for r in rows[:28]:
    line = f'{r["Name"][:90]:90s} calls {int(r["Calls"]):6d} total_us {float(r["TotalDurationNs"])/1e3:10.1f} avg_us {float(r["AverageNs"])/1e3:9.2f} pct {100*float(r["TotalDurationNs"])/tot:5.1f} max_us {float(r["MaxNs"])/1e3:9.1f}'
    print(line); out.write(line + "\n")
PY
done
# one frame's timeline (unpipelined): kernel start/end relative to the frame's first kernel
python - <<PY
import csv, glob
f = glob.glob("$O/pipe0/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last frame: from the last k_points_fast on
idx = [i for i, r in enumerate(rows) if "k_points_fast" in r["Kernel_Name"]]
a = idx[-2]; b = idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
out = open("$O/one_frame_timeline.txt", "w")
for r in rows[a:b]:
    n = r["Kernel_Name"].replace("ksk::", "").replace("ksrs::", "").replace("void ", "").split("(")[0][:40]
    line = f'{(int(r["Start_Timestamp"])-t0)/1e3:9.1f} us  +{(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:8.1f} us  {n}  grid {r.get("Grid_Size_X", r.get("Grid_Size", ""))}'
    out.write(line + "\n")
print(open("$O/one_frame_timeline.txt").read()[-6000:])
PY
python $R/tools/exact_overlap.py $O/pipe8 | tee $O/pipe8_overlap.txt
find $O -name "*.csv" -size +1M -delete
cd $R
timeout 600 python -m pytest tests/test_exact_early_out_gpu.py -m gpu -q -x 2>&1 | tail -4
timeout 300 python tools/exact_tune.py C2 "pipe=0" "pipe=8" "pipe=8,KS_EXACT_BULK_ROUNDS=6" "pipe=8,KS_EXACT_BULK_ROUNDS=14" "pipe=8,KS_EXACT_SEED_GROWTH=64" 2>&1 | grep "^C2"

