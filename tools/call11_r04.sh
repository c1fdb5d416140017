# Round 4, experiment call: KS_XLONG_PF (ks_k_apply.h: k_apply_long<MODE, PF, XLONG>) on C3 / C4-merged.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04x
rm -rf $O; mkdir -p $O
cd $R
timeout 150 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "runs_next_to_the_sensor or close_up_long_runs" 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
timeout 260 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ab -o run -- python $R/tools/xlong_ab.py --c4 > $O/ab.log 2>&1
grep -v amdgpu.ids $O/ab.log | tail -12
cd $R
python - <<'PY'
import csv, glob, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r04x")
f = glob.glob(O + "/ab/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
out = open(O + "/ab_kernel_stats.txt", "w")
for r in rows[:40]:
    line = "%-110s calls %6s avg_us %10.1f pct %5s" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"])
    print(line); out.write(line + "\n")
PY
find $O -name "*.csv" -size +2M -delete
