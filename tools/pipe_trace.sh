# Kernel-level view of the PIPELINED headline run: per kernel name the average duration (to compare with the
# unpipelined chain of tools/phase_trace.sh), how many kernels run concurrently, busy fraction.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pipe_trace
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o run -- env KS_BENCH_PIPE=${KS_BENCH_PIPE:-4} python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-secondary --no-oracle-count > $O/log.txt 2>&1
tail -c 600 $O/log.txt
cd $R
python - <<'PY'
import csv, glob, os, collections
root = os.environ["GRAFT_REPO_ROOT"]
f = glob.glob(root + "/gpurun_out/pipe_trace/**/run_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sp = [i for i, r in enumerate(rows) if "k_set_params" in r["Kernel_Name"]]
# timed part: the last 100 frames (5 regions x 20)
a, b = sp[-101], sp[-1]
seg = rows[a:b]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
by = collections.defaultdict(lambda: [0, 0.0])
ev = []
for r in seg:
    n = r["Kernel_Name"].split("(")[0]
    n = n.replace("void ", "")[:44]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    by[n][0] += 1; by[n][1] += (e - s) / 1e3
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
cur, last, hist = 0, t0, collections.Counter()
for t, d in ev:
    hist[cur] += t - last
    last = t; cur += d
tot = t1 - t0
print(f"window {tot/1e6:.2f} ms for 100 frames -> {tot/1e5:.1f} us/frame; sum of kernel time {sum(v[1] for v in by.values())/100:.1f} us/frame")
print("concurrency:", {k: round(v / tot, 3) for k, v in sorted(hist.items())})
for n, (c, us) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:46s} calls/frame {c/100:5.2f}  avg {us/c:7.1f} us   per frame {us/100:7.1f} us")
PY
