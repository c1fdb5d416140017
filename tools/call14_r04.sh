# Round 4: the pipelined headline run, queue by queue (tools/pipe_view.py)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pipe_view
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O -o run -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-secondary --no-oracle-count > $O/log.txt 2>&1
cd $R
python tools/pipe_view.py $(find $O -name "*kernel_trace.csv" | head -1) 2600 > $O/view.txt
head -400 $O/view.txt
find $O -name "*.csv" -size +2M -delete
