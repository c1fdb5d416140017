set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04f
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/pipe8 -o run -- python $R/tools/exact_tune.py C2 "pipe=8,frames=40" > $O/pipe8.log 2>&1
python $R/tools/exact_overlap.py $O/pipe8 | tee $O/pipe8_overlap.txt | tail -14
find $O -name "*.csv" -size +1M -delete
cd $R
timeout 300 python tools/exact_tune.py C2 "pipe=8" "pipe=8,KS_MARCH_STREAMS=3" "pipe=8,KS_MARCH_STREAMS=6" "pipe=8,KS_EXACT_BULK_ROUNDS=12" "pipe=8,KS_NO_TAIL_THREAD=1" 2>&1 | grep "^C2"
