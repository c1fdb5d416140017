# Round 5, eighth GPU call: full-size C4 frames, default `fast` mode: filtered sweeps + a confirming full one (KS_EXACT_TRACE=1)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call8_r05
rm -rf $O; mkdir -p $O
cd $R
for v in KS_EXACT_SWEEP_ORDER=1 KS_EXACT_SWEEP_ORDER=0; do
  env KS_EXACT_TRACE=1 $v timeout 300 python tools/c4_fast_ab.py 4 0 2>&1 | grep -v amdgpu.ids | grep "ks exact\|ms/frame" | tail -3 | cut -c1-700 | tee -a $O/c4_trace.txt
done
sh tools/frame_trace.sh C4-fast > $O/c4_frame.log 2>&1; cp gpurun_out/frame_trace_C4-fast/one_frame.txt $O/c4_fast_one_frame.txt
timeout 400 python -m pytest tests/test_exact_early_out_gpu.py -m gpu -q -x -k "c4 or full_size" 2>&1 | tail -5 | tee $O/pytest_c4.txt
