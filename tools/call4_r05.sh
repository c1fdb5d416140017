# (HISTORIC: a script of the round-5 search for the C4 device loop, kept as the provenance of profiles/r05_c4_fast_device_path.txt.  The switches it sets — KS_EXACT_EPOCHS, KS_EXACT_DENSE — existed only in the commits of that search; HEAD has KS_EXACT_SWEEPS / KS_EXACT_SWEEP_ORDER, see tools/call8_r05.sh, tools/call10_r05.sh.)
# Round 5, fourth GPU call: full-size C4 frames, default `fast` mode on the device — dense-iteration schedules (KS_EXACT_TRACE=1)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call4_r05
rm -rf $O; mkdir -p $O
cd $R
for v in "KS_EXACT_DENSE=3,10,14" "KS_EXACT_DENSE=2,8,8" "KS_EXACT_DENSE=2,6,6,6"; do
  env KS_EXACT_TRACE=1 $v timeout 300 python tools/c4_fast_ab.py 4 0 2>&1 | grep -v amdgpu.ids | grep "ks exact\|ms/frame" | tail -4 | cut -c1-1100 | tee -a $O/c4_trace.txt
done
sh tools/frame_trace.sh C4-fast > $O/c4_frame.log 2>&1; cp gpurun_out/frame_trace_C4-fast/one_frame.txt $O/c4_fast_one_frame.txt
grep -c . $O/c4_fast_one_frame.txt
