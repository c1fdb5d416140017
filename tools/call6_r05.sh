# (HISTORIC: a script of the round-5 search for the C4 device loop, kept as the provenance of profiles/r05_c4_fast_device_path.txt.  The switches it sets — KS_EXACT_EPOCHS, KS_EXACT_DENSE — existed only in the commits of that search; HEAD has KS_EXACT_SWEEPS / KS_EXACT_SWEEP_ORDER, see tools/call8_r05.sh, tools/call10_r05.sh.)
# Round 5, sixth GPU call: full-size C4 frames, default `fast` mode on the device: whole-ray views + sweeps (KS_EXACT_TRACE=1)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call6_r05
rm -rf $O; mkdir -p $O
cd $R
env KS_EXACT_TRACE=1 timeout 300 python tools/c4_fast_ab.py 4 0 2>&1 | grep -v amdgpu.ids | grep "ks exact\|ms/frame" | tail -4 | cut -c1-700 | tee -a $O/c4_trace.txt
sh tools/frame_trace.sh C4-fast > $O/c4_frame.log 2>&1; cp gpurun_out/frame_trace_C4-fast/one_frame.txt $O/c4_fast_one_frame.txt
grep -c . $O/c4_fast_one_frame.txt
timeout 300 python -m pytest tests/test_exact_early_out_gpu.py -m gpu -q -x -k "c4 or full_size" 2>&1 | tail -5 | tee $O/pytest_c4.txt
