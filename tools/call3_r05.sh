# (HISTORIC: a script of the round-5 search for the C4 device loop, kept as the provenance of profiles/r05_c4_fast_device_path.txt.  The switches it sets — KS_EXACT_EPOCHS, KS_EXACT_DENSE — existed only in the commits of that search; HEAD has KS_EXACT_SWEEPS / KS_EXACT_SWEEP_ORDER, see tools/call8_r05.sh, tools/call10_r05.sh.)
# Round 5, third GPU call: why the device loop gives up full-size C4 frames (KS_EXACT_TRACE=1)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call3_r05
rm -rf $O; mkdir -p $O
cd $R
for v in "KS_EXACT_EPOCHS=3 KS_EXACT_DENSE=5" "KS_EXACT_EPOCHS=4 KS_EXACT_DENSE=4"; do
  env KS_EXACT_TRACE=1 $v timeout 300 python tools/c4_fast_ab.py 4 0 2>&1 | grep -v amdgpu.ids | grep "ks exact\|ms/frame" | cut -c1-900 | tee -a $O/c4_trace.txt
done
