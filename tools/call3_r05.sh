# Round 5, third GPU call: why the device loop gives up full-size C4 frames (KS_EXACT_TRACE=1)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call3_r05
rm -rf $O; mkdir -p $O
cd $R
for v in "KS_EXACT_EPOCHS=3 KS_EXACT_DENSE=5" "KS_EXACT_EPOCHS=4 KS_EXACT_DENSE=4"; do
  env KS_EXACT_TRACE=1 $v timeout 300 python tools/c4_fast_ab.py 4 0 2>&1 | grep -v amdgpu.ids | grep "ks exact\|ms/frame" | cut -c1-900 | tee -a $O/c4_trace.txt
done
