# Round 5, GPU call (gpurun, repo root:  bash tools/call13_r05.sh): does the finer seed (default 22 since call 12) also help the long
# rays' sweeps (C4-fast)?  KS_EXACT_SEED_GROWTH 32 / 22 / 18 / 26, four full-size frames each.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call13_r05
rm -rf $O; mkdir -p $O
cd $R
for g in 32 22 18 26; do
  env KS_EXACT_SEED_GROWTH=$g timeout 200 python tools/c4_fast_ab.py 4 0 2>&1 | grep -v amdgpu.ids | grep "ms/frame" | tail -1 | cut -c1-300 | tee -a $O/c4_seed_ab.txt
done
