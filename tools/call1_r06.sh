cd $GRAFT_REPO_ROOT
bash tools/sq_pass.sh C4-merged 3 sq_c4_merged > gpurun_out/sq_c4_merged.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x -n 4 --durations=10 2>&1 | tail -25 > gpurun_out/pytest_gpu_call1.txt
tail -5 gpurun_out/pytest_gpu_call1.txt
