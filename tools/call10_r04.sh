set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04g
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -25 | tee $O/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" 2>&1 | grep -v amdgpu.ids | tail -8
