cd $GRAFT_REPO_ROOT
O=gpurun_out/call3; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_apply_runs_gpu.py -m gpu -q -x -s 2>&1 | grep -v amdgpu.ids | tail -14 > $O/pytest_runs.txt; tail -4 $O/pytest_runs.txt
timeout 600 python -m pytest tests/test_hip_vs_ref_gpu.py tests/test_parity_gpu.py -m gpu -q -x -n 4 -k "full_size_c4 or merged_bit_exact or close_up or long or xlong or sensor" 2>&1 | tail -5 > $O/pytest_c4.txt; tail -2 $O/pytest_c4.txt
for W in C4-merged C3; do
  sh tools/frame_trace.sh $W > $O/frame_$W.log 2>&1; cp gpurun_out/frame_trace_$W/one_frame.txt $O/one_frame_$W.txt
done
KS_BENCH_GROWTH=32 sh tools/frame_trace.sh C4-fast > $O/frame_C4-fast-ordered.log 2>&1; cp gpurun_out/frame_trace_C4-fast/one_frame.txt $O/one_frame_C4-fast-ordered.txt
KS_DEBUG=1 KS_APPLY_RUNS=1 sh tools/frame_trace.sh C2 > $O/frame_runs_C2.log 2>&1; cp gpurun_out/frame_trace_C2/one_frame.txt $O/one_frame_runs_C2.txt
bash tools/sq_pass.sh C4-merged 3 sq_c4_merged_runs2 > $O/sq.log 2>&1
python - <<'PY' > $O/xl_stats.txt 2>&1
import sys; sys.path.insert(0, '.')
import bench
from kimera_semantics_amd import binding as B
for name in ("C4-merged", "C3"):
    wl = bench.WORKLOADS[name]
    frames = bench.make_frames(wl, range(6))
    h = B.HipIntegrator(B.default_config(max_tiles=1 << 16 if name.startswith("C4") else 1 << 13, max_points=wl["w"] * wl["h"], **bench.integ_cfg(wl)))
    for f in frames:
        h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        print(name, h.update_stats(), flush=True)
    h.close()
PY
cat $O/xl_stats.txt | grep -v amdgpu
grep -h "k_apply\|k_find_long\|k_xl" $O/one_frame_*.txt | head -40
