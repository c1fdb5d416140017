cd $GRAFT_REPO_ROOT
O=gpurun_out/call4; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_apply_runs_gpu.py -m gpu -q -x -s 2>&1 | grep -v amdgpu.ids | tail -14 > $O/pytest_runs.txt; tail -4 $O/pytest_runs.txt
timeout 600 python -m pytest tests/test_hip_vs_ref_gpu.py tests/test_parity_gpu.py -m gpu -q -x -n 4 -k "full_size_c4 or merged_bit_exact or close_up or long or xlong or sensor" 2>&1 | tail -5 > $O/pytest_c4.txt; tail -2 $O/pytest_c4.txt
for W in C4-merged C3; do
  sh tools/frame_trace.sh $W > $O/frame_$W.log 2>&1; cp gpurun_out/frame_trace_$W/one_frame.txt $O/one_frame_$W.txt
done
python - <<'PY' > $O/xl_stats.txt 2>&1
import sys, time; sys.path.insert(0, '.')
import bench
from kimera_semantics_amd import binding as B
for name, turns in (("C4-merged", 3), ("C3", 3)):
    wl = bench.WORKLOADS[name]
    frames = bench.make_frames(wl, range(8))
    h = B.HipIntegrator(B.default_config(max_tiles=1 << 16 if name.startswith("C4") else 1 << 13, max_points=wl["w"] * wl["h"], **bench.integ_cfg(wl)))
    prev = dict(walked=0, serial=0, chunks=0, replayed=0)
    for t in range(turns):
        for k, f in enumerate(frames):
            t0 = time.time()
            h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
            dt = time.time() - t0
            st = h.update_stats()
            print(name, "turn", t, "frame", k, f"{dt*1e3:.2f} ms", {q: st[q] - prev[q] for q in st}, flush=True)
            prev = st
    h.close()
PY
grep -v amdgpu $O/xl_stats.txt
grep -h "k_apply\|k_find_long\|k_xl" $O/one_frame_*.txt | head -40
