#!/bin/bash
# Throughput of the C2 fast frame under diagnostic toggles (one bench.py run each).  Needs a GPU.
B="python bench.py --steps 200 --warmup 10 --method fast --no-cpu-baseline --no-secondary --no-oracle-count"
for cfg in "X=0" "X=1" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=2" "KS_MARCH_STREAMS=1" "KS_MARCH_STREAMS=2" "KS_BENCH_GROWTH=64" "KS_BENCH_GROWTH=128" "KS_BENCH_GROWTH=16" "KS_NO_GRAPH=1" "KS_NO_TAIL_THREAD=1" "KS_NO_GRAPH=1 KS_NO_TAIL_THREAD=1"; do
  echo -n "$cfg: "
  env $cfg timeout 120 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'], d['roofline'].get('stages_us', ''))"
done
