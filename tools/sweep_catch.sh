#!/bin/bash
run() { python bench.py --workload catch --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$1', round(d['value']/1e9,3), 'Gsteps/s', round(r['achieved'],1), 'GB/s kernel_ms', round(r['kernel_ms'],4))"; }
for cfg in "64 4" "64 8" "128 2" "128 4" "256 1" "256 2" "256 3" "512 1" "512 2" "1024 1"; do
  set -- $cfg
  BSX_STREAM_BS=$1 BSX_STREAM_K=$2 run bs$1_k$2
done
