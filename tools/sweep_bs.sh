#!/bin/bash
run() { python bench.py --workload $1 --steps 60 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$2', '$1', round(d['value']/1e9,4), 'Gsteps/s', round(r['achieved'],1), 'GB/s kernel_ms', round(r['kernel_ms'],4))"; }
for w in deep_sea; do
  for cfg in "64 16" "64 8" "64 4" "128 8" "128 4" "128 2" "256 4" "256 2" "512 2" "512 1" "512 4" "1024 1" "1024 2"; do
    set -- $cfg
    BSX_STREAM_BS=$1 BSX_STREAM_K=$2 run $w bs$1_k$2
  done
done
