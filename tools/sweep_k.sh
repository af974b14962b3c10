#!/bin/bash
# observation stream kernel: stores per thread (K) x block size, deep_sea N=30 B=2^20
run() { python bench.py --workload $1 --steps 60 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$2', '$1', round(d['value']/1e9,4), 'Gsteps/s', round(r['achieved'],1), 'GB/s kernel_ms', round(r['kernel_ms'],4))"; }
for cfg in "256 3" "256 4" "256 5" "256 6" "256 8" "128 4" "128 8" "512 2" "512 4" "64 8" "64 16" "1024 1" "1024 2"; do
  set -- $cfg
  BSX_STREAM_BS=$1 BSX_STREAM_K=$2 run deep_sea bs$1_k$2
done
