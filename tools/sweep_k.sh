#!/bin/bash
run() { python bench.py --workload $1 --steps 60 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$2', '$1', round(d['value']/1e9,4), 'Gsteps/s', round(r['achieved'],1), 'GB/s kernel_ms', round(r['kernel_ms'],4))"; }
for w in deep_sea catch; do
  for k in 1 2 3 4 5 6 8 12 16; do BSX_STREAM_K=$k run $w split_k$k; done
done
