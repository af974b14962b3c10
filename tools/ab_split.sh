#!/bin/bash
# A/B: fused tile writer vs split advance+stream kernels, interleaved on one box.
run() { python bench.py --workload $1 --steps 60 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$2', '$1', round(d['value']/1e9,4), 'Gsteps/s', round(d['roofline']['achieved'],1), 'GB/s kernel_ms', round(d['roofline']['kernel_ms'],4))"; }
for rep in 1 2 3; do
  BSX_DS_SPLIT=0 run deep_sea fused
  BSX_DS_SPLIT=1 run deep_sea split
  BSX_CATCH_SPLIT=0 run catch fused
  BSX_CATCH_SPLIT=1 run catch split
done
