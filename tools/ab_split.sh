#!/bin/bash
# A/B: fused tile writer vs split advance+stream kernels (K stores per thread), interleaved on one box.
run() { python bench.py --workload $1 --steps 60 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$2', '$1', round(d['value']/1e9,4), 'Gsteps/s', round(r['achieved'],1), 'GB/s kernel_ms', round(r['kernel_ms'],4), 'box_ceiling', round(r['box_store_ceiling_GBps']))"; }
for rep in 1 2; do
  for w in deep_sea catch; do
    BSX_DS_SPLIT=0 BSX_CATCH_SPLIT=0 run $w fused
    for k in 1 2 4 8; do BSX_DS_SPLIT=1 BSX_CATCH_SPLIT=1 BSX_STREAM_K=$k run $w split_k$k; done
  done
done
