#!/bin/bash
# A/B inside one gpurun call: config-5 sweep step.
#   per_family / mixed : one grouped launch per (family, tile class) vs the small families merged
#   grouped / grouped_graph:N : serial launches vs N concurrent branches of one HIP graph
#   map0 / map1 : workgroup -> segment by binary search vs by the per-workgroup map
for rep in 1 2; do
  for cfg in per_family:grouped:1:0 mixed:grouped:1:0 mixed:grouped:1:1 mixed:grouped_graph:2:0 mixed:grouped_graph:2:1 mixed:grouped_graph:3:1; do
    IFS=: read mix m st map <<< "$cfg"
    [ "$mix" = mixed ] && mx=1 || mx=0
    BSX_GROUP_MAP=$map BSX_SWEEP_MIX_SMALL=$mx BSX_SWEEP_MODE=$m BSX_SWEEP_STREAMS=$st timeout 200 python bench.py --workload sweep --steps 200 --warmup 20 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$mix $m streams=$st map=$map', round(d['value']/1e9,3),'Gsteps/s', round(d['roofline']['kernel_ms'],4),'ms/sweep-step frac', round(d['roofline']['frac'],3))"
  done
done
