#!/bin/bash
# A/B inside one gpurun call (boxes differ by several percent): config-5 sweep step schedules.
#   whole   BSX_FAM_SWEEP_MIXED: two launches per sweep step (phase 0 = every lane + counter bump, phase 1 = store stream)
#   pairs   mixed two-kernel group + two mixed small groups, serial / two HIP streams / HIP graph
#   r01     one group per two-kernel family + two mixed small groups as two branches of one HIP graph
for rep in 1 2 3; do
  for cfg in "whole 1 1 grouped 1" "whole_graph 1 1 grouped_graph 1" "pairs_serial 0 1 grouped 1" "pairs_streams 0 1 grouped_streams 1" \
             "pairs_graph 0 1 grouped_graph 1" "r01_graph 0 0 grouped_graph 0"; do
    set -- $cfg
    BSX_SWEEP_MIX_ALL=$2 BSX_SWEEP_MIX_PAIRS=$3 BSX_SWEEP_MODE=$4 BSX_SWEEP_PHASED=$5 BSX_SWEEP_SMALL_BESIDE=stream \
      timeout 200 python bench.py --workload sweep --steps 200 --warmup 20 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value']/1e9,3),'Gsteps/s', round(d['roofline']['kernel_ms']*1e3,1),'us/sweep-step frac', round(d['roofline']['frac'],3))"
  done
done
