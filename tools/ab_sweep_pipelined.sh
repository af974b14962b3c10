#!/bin/bash
# A/B inside one gpurun call: the config-5 sweep as ONE launch per step (store stream of step s beside the lane
# advance of step s+1, BSX_SWEEP_PIPELINED=1) against two launches per step.
out=$PWD/gpurun_out/ab_sweep_pipe; mkdir -p $out
line() { python -c "
import sys,json
l=[x for x in sys.stdin.read().splitlines() if x.startswith('{')]
d=json.loads(l[-1]); r=d['roofline']; print('%-14s %.3e env-steps/s  %.2f us/step  frac %.3f' % ('$1', d['value'], r['kernel_ms']*1e3, r['frac']))"; }
for rep in 1 2 3; do
  BSX_SWEEP_PIPELINED=0 timeout 200 python bench.py --workload sweep --steps 200 --warmup 20 --no-cpu-baseline --no-also 2>/dev/null | line two_launches
  for place in 0 1 2; do
    BSX_PIPELINED_PLACE=$place BSX_SWEEP_PIPELINED=1 timeout 200 python bench.py --workload sweep --steps 200 --warmup 20 --no-cpu-baseline --no-also 2>/dev/null | line pipelined_place$place
  done
done | tee $out/ab.log
