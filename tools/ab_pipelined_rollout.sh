#!/bin/bash
# A/B inside one gpurun call: software-pipelined rollouts of the two-kernel families (state_alt) against the
# kernel pair per step (BSX_ROLLOUT_PIPELINED=0) and against eager step(); BSX_PIPELINED_PLACE = where the
# advance workgroups sit in the fused grid (0 first, 1 last, 2 spread evenly).
out=$PWD/gpurun_out/ab_pipe; mkdir -p $out
line() { python -c "
import sys,json
l=[x for x in sys.stdin.read().splitlines() if x.startswith('{')]
d=json.loads(l[-1]); r=d['roofline']; print('%-22s %-9s %.3e env-steps/s  %.2f us/step  frac %.3f' % ('$1', '$2', d['value'], r['kernel_ms']*1e3, r['frac']))"; }
for rep in 1 2 3; do
  timeout 200 python bench.py --workload catch --steps 256 --warmup 32 --no-cpu-baseline --no-also 2>/dev/null | line eager catch
  BSX_ROLLOUT_PIPELINED=0 timeout 200 python bench.py --workload catch --rollout 32 --steps 256 --warmup 32 --no-cpu-baseline --no-also 2>/dev/null | line rollout32_pairs catch
  for place in 0 1 2; do
    BSX_PIPELINED_PLACE=$place timeout 200 python bench.py --workload catch --rollout 32 --steps 256 --warmup 32 --no-cpu-baseline --no-also 2>/dev/null | line rollout32_pipe_place$place catch
  done
  timeout 200 python bench.py --workload deep_sea --steps 64 --warmup 16 --no-cpu-baseline --no-also 2>/dev/null | line eager deep_sea
  BSX_ROLLOUT_PIPELINED=0 timeout 200 python bench.py --workload deep_sea --rollout 16 --steps 64 --warmup 16 --no-cpu-baseline --no-also 2>/dev/null | line rollout16_pairs deep_sea
  for place in 0 2; do
    BSX_PIPELINED_PLACE=$place timeout 200 python bench.py --workload deep_sea --rollout 16 --steps 64 --warmup 16 --no-cpu-baseline --no-also 2>/dev/null | line rollout16_pipe_place$place deep_sea
  done
done | tee $out/ab.log
