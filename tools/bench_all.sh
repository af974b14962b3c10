#!/bin/bash
# One line per workload: eager step() and fused rollout(T=32) launch modes.
fmt() { python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$1', d['launch'], round(d['value']/1e9,3), 'Gsteps/s', round(r['achieved'],1), 'GB/s frac', round(r['frac'],3), 'ms/step', round(d['ms_per_step'],5))"; }
for w in deep_sea catch cartpole mountain_car bandit memory_len umbrella_length discounting_chain mnist; do
  python bench.py --workload $w --steps 128 --warmup 32 --no-cpu-baseline 2>/dev/null | fmt $w
  python bench.py --workload $w --steps 128 --warmup 32 --rollout 32 --no-cpu-baseline 2>/dev/null | fmt $w
done
