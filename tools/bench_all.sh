#!/bin/bash
# One line per workload: eager and hipGraph launch modes.
fmt() { python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$1', d['launch'], round(d['value']/1e9,3), 'Gsteps/s', round(r['achieved'],1), 'GB/s frac', round(r['frac'],3), 'of_box_store', round(r['frac_of_box_store_ceiling'],3), 'ceil', round(r['box_store_ceiling_GBps']), 'ms/step', round(d['ms_per_step'],4))"; }
for w in deep_sea catch cartpole mountain_car bandit memory_len umbrella_length discounting_chain; do
  python bench.py --workload $w --steps 128 --warmup 32 --no-cpu-baseline 2>/dev/null | fmt $w
  python bench.py --workload $w --steps 128 --warmup 32 --graph 32 --no-cpu-baseline 2>/dev/null | fmt $w
done
