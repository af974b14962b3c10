#!/bin/bash
# Round 6: non-temporal ACTION loads (the action column / slab is read once) vs the product
set -u
out=$PWD/gpurun_out/r06c; mkdir -p $out
{
for rep in 1 2 3; do for v in product nta; do
lib=tools/ab/libbsuite_amd_$v.so; [ $v = product ] && lib=""
for w in cartpole mountain_car bandit discounting_chain memory_len umbrella_length; do
  ns=""; [ $w = mountain_car ] && ns="--no-stagger"
  for md in "" "--rollout 16"; do
  BSX_NATIVE_LIB=$lib timeout 200 python bench.py --workload $w $md --steps 320 --warmup 32 --no-cpu-baseline --no-also $ns 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l); r=d['roofline']; print('%-18s %-12s lib=%-8s %.2f us/step' % (sys.argv[1], sys.argv[3] or 'eager', sys.argv[2], r['kernel_ms']*1e3))
" "$w" "$v" "$md"
  done
done
for w in deep_sea catch; do
  BSX_NATIVE_LIB=$lib timeout 200 python bench.py --workload $w --steps 200 --warmup 32 --no-cpu-baseline --no-also 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l); r=d['roofline']; print('%-18s %-12s lib=%-8s %.2f us/step' % (sys.argv[1], 'eager', sys.argv[2], r['kernel_ms']*1e3))
" "$w" "$v"
done
BSX_NATIVE_LIB=$lib timeout 300 python bench.py --workload sweep --steps 200 --warmup 40 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l); print('sweep              split        lib=%-8s %.2f us/step' % (sys.argv[1], d['ms_per_step']*1e3)); print('sweep              pipelined    lib=%-8s %.2f us/step' % (sys.argv[1], d['pipelined']['ms_per_step']*1e3))
" "$v"
done; done
} > $out/ab_nt_action_loads.log 2>&1
