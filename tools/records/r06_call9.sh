#!/bin/bash
# Round 6: non-temporal OUTPUT stores of the small-observation families (BSX_SMALL_NT bits: 1 scalars, 2 rows of 1-2 floats,
# 4 wave-staged 16-byte row chunks), eager and fused rollout, vs the product
set -u
out=$PWD/gpurun_out/r06c; mkdir -p $out
{
for rep in 1 2 3; do for v in snt0 product; do
lib=tools/ab/libbsuite_amd_$v.so; [ $v = product ] && lib=""
for w in cartpole mountain_car bandit discounting_chain memory_len umbrella_length; do
  ns=""; [ $w = mountain_car ] && ns="--no-stagger"
  for md in "" "--rollout 16"; do
  BSX_NATIVE_LIB=$lib timeout 200 python bench.py --workload $w $md --steps 320 --warmup 32 --no-cpu-baseline --no-also $ns 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l); r=d['roofline']; print('%-18s %-12s lib=%-8s %.2f us/step frac %.3f' % (sys.argv[1], sys.argv[3] or 'eager', sys.argv[2], r['kernel_ms']*1e3, r.get('hbm',r)['frac'] if 'hbm' in r else r['frac']))
" "$w" "$v" "$md"
  done
done; done; done
} > $out/ab_small_families_nt_final.log 2>&1; cat $out/ab_small_families_nt_final.log
