#!/bin/bash
# Round 6: non-temporal observation stores in the sweep's mixed stream (one-hot bodies / mnist body / both) and in the stand-alone
# deep_sea / catch streams, vs the product — do 842 MB of ordinary stores per sweep step evict what phase 0 wants to find in cache?
set -u
out=$PWD/gpurun_out/r06c; mkdir -p $out
sw() { python -c "
import sys,json
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l)
    o=[k for k in ('closed','split') if k in d][0]
    print('%-40s main(%s) %.2f us frac %.3f | %s %.2f us frac %.3f | pipelined %.2f us frac %.3f' % (sys.argv[1], d['launch'][:28], d['ms_per_step']*1e3, d['roofline']['frac'], o, d[o]['ms_per_step']*1e3, d[o]['frac'], d['pipelined']['ms_per_step']*1e3, d['pipelined']['roofline']['frac']))
" "$1"; }
one() { python -c "
import sys,json
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l); r=d['roofline']
    print('%-40s %.4e env-steps/s  %8.2f us/step  %7.0f GB/s  frac %.3f' % (sys.argv[1], d['value'], r.get('kernel_ms', d['ms_per_step'])*1e3, r['achieved'], r['frac']))
" "$1"; }
{
for rep in 1 2 3; do
for v in product nthot ntmn ntall; do
  lib=tools/ab/libbsuite_amd_$v.so; [ $v = product ] && lib=""
  BSX_NATIVE_LIB=$lib timeout 300 python bench.py --workload sweep --steps 200 --warmup 40 2>/dev/null | sw "sweep lib=$v (rep $rep)"
done
done
for rep in 1 2; do
for v in product nthot; do
  lib=tools/ab/libbsuite_amd_$v.so; [ $v = product ] && lib=""
  for w in deep_sea catch; do
    BSX_NATIVE_LIB=$lib timeout 200 python bench.py --workload $w --steps 200 --warmup 40 --no-cpu-baseline --no-also 2>/dev/null | one "$w lib=$v (rep $rep)"
  done
done
done
} > $out/ab_nontemporal_stores.log 2>&1; cat $out/ab_nontemporal_stores.log
