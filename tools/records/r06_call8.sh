#!/bin/bash
# Round 6: (a) non-temporal stores in the stream half of the pipelined rollouts of catch / deep_sea (tools/ab pnt) vs the product;
# (b) the split step's first-launch size re-swept under the non-temporal mixed stream (tuning build, BSX_SPLIT_ROUND)
set -u
out=$PWD/gpurun_out/r06c; mkdir -p $out
A="--no-cpu-baseline --no-also"
one() { python -c "
import sys,json
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l); r=d['roofline']
    print('%-44s %.4e env-steps/s  %8.2f us/step  %7.0f GB/s  frac %.3f' % (sys.argv[1], d['value'], r.get('kernel_ms', d['ms_per_step'])*1e3, r['achieved'], r['frac']))
" "$1"; }
sw() { python -c "
import sys,json
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l)
    o=[k for k in ('closed','split') if k in d][0]
    print('%-40s main(%s) %.2f us frac %.3f | %s %.2f us frac %.3f | pipelined %.2f us frac %.3f' % (sys.argv[1], d['launch'][:28], d['ms_per_step']*1e3, d['roofline']['frac'], o, d[o]['ms_per_step']*1e3, d[o]['frac'], d['pipelined']['ms_per_step']*1e3, d['pipelined']['roofline']['frac']))
" "$1"; }
{
for rep in 1 2 3; do
for v in product pnt; do
  lib=tools/ab/libbsuite_amd_$v.so; [ $v = product ] && lib=""
  BSX_NATIVE_LIB=$lib timeout 200 python bench.py --workload catch --rollout 32 --steps 224 --warmup 32 $A 2>/dev/null | one "catch r32 lib=$v (rep $rep)"
  BSX_NATIVE_LIB=$lib timeout 200 python bench.py --workload deep_sea --rollout 16 --steps 48 --warmup 16 $A 2>/dev/null | one "deep_sea r16 lib=$v (rep $rep)"
done
done
} > $out/ab_pipelined_rollout_nt.log 2>&1; cat $out/ab_pipelined_rollout_nt.log
T=bsuite_amd/_lib/libbsuite_amd_tuning.so
{
for rep in 1 2; do
for r in -1 0 1200 2048 2416 3000 3600; do
  BSX_NATIVE_LIB=$T BSX_SPLIT_ROUND=$r timeout 300 python bench.py --workload sweep --sweep-schedule split --steps 200 --warmup 40 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l); print('sweep split round=%-6s %.2f us frac %.3f' % (sys.argv[1], d['ms_per_step']*1e3, d['roofline']['frac']))
" "$r"
done
done
} > $out/ab_sweep_split_round_nt.log 2>&1; cat $out/ab_sweep_split_round_nt.log
