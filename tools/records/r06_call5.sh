#!/bin/bash
# Round 6: KiB-runs per workgroup of the deep_sea / catch bodies inside the sweep's mixed stream (mnist at 6), vs the product and r05
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
{
for rep in 1 2; do
for v in prev product d3 d5 d6 d8 c1 c4; do
  lib=tools/ab/libbsuite_amd_$v.so; [ $v = product ] && lib=""
  BSX_NATIVE_LIB=$lib timeout 300 python bench.py --workload sweep --steps 200 --warmup 40 2>/dev/null | sw "sweep lib=$v (rep $rep)"
done
done
} > $out/ab_sweep_deep_sea_catch_k.log 2>&1; cat $out/ab_sweep_deep_sea_catch_k.log
