#!/bin/bash
# Round 6: the mnist body of the sweep's mixed store stream — r05 guarded body (g) vs straight-line (f) vs straight-line
# without the caller-table path (a), at 4 / 6 / 8 KiB-runs per workgroup; libraries from tools/ab_flag_lib.py.
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
for v in prev g8 g4 f8 f6 pk4 a8 a4; do
  echo "== $v (rep $rep)"
  BSX_NATIVE_LIB=tools/ab/libbsuite_amd_$v.so timeout 300 python tools/sweep_stream_parts.py 2>&1 | grep "^mnist\|^deep_sea+\|^all"
done
done
} > $out/sweep_stream_parts_mnist_bodies.log 2>&1; cat $out/sweep_stream_parts_mnist_bodies.log
{
for v in prev g8 g4 f8 f6 pk4 a8 a4; do
  BSX_NATIVE_LIB=tools/ab/libbsuite_amd_$v.so timeout 300 python bench.py --workload sweep --steps 200 --warmup 40 2>/dev/null | sw "sweep lib=$v"
done
} > $out/ab_sweep_mnist_bodies.log 2>&1; cat $out/ab_sweep_mnist_bodies.log
