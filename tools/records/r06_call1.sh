#!/bin/bash
# Round 6, first GPU call: the bisection of the mnist observation stream.
#  (a) deep_sea size=28 (784-float rows, the same geometry) through the product library, beside mnist/0 and deep_sea/10
#  (b) the bare patterns of tools/micro/mnist_stream.hip (each element of mnist's chain on/off; pipelined rounds)
set -u
out=$PWD/gpurun_out/r06a; mkdir -p $out
A="--no-cpu-baseline --no-also"
one() { python -c "
import sys,json
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l); r=d['roofline']
    print('%-34s %.4e env-steps/s  %8.2f us/step  %7.0f GB/s  frac %.3f' % (sys.argv[1], d['value'], r.get('kernel_ms', d['ms_per_step'])*1e3, r['achieved'], r['frac']))
" "$1"; }
{
for rep in 1 2; do
  for w in deep_sea_28 mnist deep_sea; do
    timeout 200 python bench.py --workload $w --steps 200 --warmup 40 $A 2>/dev/null | one "$w (rep $rep)"
  done
done
} > $out/ab_geometry_control.log 2>&1; cat $out/ab_geometry_control.log
timeout 600 tools/ab/mnist_stream > $out/mnist_stream_microbench.log 2>&1; cat $out/mnist_stream_microbench.log
timeout 300 python tools/kernel_stats.py $out/deep_sea_28_kernel_stats.csv -- --workload deep_sea_28 --steps 200 --warmup 20 $A > /dev/null 2>>$out/kernel_stats.err; cat $out/deep_sea_28_kernel_stats.csv
ls -la $out
