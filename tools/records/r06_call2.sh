#!/bin/bash
# Round 6, second GPU call: the straight-line mnist observation stream in the library — parity, then same-call A/B
# against the r05 library (tools/ab/libbsuite_amd_prev.so) stand-alone and inside the sweep (8 vs 4 KiB-runs per workgroup).
set -u
out=$PWD/gpurun_out/r06b; mkdir -p $out
A="--no-cpu-baseline --no-also"
one() { python -c "
import sys,json
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l); r=d['roofline']
    print('%-40s %.4e env-steps/s  %8.2f us/step  %7.0f GB/s  frac %.3f' % (sys.argv[1], d['value'], r.get('kernel_ms', d['ms_per_step'])*1e3, r['achieved'], r['frac']))
" "$1"; }
sw() { python -c "
import sys,json
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l)
    o=[k for k in ('closed','split') if k in d][0]
    print('%-40s main(%s) %.2f us frac %.3f | %s %.2f us frac %.3f | pipelined %.2f us frac %.3f' % (sys.argv[1], d['launch'][:28], d['ms_per_step']*1e3, d['roofline']['frac'], o, d[o]['ms_per_step']*1e3, d[o]['frac'], d['pipelined']['ms_per_step']*1e3, d['pipelined']['roofline']['frac']))
" "$1"; }
( time timeout 1200 python -m pytest -q -x -m gpu tests -k "mnist or sweep or all_ids or pair" --durations=5 ) > $out/pytest_mnist_paths.log 2>&1; tail -5 $out/pytest_mnist_paths.log
{
for rep in 1 2; do
  for lib in tools/ab/libbsuite_amd_prev.so ""; do
    BSX_NATIVE_LIB=$lib timeout 200 python bench.py --workload mnist --steps 200 --warmup 40 $A 2>/dev/null | one "mnist lib=${lib:-new} (rep $rep)"
  done
done
} > $out/ab_mnist_straight_line.log 2>&1; cat $out/ab_mnist_straight_line.log
{
for rep in 1 2; do
  for lib in tools/ab/libbsuite_amd_prev.so "" tools/ab/libbsuite_amd_pk4.so; do
    BSX_NATIVE_LIB=$lib timeout 300 python bench.py --workload sweep --steps 200 --warmup 40 2>/dev/null | sw "sweep lib=${lib:-new(K8)} (rep $rep)"
  done
done
} > $out/ab_sweep_mnist_stream.log 2>&1; cat $out/ab_sweep_mnist_stream.log
timeout 300 python tools/kernel_stats.py $out/mnist_kernel_stats.csv -- --workload mnist --steps 200 --warmup 20 $A > /dev/null 2>>$out/kernel_stats.err; cat $out/mnist_kernel_stats.csv
ls -la $out
