#!/bin/bash
# Round 5, GPU call 2: the flat-plane form of the wide-row path — parity, then the same A/Bs as call 1.
set -u
out=$PWD/gpurun_out/r05b; mkdir -p $out
A="--no-cpu-baseline --no-also"
T=$(python -c "from bsuite_amd import build; print(build.build(tuning=True))")
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
( time timeout 900 python -m pytest -q -x -m gpu tests/test_gpu_wide_rows.py tests/test_gpu_sweep_batch.py tests/test_c_host.py --durations=5 ) > $out/pytest_new_paths.log 2>&1; tail -4 $out/pytest_new_paths.log
{
for w in umbrella_length umbrella_distract memory_size; do
  for rp in off on; do
    timeout 120 python bench.py --workload $w --steps 200 --warmup 40 $A --row-path $rp 2>/dev/null | one "$w row-path=$rp"
  done
done
for k in 1 4; do
  for w in umbrella_length umbrella_distract memory_size; do
    BSX_NATIVE_LIB=$T BSX_ROW_STREAM_K=$k timeout 120 python bench.py --workload $w --steps 200 --warmup 40 $A --row-path on 2>/dev/null | one "$w row-path=on K=$k"
  done
done
for lanes in 16384 65536 262144; do
  for rp in off on; do
    timeout 120 python bench.py --workload umbrella_length --lanes $lanes --steps 400 --warmup 40 $A --row-path $rp 2>/dev/null | one "umbrella_length lanes=$lanes row-path=$rp"
    timeout 120 python bench.py --workload umbrella_distract --lanes $lanes --steps 400 --warmup 40 $A --row-path $rp 2>/dev/null | one "umbrella_distract lanes=$lanes row-path=$rp"
    timeout 120 python bench.py --workload memory_size --lanes $lanes --steps 400 --warmup 40 $A --row-path $rp 2>/dev/null | one "memory_size lanes=$lanes row-path=$rp"
  done
done
} > $out/ab_wide_rows.log 2>&1; cat $out/ab_wide_rows.log
{
timeout 300 python bench.py --workload sweep --steps 200 --warmup 40 2>/dev/null | sw "sweep rows in stream (default)"
timeout 300 python bench.py --workload sweep --steps 200 --warmup 40 --row-path off 2>/dev/null | sw "sweep rows in phase 0 (LDS planes)"
BSX_NATIVE_LIB=$T BSX_SPLIT_PLACE=2 timeout 300 python bench.py --workload sweep --steps 200 --warmup 40 2>/dev/null | sw "sweep, split: small families SPREAD"
timeout 300 python bench.py --workload sweep --steps 200 --warmup 40 2>/dev/null | sw "sweep rows in stream (default, again)"
timeout 300 python bench.py --workload sweep --steps 200 --warmup 40 --row-path off 2>/dev/null | sw "sweep rows in phase 0 (again)"
} > $out/ab_sweep.log 2>&1; cat $out/ab_sweep.log
for rp in auto off; do
  timeout 400 python tools/kernel_stats.py $out/bench_sweep_kernel_stats_rows_$rp.csv --last 1000 -- --workload sweep --steps 100 --warmup 20 --row-path $rp > /dev/null 2>&1; cat $out/bench_sweep_kernel_stats_rows_$rp.csv
done
timeout 300 python tools/kernel_stats.py $out/umbrella_length_rows_kernel_stats.csv -- --workload umbrella_length --steps 200 --warmup 20 $A > /dev/null 2>>$out/kernel_stats.err; cat $out/umbrella_length_rows_kernel_stats.csv
timeout 300 python tools/kernel_stats.py $out/memory_size_rows_kernel_stats.csv -- --workload memory_size --steps 200 --warmup 20 $A > /dev/null 2>>$out/kernel_stats.err; cat $out/memory_size_rows_kernel_stats.csv
pm() { timeout 300 python tools/pmc.py "$@" 2>&1 | tail -1; }
pm sq umbrella_length_rows $out/umbrella_length_rows_pmc_sq.json --kernels "small_obs_kernel<umbrella_chain_env" "bsx_row_stream_kernel<umbrella_rows" -- --workload umbrella_length --steps 20 --warmup 4 $A
ls -la $out
