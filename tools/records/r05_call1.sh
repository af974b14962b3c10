#!/bin/bash
# Round 5, first GPU call: parity of the new paths, then same-call A/Bs (wide rows, mnist stream, sweep schedules).
set -u
out=$PWD/gpurun_out/r05a; mkdir -p $out
B=1048576
A="--no-cpu-baseline --no-also"
T=$(python -c "from bsuite_amd import build; print(build.build(tuning=True))")
one() { python -c "
import sys,json
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l); r=d['roofline']
    print('%-34s %.4e env-steps/s  %8.2f us/step  %7.0f GB/s  frac %.3f' % (sys.argv[1], d['value'], r.get('kernel_ms', d['ms_per_step'])*1e3, r['achieved'], r['frac']))
" "$1"; }
( time timeout 900 python -m pytest -q -x -m gpu tests/test_gpu_wide_rows.py tests/test_gpu_deep_sea_single_launch.py tests/test_gpu_sweep_batch.py tests/test_gpu_logging.py tests/test_c_host.py tests/test_gpu_oracle_batch.py --durations=5 ) > $out/pytest_new_paths.log 2>&1; tail -4 $out/pytest_new_paths.log
( time timeout 900 python -m pytest -q -x -m gpu tests/test_gpu_full_size.py tests/test_gpu_benched_sizes.py --durations=5 ) > $out/pytest_sizes.log 2>&1; tail -4 $out/pytest_sizes.log
# ---- wide rows, stand-alone, 2^20 lanes: row path (advance + store stream) vs the one-launch LDS bit planes
{
for w in umbrella_length umbrella_distract memory_size; do
  for rp in off on; do
    timeout 120 python bench.py --workload $w --steps 200 --warmup 40 $A --row-path $rp 2>/dev/null | one "$w row-path=$rp"
  done
done
for k in 1 4; do
  for w in umbrella_length umbrella_distract; do
    BSX_NATIVE_LIB=$T BSX_ROW_STREAM_K=$k timeout 120 python bench.py --workload $w --steps 200 --warmup 40 $A --row-path on 2>/dev/null | one "$w row-path=on K=$k"
  done
done
for lanes in 16384 65536 262144; do
  for rp in off on; do
    timeout 120 python bench.py --workload umbrella_length --lanes $lanes --steps 400 --warmup 40 $A --row-path $rp 2>/dev/null | one "umbrella_length lanes=$lanes row-path=$rp"
    timeout 120 python bench.py --workload memory_size --lanes $lanes --steps 400 --warmup 40 $A --row-path $rp 2>/dev/null | one "memory_size lanes=$lanes row-path=$rp"
  done
done
} > $out/ab_wide_rows.log 2>&1; cat $out/ab_wide_rows.log
# ---- mnist observation stream: table-free pixel values vs the LDS table
{
for ar in 1 0; do
  BSX_NATIVE_LIB=$T BSX_MNIST_ARITH=$ar timeout 120 python bench.py --workload mnist --steps 200 --warmup 40 $A 2>/dev/null | one "mnist arith=$ar"
done
} > $out/ab_mnist_arith.log 2>&1; cat $out/ab_mnist_arith.log
# ---- the sweep: rows in the stream vs in phase 0; mnist arith; split schedule (in every line: closed / split / pipelined)
sw() { python -c "
import sys,json
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l)
    o=[k for k in ('closed','split') if k in d][0]
    print('%-40s main(%s) %.2f us frac %.3f | %s %.2f us frac %.3f | pipelined %.2f us frac %.3f' % (sys.argv[1], d['launch'][:28], d['ms_per_step']*1e3, d['roofline']['frac'], o, d[o]['ms_per_step']*1e3, d[o]['frac'], d['pipelined']['ms_per_step']*1e3, d['pipelined']['roofline']['frac']))
" "$1"; }
{
timeout 300 python bench.py --workload sweep --steps 200 --warmup 40 2>/dev/null | sw "sweep rows in stream (default)"
timeout 300 python bench.py --workload sweep --steps 200 --warmup 40 --row-path off 2>/dev/null | sw "sweep rows in phase 0 (LDS planes)"
BSX_NATIVE_LIB=$T BSX_MNIST_ARITH=0 timeout 300 python bench.py --workload sweep --steps 200 --warmup 40 2>/dev/null | sw "sweep, mnist LUT in LDS"
BSX_NATIVE_LIB=$T BSX_SPLIT_PLACE=1 timeout 300 python bench.py --workload sweep --steps 200 --warmup 40 2>/dev/null | sw "sweep, split: small families LAST"
timeout 300 python bench.py --workload sweep --steps 200 --warmup 40 2>/dev/null | sw "sweep rows in stream (default, again)"
} > $out/ab_sweep.log 2>&1; cat $out/ab_sweep.log
# ---- kernel averages of the sweep's launches (closed / split / pipelined in one run: the last 100 of each kernel belong to... see csv)
timeout 400 python tools/kernel_stats.py $out/bench_sweep_kernel_stats.csv --last 100 -- --workload sweep --steps 100 --warmup 20 > $out/kernel_stats_sweep.log 2>&1; cat $out/bench_sweep_kernel_stats.csv
timeout 300 python tools/kernel_stats.py $out/umbrella_length_rows_kernel_stats.csv -- --workload umbrella_length --steps 200 --warmup 20 $A > /dev/null 2>>$out/kernel_stats.err; cat $out/umbrella_length_rows_kernel_stats.csv
timeout 300 python tools/kernel_stats.py $out/umbrella_length_lds_kernel_stats.csv -- --workload umbrella_length --steps 200 --warmup 20 $A --row-path off > /dev/null 2>>$out/kernel_stats.err; cat $out/umbrella_length_lds_kernel_stats.csv
# ---- issue-side counters: the one-launch LDS path (before) and the row path
pm() { timeout 300 python tools/pmc.py "$@" 2>&1 | tail -1; }
pm sq umbrella_length_lds $out/umbrella_length_lds_pmc_sq.json --kernels "small_obs_kernel<umbrella_chain_env" -- --workload umbrella_length --steps 20 --warmup 4 $A --row-path off
pm sq umbrella_length_rows $out/umbrella_length_rows_pmc_sq.json --kernels "small_obs_kernel<umbrella_chain_env" "bsx_row_stream_kernel<umbrella_rows" -- --workload umbrella_length --steps 20 --warmup 4 $A
timeout 120 python tools/host_overhead.py --n 10000 > $out/host_overhead.log 2>&1; tail -22 $out/host_overhead.log
# ---- the default line
timeout 500 python bench.py > $out/bench_default.json 2> $out/bench_default.err; wc -c $out/bench_default.json; tail -3 $out/bench_default.err
ls -la $out
