out=$PWD/gpurun_out/r05d; mkdir -p $out
A="--no-cpu-baseline --no-also"
one() { python -c "
import sys,json
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l); r=d['roofline']
    print('%-52s %8.2f us/step  frac %.3f' % (sys.argv[1], r.get('kernel_ms', d['ms_per_step'])*1e3, r['frac']))
" "$1"; }
timeout 600 python -m pytest -q -x -m gpu tests/test_gpu_oracle_batch.py tests/test_gpu_golden.py tests/test_gpu_rollout.py tests/test_gpu_deep_sea_single_launch.py tests/test_gpu_wide_rows.py -k "umbrella or memory or wide" 2>&1 | tail -3
{
for rep in 1 2; do
 for lib in tools/ab/libbsuite_amd_prev.so ""; do
  n=$( [ -z "$lib" ] && echo "time fractions from LDS" || echo "block 0 shared only" )
  for w in umbrella_length umbrella_distract memory_size; do
    BSX_NATIVE_LIB=$lib timeout 120 python bench.py --workload $w --steps 300 --warmup 40 $A 2>/dev/null | one "$w eager, $n (rep $rep)"
  done
  BSX_NATIVE_LIB=$lib timeout 120 python bench.py --workload umbrella_length --rollout 16 --steps 320 --warmup 32 $A 2>/dev/null | one "umbrella_length r16, $n (rep $rep)"
 done
done
for lib in tools/ab/libbsuite_amd_prev.so ""; do
  n=$( [ -z "$lib" ] && echo "time fractions from LDS" || echo "block 0 shared only" )
  BSX_NATIVE_LIB=$lib timeout 300 python bench.py --workload sweep --sweep-schedule split --steps 200 --warmup 40 2>/dev/null | one "sweep split, $n"
done
} > $out/ab_chain_time_fraction_table.log 2>&1; cat $out/ab_chain_time_fraction_table.log
