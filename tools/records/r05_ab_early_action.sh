# chains' step(): the action loaded beside the state word instead of in a second, dependent round trip on the step that
# needs it (umbrella_chain: the episode's first step; memory_chain: its last).  prev = the library of the commit before.
out=$PWD/gpurun_out/r05i; mkdir -p $out
A="--no-cpu-baseline --no-also"
one() { python -c "
import sys,json
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l); r=d['roofline']
    print('%-66s %8.2f us/step  frac %.3f' % (sys.argv[1], r.get('kernel_ms', d['ms_per_step'])*1e3, r['frac']))
" "$1"; }
timeout 900 python -m pytest tests/test_gpu_oracle_batch.py tests/test_gpu_golden.py tests/test_gpu_dm_env_conformance.py tests/test_gpu_engine_features.py tests/test_gpu_full_size.py tests/test_gpu_wide_rows.py tests/test_gpu_rollout.py tests/test_gpu_vs_reference_live.py tests/test_gpu_sweep_batch.py -x -q -m gpu -k "memory or umbrella or wide or chain or sweep" 2>&1 | tail -3
{
for rep in 1 2; do
 for lib in tools/ab/libbsuite_amd_prev.so ""; do
  n=$( [ -z "$lib" ] && echo "action beside the state" || echo "action where needed" )
  for w in umbrella_length umbrella_distract memory_size memory_len; do
    BSX_NATIVE_LIB=$lib timeout 120 python bench.py --workload $w --steps 300 --warmup 40 $A 2>/dev/null | one "$w eager, $n (rep $rep)"
  done
  for w in umbrella_length memory_len; do
    BSX_NATIVE_LIB=$lib timeout 120 python bench.py --workload $w --rollout 16 --steps 320 --warmup 32 $A 2>/dev/null | one "$w r16, $n (rep $rep)"
  done
  BSX_NATIVE_LIB=$lib timeout 300 python bench.py --workload sweep --sweep-schedule split --steps 200 --warmup 40 2>/dev/null | one "sweep split, $n (rep $rep)"
 done
done
} > $out/ab_chains_early_action.log 2>&1; cat $out/ab_chains_early_action.log
