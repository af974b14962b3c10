# eager step of the register-resident families with 2 or 4 lanes per thread (small_obs_eager2_kernel<Env, V, LPT>): bandit,
# discounting_chain and memory_len joined those families this round; r04 measured cartpole / mountain_car only (equal).
# Tuning build: BSX_EAGER2_MIN_BLOCKS (0 = one lane per thread), BSX_EAGER_LPT.
out=$PWD/gpurun_out/r05e; mkdir -p $out
A="--no-cpu-baseline --no-also"
T=$(python -c "from bsuite_amd import build; print(build.build(tuning=True))")
one() { python -c "
import sys,json
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l); r=d['roofline']
    print('%-60s %8.2f us/step  frac %.3f' % (sys.argv[1], r.get('kernel_ms', d['ms_per_step'])*1e3, r['frac']))
" "$1"; }
for lpt in 2 4; do
BSX_NATIVE_LIB=$T BSX_EAGER2_MIN_BLOCKS=1 BSX_EAGER_LPT=$lpt timeout 900 python -m pytest tests/test_gpu_oracle_batch.py tests/test_gpu_golden.py tests/test_gpu_dm_env_conformance.py tests/test_gpu_engine_features.py tests/test_gpu_full_size.py -x -q -m gpu -k "bandit or discounting or memory or cartpole or mountain_car or swingup" 2>&1 | tail -3
done
{
for rep in 1 2; do
 for cfg in "0 2" "1 2" "1 4"; do
  set -- $cfg; mb=$1; lpt=$2
  for w in bandit discounting_chain memory_len mountain_car cartpole; do
   for lanes in 1048576 524288 262144; do
    BSX_NATIVE_LIB=$T BSX_EAGER2_MIN_BLOCKS=$mb BSX_EAGER_LPT=$lpt timeout 120 python bench.py --workload $w --lanes $lanes --steps 300 --warmup 40 $A 2>/dev/null | one "$w eager, $lanes lanes, $( [ $mb = 0 ] && echo 1 || echo $lpt ) lanes per thread (rep $rep)"
   done
  done
 done
done
} > $out/ab_eager_lanes_per_thread.log 2>&1; cat $out/ab_eager_lanes_per_thread.log
