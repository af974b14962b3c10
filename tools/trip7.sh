#!/bin/bash
set -u
out=$PWD/gpurun_out/r03; mkdir -p $out
( time timeout 1200 python -m pytest tests -m gpu -q -x -k "rollout or golden or oracle_batch or all_ids or live or mt19937 or logging or full_size" ) > $out/pytest_gpu_roll.log 2>&1; tail -4 $out/pytest_gpu_roll.log
P=$PWD/tools/ab/libbsuite_amd_prev.so
{
for lib in $P ""; do
  echo "# BSX_NATIVE_LIB=$lib"
  BSX_NATIVE_LIB=$lib timeout 300 python tools/lanes_sweep.py --mode rollout --T 16 --steps 256 cartpole mountain_car -- 2**20 2>&1 | grep '^{'
  BSX_NATIVE_LIB=$lib timeout 300 python tools/lanes_sweep.py --mode rollout --T 64 --steps 256 cartpole mountain_car -- 2**20 2>&1 | grep "^{"
  BSX_NATIVE_LIB=$lib timeout 300 python tools/lanes_sweep.py cartpole -- 2**20 2>&1 | grep "^{"
done
} > $out/ab_regs_rollout.log 2>&1
cat $out/ab_regs_rollout.log
for w in cartpole mountain_car; do
  timeout 400 python tools/pmc.py sq ${w}_rollout16 $out/${w}_rollout16_pmc_sq.json --kernels "small_obs_kernel<${w}_env, true" --last 4 -- --workload $w --rollout 16 --steps 64 --warmup 16 --no-cpu-baseline --no-also 2>&1 | tail -1
done
