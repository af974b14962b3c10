#!/bin/bash
set -u
out=$PWD/gpurun_out/r03; mkdir -p $out
( time timeout 1200 python -m pytest tests -m gpu -q -x -k "golden or oracle_batch or rollout or all_ids or live or sweep_batch or mt19937" ) > $out/pytest_gpu_wide.log 2>&1; tail -4 $out/pytest_gpu_wide.log
P=$PWD/tools/ab/libbsuite_amd_prev.so
{
for lib in $P ""; do
  echo "# BSX_NATIVE_LIB=$lib"
  BSX_NATIVE_LIB=$lib timeout 300 python tools/lanes_sweep.py umbrella_length umbrella_distract memory_size -- 2**20 2>&1 | grep '^{'
  BSX_NATIVE_LIB=$lib timeout 300 python tools/lanes_sweep.py --mode rollout --T 16 --steps 64 umbrella_length umbrella_distract memory_size -- 2**20 2>&1 | grep '^{'
done
} > $out/ab_wide_rows_v4.log 2>&1
cat $out/ab_wide_rows_v4.log
for w in umbrella_distract memory_size; do
  timeout 400 python tools/pmc.py sq ${w} $out/${w}_pmc_sq.json --kernels small_obs_kernel -- --workload $w --steps 20 --warmup 4 --no-cpu-baseline --no-also 2>&1 | tail -1
done
