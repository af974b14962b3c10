#!/bin/bash
set -u
out=$PWD/gpurun_out/r03; mkdir -p $out
P=$PWD/tools/ab/libbsuite_amd_prev.so
N=$PWD/tools/ab/libbsuite_amd_nohead.so
{
for lib in $P "" $N; do
  echo "# BSX_NATIVE_LIB=$lib"
  BSX_NATIVE_LIB=$lib timeout 300 python tools/lanes_sweep.py umbrella_length umbrella_distract memory_size -- 2**20 2>&1 | grep '^{'
done
} > $out/ab_wide_rows_nohead.log 2>&1
cat $out/ab_wide_rows_nohead.log
B=1048576
BSX_NATIVE_LIB=$P timeout 400 python tools/pmc.py traffic umbrella_distract_before $out/umbrella_distract_before_pmc_traffic.json --kernels small_obs_kernel --alg-bytes $((433*B)) -- --workload umbrella_distract --steps 20 --warmup 4 --no-cpu-baseline --no-also 2>&1 | tail -1
timeout 400 python tools/pmc.py traffic umbrella_distract $out/umbrella_distract_pmc_traffic.json --kernels small_obs_kernel --alg-bytes $((433*B)) -- --workload umbrella_distract --steps 20 --warmup 4 --no-cpu-baseline --no-also 2>&1 | tail -1
