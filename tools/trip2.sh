#!/bin/bash
set -u
out=$PWD/gpurun_out/r03; mkdir -p $out
( time timeout 1700 python -m pytest tests -m gpu -q --durations=12 ) > $out/pytest_gpu.log 2>&1; tail -40 $out/pytest_gpu.log
T=$PWD/bsuite_amd/_lib/libbsuite_amd_tuning.so
{
for cfg in "BSX_FUSED_TILE_MAX_CELLS=0" "BSX_FUSED_TILE_MAX_MIB=100000 BSX_FUSED_ROLLOUT_MAX_MIB=100000"; do
  echo "# $cfg"
  env $cfg BSX_NATIVE_LIB=$T timeout 300 python tools/lanes_sweep.py catch -- 2**17 2**19 2**20 2**21 2>&1 | grep '^{'
  env $cfg BSX_NATIVE_LIB=$T timeout 300 python tools/lanes_sweep.py --mode rollout --T 32 --steps 256 catch -- 2**15 2**17 2**18 2**19 2**20 2>&1 | grep '^{'
done
} > $out/ab_fused_tile_large.log 2>&1
cat $out/ab_fused_tile_large.log
