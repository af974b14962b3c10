#!/bin/bash
set -u
out=$PWD/gpurun_out/r03; mkdir -p $out
( time timeout 1200 python -m pytest tests -m gpu -q -x -k "golden or oracle_batch or rollout or full_size or benched or live or engine or delta or logging" ) > $out/pytest_gpu_wave.log 2>&1; tail -8 $out/pytest_gpu_wave.log
T=$PWD/bsuite_amd/_lib/libbsuite_amd_tuning.so
{
for cfg in "BSX_FUSED_WAVE=0" "BSX_FUSED_WAVE=1"; do
  echo "# $cfg"
  env $cfg BSX_NATIVE_LIB=$T timeout 300 python tools/lanes_sweep.py catch -- 2**13 2**17 2**19 2**20 2>&1 | grep '^{'
  env $cfg BSX_NATIVE_LIB=$T timeout 300 python tools/lanes_sweep.py --mode rollout --T 32 --steps 256 catch -- 2**17 2**19 2**20 2>&1 | grep '^{'
done
} > $out/ab_fused_wave.log 2>&1
cat $out/ab_fused_wave.log
