#!/bin/bash
# Round 6: where do the fused rollouts' two modes come from?  (tools/rollout_alloc_modes.py)  $1 = output tag
set -u
out=$PWD/gpurun_out/r06e; mkdir -p $out
tag=${1:-1}
{
BSX_NATIVE_LIB=bsuite_amd/_lib/libbsuite_amd_tuning.so BSX_FUSED_ROLLOUT_MAX_MIB=256 timeout 300 python tools/rollout_alloc_modes.py catch/0 16
echo "--- product library from here (catch, deep_sea: the pipelined rollout)"
for w in "umbrella_length/10 16" "cartpole/0 16" "mountain_car/0 16" "bandit/0 16" "discounting_chain/0 16" "memory_len/10 16" "catch/0 32" "deep_sea/10 16"; do
  timeout 300 python tools/rollout_alloc_modes.py $w
done
} > $out/rollout_alloc_modes_$tag.log 2>&1
cat $out/rollout_alloc_modes_$tag.log
