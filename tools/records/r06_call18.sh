#!/bin/bash
# Round 6: the fused rollouts' two modes under torch's expandable segments (hipMemCreate'd 2 MiB physical granules mapped into one
# VA range: every allocation then has the same physical granularity) against the default allocator (one hipMalloc per large tensor)
set -u
out=$PWD/gpurun_out/r06e; mkdir -p $out
{
for rep in 1 2; do for conf in default expandable; do
  if [ $conf = expandable ]; then export PYTORCH_HIP_ALLOC_CONF=expandable_segments:True PYTORCH_CUDA_ALLOC_CONF=expandable_segments:True; else unset PYTORCH_HIP_ALLOC_CONF PYTORCH_CUDA_ALLOC_CONF; fi
  echo "=== allocator: $conf (repetition $rep)"
  BSX_NATIVE_LIB=bsuite_amd/_lib/libbsuite_amd_tuning.so BSX_FUSED_ROLLOUT_MAX_MIB=256 timeout 300 python tools/rollout_alloc_modes.py catch/0 16
  timeout 300 python tools/rollout_alloc_modes.py umbrella_length/10 16
  timeout 300 python tools/rollout_alloc_modes.py memory_size/16 16
done; done
} > $out/rollout_alloc_modes_expandable.log 2>&1
grep -v amdgpu.ids $out/rollout_alloc_modes_expandable.log
