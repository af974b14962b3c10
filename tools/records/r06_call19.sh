#!/bin/bash
# Round 6: the WRAPPED catch step (RewardNoise: an 18 us lane advance in front of the 32 us stream) at 2^20 lanes as ONE fused launch
# (the advance hides among the other workgroups' tile stores) against the decoupled pair; the same for the rollout
set -u
out=$PWD/gpurun_out/r06e; mkdir -p $out
export BSX_NATIVE_LIB=bsuite_amd/_lib/libbsuite_amd_tuning.so
us() { python -c "
import sys,json
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l); print('%.2f' % (d['roofline']['kernel_ms']*1e3))"; }
{
for rep in 1 2 3 4; do for mib in 128 256; do
  e=$(BSX_FUSED_TILE_MAX_MIB=$mib timeout 200 python bench.py --workload catch_noise --steps 400 --warmup 40 --no-cpu-baseline --no-also 2>/dev/null | us)
  r32=$(BSX_FUSED_ROLLOUT_MAX_MIB=$mib timeout 200 python bench.py --workload catch_noise --rollout 32 --steps 224 --warmup 32 --no-cpu-baseline --no-also 2>/dev/null | us)
  r8=$(BSX_FUSED_ROLLOUT_MAX_MIB=$mib timeout 200 python bench.py --workload catch_noise --rollout 8 --steps 224 --warmup 32 --no-cpu-baseline --no-also 2>/dev/null | us)
  echo "catch_noise/0@2^20 fused-up-to=${mib}MiB  eager $e us | rollout r32 $r32 | r8 $r8 us per step"
done; done
} > $out/ab_catch_noise_fused_at_2p20.log 2>&1
cat $out/ab_catch_noise_fused_at_2p20.log
