#!/bin/bash
# Round 6: catch's rollout at 2^20 lanes — ONE fused launch (non-temporal tiles) against the pipelined T+1 launches; r03 had found
# the winner box-dependent with ordinary stores (36.9 vs 39.6 on one box, 43.4-45.5 vs 39.7-40.9 on another): repeat per box
set -u
out=$PWD/gpurun_out/r06e; mkdir -p $out
export BSX_NATIVE_LIB=bsuite_amd/_lib/libbsuite_amd_tuning.so
us() { python -c "
import sys,json
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l); print('%.2f' % (d['roofline']['kernel_ms']*1e3))"; }
tag=${1:-box}
{
python -c "import bench" 2>/dev/null
for rep in 1 2 3 4; do for mib in 128 256; do
  r32=$(BSX_FUSED_ROLLOUT_MAX_MIB=$mib timeout 200 python bench.py --workload catch --rollout 32 --steps 224 --warmup 32 --no-cpu-baseline --no-also 2>/dev/null | us)
  r16=$(BSX_FUSED_ROLLOUT_MAX_MIB=$mib timeout 200 python bench.py --workload catch --rollout 16 --steps 224 --warmup 32 --no-cpu-baseline --no-also 2>/dev/null | us)
  r8=$(BSX_FUSED_ROLLOUT_MAX_MIB=$mib timeout 200 python bench.py --workload catch --rollout 8 --steps 224 --warmup 32 --no-cpu-baseline --no-also 2>/dev/null | us)
  n32=$(BSX_FUSED_ROLLOUT_MAX_MIB=$mib timeout 200 python bench.py --workload catch_noise --rollout 32 --steps 224 --warmup 32 --no-cpu-baseline --no-also 2>/dev/null | us)
  echo "catch@2^20 fused-rollout-up-to=${mib}MiB  r32 $r32 | r16 $r16 | r8 $r8 us per step | catch_noise r32 $n32"
done; done
} > $out/ab_catch_fused_rollout_at_2p20_$tag.log 2>&1
cat $out/ab_catch_fused_rollout_at_2p20_$tag.log
