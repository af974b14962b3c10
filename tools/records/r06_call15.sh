#!/bin/bash
# Round 6: catch at 2^20 lanes as ONE launch per step (256-lane fused tiles, write-through chunks since this round) against the
# decoupled pair (r03 with ordinary stores: box-dependent, 41.9 vs 43.5 / 44.4-45.2 vs 43.2-43.6), and the fused rollout
# (non-temporal tiles) against the pipelined one; each also as a closed loop with a device-side reader.  Tuning build, env knobs.
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
  e=$(BSX_FUSED_TILE_MAX_MIB=$mib timeout 200 python bench.py --workload catch --steps 400 --warmup 40 --no-cpu-baseline --no-also 2>/dev/null | us)
  r=$(BSX_FUSED_ROLLOUT_MAX_MIB=$mib timeout 200 python bench.py --workload catch --rollout 32 --steps 224 --warmup 32 --no-cpu-baseline --no-also 2>/dev/null | us)
  c=$(BSX_FUSED_TILE_MAX_MIB=$mib timeout 200 python examples/closed_loop_policy.py catch/0 1048576 400 2>/dev/null | tail -1 | python -c "import sys,json; print('%.1f' % (json.loads(sys.stdin.read())['ms_per_step']*1e3))")
  echo "catch/0@2^20 fused-up-to=${mib}MiB  eager $e us | rollout r32 $r us per step | closed loop with policy $c us"
done; done
} > $out/ab_catch_fused_at_2p20.log 2>&1
cat $out/ab_catch_fused_at_2p20.log
