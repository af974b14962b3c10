#!/bin/bash
# Round 6: the 64-lane write-through tiles beyond 2^18 lanes (measured there with non-temporal chunks only: 2^19 19.3 -> 19.8) and
# at 2^20 lanes against the 256-lane tiles and the decoupled pair (lean catch step)
set -u
out=$PWD/gpurun_out/r06f; mkdir -p $out
export BSX_NATIVE_LIB=bsuite_amd/_lib/libbsuite_amd_tuning.so
us() { python -c "
import sys,json
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l); print('%.2f' % (d['roofline']['kernel_ms']*1e3))"; }
{
for rep in 1 2 3; do
  for lanes in 524288 1048576; do
    a=$(BSX_FUSED_TILE_MAX_MIB=128 timeout 200 python bench.py --workload catch --lanes $lanes --steps 400 --warmup 40 --no-cpu-baseline --no-also 2>/dev/null | us)
    b=$(BSX_FUSED_TILE_MAX_MIB=256 timeout 200 python bench.py --workload catch --lanes $lanes --steps 400 --warmup 40 --no-cpu-baseline --no-also 2>/dev/null | us)
    c=$(BSX_FUSED_TILE_MAX_MIB=256 BSX_FUSED_TILE64_MAX_LANES=1048576 timeout 200 python bench.py --workload catch --lanes $lanes --steps 400 --warmup 40 --no-cpu-baseline --no-also 2>/dev/null | us)
    echo "catch/0 lean, $lanes lanes: product rule $a us | 256-lane tiles $b | 64-lane tiles $c"
  done
done
} > $out/ab_catch_tile64_wt_larger.log 2>&1
cat $out/ab_catch_tile64_wt_larger.log
