#!/bin/bash
# Round 6: the tile stream (fused catch steps / rollouts, the sweep's catch tiles) with every wave streaming its own contiguous
# quarter of the tile (tools/ab/libbsuite_amd_twc.so, -DBSX_AB_TILE_WAVE_CONTIG) against every fourth KiB per wave (product)
set -u
out=$PWD/gpurun_out/r06h; mkdir -p $out
us() { python -c "
import sys,json
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l); print('%.2f' % (d['roofline']['kernel_ms']*1e3))"; }
{
for rep in 1 2 3; do for v in product twc; do
  lib=tools/ab/libbsuite_amd_$v.so; [ $v = product ] && lib=""
  line="lib=$v"
  for lanes in 131072 262144 524288; do
    e=$(BSX_NATIVE_LIB=$lib timeout 200 python bench.py --workload catch --lanes $lanes --steps 400 --warmup 40 --no-cpu-baseline --no-also 2>/dev/null | us)
    line="$line | catch $lanes: $e"
  done
  n=$(BSX_NATIVE_LIB=$lib timeout 200 python bench.py --workload catch_noise --steps 400 --warmup 40 --no-cpu-baseline --no-also 2>/dev/null | us)
  r=$(BSX_NATIVE_LIB=$lib timeout 200 python bench.py --workload catch --lanes 131072 --rollout 32 --steps 224 --warmup 32 --no-cpu-baseline --no-also 2>/dev/null | us)
  s=$(BSX_NATIVE_LIB=$lib timeout 300 python bench.py --workload sweep --sweep-schedule split --steps 100 --warmup 20 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l); print('%.1f' % (d['ms_per_step']*1e3))")
  echo "$line | catch_noise 2^20: $n | catch r32 2^17: $r | sweep split: $s us"
done; done
} > $out/ab_tile_wave_contiguous.log 2>&1
cat $out/ab_tile_wave_contiguous.log
