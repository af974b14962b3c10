#!/bin/bash
# Round 6: the wrapped catch step beyond 2^20 lanes — fused launch (limit raised to 4 GiB) against the decoupled pair (limit 256 MiB)
set -u
out=$PWD/gpurun_out/r06g; mkdir -p $out
export BSX_NATIVE_LIB=bsuite_amd/_lib/libbsuite_amd_tuning.so
us() { python -c "
import sys,json
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l); print('%.2f' % (d['roofline']['kernel_ms']*1e3))"; }
{
for rep in 1 2 3; do for lanes in 1572864 2097152 4194304; do
  a=$(BSX_FUSED_WRAPPED_MAX_MIB=256 timeout 200 python bench.py --workload catch_noise --lanes $lanes --steps 200 --warmup 40 --no-cpu-baseline --no-also 2>/dev/null | us)
  b=$(BSX_FUSED_WRAPPED_MAX_MIB=4096 timeout 200 python bench.py --workload catch_noise --lanes $lanes --steps 200 --warmup 40 --no-cpu-baseline --no-also 2>/dev/null | us)
  echo "catch_noise/0 eager, $lanes lanes: pair $a us | fused $b us"
done; done
} > $out/ab_catch_wrapped_fused_larger.log 2>&1
cat $out/ab_catch_wrapped_fused_larger.log
