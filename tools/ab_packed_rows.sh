#!/bin/bash
# A/B inside one gpurun call: packed observation records (this tree) against a baseline build
# (bsuite_amd/_lib_base/libbsuite_amd.so, built from the previous commit).  Sweep step, the families whose
# rows were f32 LDS tiles, and the ones that moved from the tile to per-thread stores in non-lean calls.
out=$PWD/gpurun_out/ab_packed; mkdir -p $out
base=$PWD/bsuite_amd/_lib_base/libbsuite_amd.so
line() { python -c "
import sys,json
l=[x for x in sys.stdin.read().splitlines() if x.startswith('{')]
d=json.loads(l[-1]); r=d['roofline']; print('%-9s %-18s %.3e env-steps/s  %.2f us/step  frac %.3f' % ('$1', '$2', d['value'], r['kernel_ms']*1e3, r['frac']))"; }
for rep in 1 2 3; do
  for w in sweep umbrella_length umbrella_distract memory_size; do
    BSX_NATIVE_LIB=$base timeout 200 python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline --no-also 2>/dev/null | line base $w
    timeout 200 python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline --no-also 2>/dev/null | line packed $w
  done
done | tee $out/ab.log
