#!/bin/bash
# (the --calibrate-after flag existed only in the tree this call ran on: calibrating first changed nothing and was not kept)
# Round 6: the driver's command (20 timed steps = 12 ms) with the box calibrated before the timed region (new order) and after it
set -u
out=$PWD/gpurun_out/r06f; mkdir -p $out
{
for rep in 1 2 3 4; do for flag in "--calibrate-after" ""; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-also --no-cpu-baseline $flag 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l); r=d['roofline']; print('steps 20 warmup 5 %-18s ms_per_step %.4f  kernel_ms %.4f  frac %.3f  (box fill %.0f GB/s)' % ('$flag' or 'calibrate first', d['ms_per_step'], r['kernel_ms'], r['frac'], r['box_fill_GBps']))"
done; done
} > $out/ab_bench_calibrate_first.log 2>&1
cat $out/ab_bench_calibrate_first.log
