#!/bin/bash
# Round 6: non-temporal stores for (w55) the wide rows' 16-byte tile chunks and (t87) the small-batch fused tile / single-launch
# streams of catch and deep_sea, vs the product
set -u
out=$PWD/gpurun_out/r06c; mkdir -p $out
run() { BSX_NATIVE_LIB=$1 timeout 200 python bench.py --workload $3 $4 --steps 320 --warmup 32 --no-cpu-baseline --no-also 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l); r=d['roofline']; print('%-18s %-28s lib=%-8s %.2f us/step' % (sys.argv[1], sys.argv[3] or 'eager', sys.argv[2], r['kernel_ms']*1e3))
" "$3" "$2" "$4"; }
{
for rep in 1 2 3; do
for v in product w55; do
  lib=tools/ab/libbsuite_amd_$v.so; [ $v = product ] && lib=""
  for w in umbrella_length umbrella_distract memory_size; do run "$lib" $v $w ""; run "$lib" $v $w "--rollout 16"; done
done
for v in product t87; do
  lib=tools/ab/libbsuite_amd_$v.so; [ $v = product ] && lib=""
  for lanes in 131072 262144 524288; do run "$lib" $v catch "--lanes $lanes"; run "$lib" $v catch "--lanes $lanes --rollout 32"; done
  for lanes in 131072 262144; do run "$lib" $v deep_sea "--lanes $lanes"; done
done
for v in product w55 t87; do
  lib=tools/ab/libbsuite_amd_$v.so; [ $v = product ] && lib=""
  BSX_NATIVE_LIB=$lib timeout 300 python bench.py --workload sweep --steps 200 --warmup 40 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l); print('sweep              split                        lib=%-8s %.2f us/step' % (sys.argv[1], d['ms_per_step']*1e3)); print('sweep              pipelined                    lib=%-8s %.2f us/step' % (sys.argv[1], d['pipelined']['ms_per_step']*1e3))
" "$v"
done
done
} > $out/ab_nt_wide_rows_and_small_batches.log 2>&1
