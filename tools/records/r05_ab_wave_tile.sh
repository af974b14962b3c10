# RECORD of a measurement, not a tool: BSX_WAVE_TILE exists in commit 81bebd8 only (the experiment was removed afterwards).
# wide rows of memory_chain / umbrella_chain, single steps: one tile of 64 lanes per WAVE (small_obs_wave_tile_kernel: wave-private
# flat bit planes, no workgroup barrier between step and stores) against one tile of 256 lanes per workgroup (the PACKED
# path: three barriers per step).  Tuning build, BSX_WAVE_TILE.
out=$PWD/gpurun_out/r05g; mkdir -p $out
A="--no-cpu-baseline --no-also"
T=$(python -c "from bsuite_amd import build; print(build.build(tuning=True))")
one() { python -c "
import sys,json
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l); r=d['roofline']
    print('%-66s %8.2f us/step  frac %.3f' % (sys.argv[1], r.get('kernel_ms', d['ms_per_step'])*1e3, r['frac']))
" "$1"; }
BSX_NATIVE_LIB=$T BSX_WAVE_TILE=1 timeout 900 python -m pytest tests/test_gpu_oracle_batch.py tests/test_gpu_golden.py tests/test_gpu_dm_env_conformance.py tests/test_gpu_engine_features.py tests/test_gpu_full_size.py tests/test_gpu_wide_rows.py tests/test_gpu_vs_reference_live.py -x -q -m gpu -k "memory or umbrella or wide or chain" 2>&1 | tail -5
{
for rep in 1 2; do
 for wt in 0 1; do
  for w in umbrella_length umbrella_distract memory_size; do
   for lanes in 1048576 262144 65536; do
    BSX_NATIVE_LIB=$T BSX_WAVE_TILE=$wt timeout 120 python bench.py --workload $w --lanes $lanes --steps 300 --warmup 40 $A 2>/dev/null | one "$w eager, $lanes lanes, $( [ $wt = 1 ] && echo 'tile per wave' || echo 'tile per workgroup' ) (rep $rep)"
   done
  done
 done
done
} > $out/ab_wide_rows_wave_tile.log 2>&1; cat $out/ab_wide_rows_wave_tile.log
