#!/bin/bash
# One gpurun call: the GPU test suite, the default bench line, the per-rank strong-scaling proxy and the
# fused-step A/B.  Outputs under gpurun_out/r03/ (copy what should be judged into profiles/r03/).
set -u
out=$PWD/gpurun_out/r03; mkdir -p $out
( time timeout 1700 python -m pytest tests -m gpu -x -q --durations=15 ) > $out/pytest_gpu.log 2>&1; tail -25 $out/pytest_gpu.log
timeout 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err; tail -c 600 $out/bench_default.err; wc -c $out/bench_default.json
timeout 400 python tools/strong_scaling_proxy.py $out/strong_scaling_proxy.json > $out/strong_scaling_proxy.log 2>&1; tail -20 $out/strong_scaling_proxy.log
T=$PWD/bsuite_amd/_lib/libbsuite_amd_tuning.so
for fused in 0 1073741824; do
  echo "# BSX_FUSED_TILE_MAX_BYTES=$fused"
  BSX_NATIVE_LIB=$T BSX_FUSED_TILE_MAX_BYTES=$fused timeout 300 python tools/lanes_sweep.py catch deep_sea -- 2**13 2**15 2**16 2**17 2**18 2**19 2>&1 | grep '^{'
  BSX_NATIVE_LIB=$T BSX_FUSED_TILE_MAX_BYTES=$fused timeout 300 python tools/lanes_sweep.py --mode rollout --T 32 catch -- 2**16 2**17 2**18 2>&1 | grep '^{'
done > $out/ab_fused_tile.log 2>&1
cat $out/ab_fused_tile.log
