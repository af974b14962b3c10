#!/bin/bash
# Round 6: wrapped steps fuse up to 256 MiB (adopted) — the GPU suite on it, the Logging-wrapped catch step both ways, the default line
set -u
out=$PWD/gpurun_out/r06f; mkdir -p $out
( time timeout 1700 python -m pytest tests -m gpu -q -x ) > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
us() { python -c "
import sys,json
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l); print('%.2f' % (d['roofline']['kernel_ms']*1e3))"; }
{
for rep in 1 2 3; do for mib in 128 256; do
  e=$(BSX_NATIVE_LIB=bsuite_amd/_lib/libbsuite_amd_tuning.so BSX_FUSED_WRAPPED_MAX_MIB=$mib timeout 200 python bench.py --workload catch --logging --steps 400 --warmup 40 --no-cpu-baseline --no-also 2>/dev/null | us)
  n=$(BSX_NATIVE_LIB=bsuite_amd/_lib/libbsuite_amd_tuning.so BSX_FUSED_WRAPPED_MAX_MIB=$mib timeout 200 python bench.py --workload catch_noise --steps 400 --warmup 40 --no-cpu-baseline --no-also 2>/dev/null | us)
  c=$(BSX_NATIVE_LIB=bsuite_amd/_lib/libbsuite_amd_tuning.so BSX_FUSED_WRAPPED_MAX_MIB=$mib timeout 200 python examples/closed_loop_policy.py catch_noise/0 1048576 400 2>/dev/null | tail -1 | python -c "import sys,json; print('%.1f' % (json.loads(sys.stdin.read())['ms_per_step']*1e3))")
  echo "2^20 lanes, wrapped steps fuse up to ${mib} MiB: catch under Logging eager $e us | catch_noise/0 eager $n us | catch_noise/0 closed loop with policy $c us"
done; done
} > $out/ab_catch_wrapped_fused.log 2>&1
cat $out/ab_catch_wrapped_fused.log
timeout 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err; wc -c $out/bench_default.json
timeout 400 python tools/kernel_stats.py $out/catch_noise_kernel_stats.csv -- --workload catch_noise --steps 200 --warmup 20 --no-cpu-baseline --no-also > /dev/null 2>$out/kernel_stats.err
timeout 240 python tools/pmc.py traffic catch_noise $out/catch_noise_pmc_traffic.json --kernels "catch_fam" "catch_hot" --alg-bytes $((221*1048576)) -- --steps 20 --warmup 4 --no-cpu-baseline --no-also --workload catch_noise 2>&1 | tail -1
