#!/bin/bash
# A/B inside one gpurun call: lean kernel instantiations inside the grouped launches (this tree) against the
# baseline build (bsuite_amd/_lib_base: every segment through the run-time LOG / NOISE / MT instantiation).
base=$PWD/bsuite_amd/_lib_base/libbsuite_amd.so
line() { python -c "
import sys,json
l=[x for x in sys.stdin.read().splitlines() if x.startswith('{')]
d=json.loads(l[-1]); r=d['roofline']; print('%-10s %-12s %.3e env-steps/s  %.2f us/step  frac %.3f' % ('$1', '$2', d['value'], r['kernel_ms']*1e3, r['frac']))"; }
for rep in 1 2 3; do
  for pipe in 1 0; do
    BSX_NATIVE_LIB=$base BSX_SWEEP_PIPELINED=$pipe timeout 200 python bench.py --workload sweep --steps 200 --warmup 20 --no-cpu-baseline --no-also 2>/dev/null | line base pipelined=$pipe
    BSX_SWEEP_PIPELINED=$pipe timeout 200 python bench.py --workload sweep --steps 200 --warmup 20 --no-cpu-baseline --no-also 2>/dev/null | line lean pipelined=$pipe
  done
done
timeout 100 python tools/sweep_phase0_trace.py --out gpurun_out/sweep_phase0_trace_lean.json 2>&1 | grep -v amdgpu | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('span', d['span_us'], 'last_start', d['last_start_us'], {k:(round(v['life_us_median'],1), round(v['life_us_p95'],1)) for k,v in d['families'].items()})
"
