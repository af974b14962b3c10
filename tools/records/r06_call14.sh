#!/bin/bash
# Round 6: parity soak of the last commit — random fuzz cases against the oracle, the unmodified reference live, physics error
set -u
out=$PWD/gpurun_out/final3; mkdir -p $out
timeout 300 python tools/physics_error.py > $out/physics_error.log 2>&1; cp gpurun_out/physics_error.json $out/ 2>/dev/null
timeout 400 python tools/fuzz_gpu.py --seconds 300 --seed 29 > $out/fuzz_gpu_300s_seed29.log 2>&1; tail -1 $out/fuzz_gpu_300s_seed29.log
( BSX_LIVE_CASES=1000 timeout 900 python -m pytest tests/test_gpu_vs_reference_live.py -q -m gpu ) > $out/live_reference_1000_cases.log 2>&1; tail -2 $out/live_reference_1000_cases.log
