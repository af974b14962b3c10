#!/bin/bash
# Round 6: longer parity soak of the final tree (wrapped steps fused up to 256 MiB): fuzz 600 s with a new seed, 16 000 live-reference cases
set -u
out=$PWD/gpurun_out/r06g; mkdir -p $out
timeout 700 python tools/fuzz_gpu.py --seconds 600 --seed 53 > $out/fuzz_gpu_600s_seed53.log 2>&1; tail -1 $out/fuzz_gpu_600s_seed53.log
( BSX_LIVE_CASES=2000 timeout 1500 python -m pytest tests/test_gpu_vs_reference_live.py -q -m gpu ) > $out/live_reference_2000_cases.log 2>&1; tail -2 $out/live_reference_2000_cases.log
