out=$PWD/gpurun_out/final; mkdir -p $out
timeout 300 python bench.py > $out/bench_default.json 2> $out/bench_default.err; wc -c $out/bench_default.json
timeout 300 python -m pytest tests/test_gpu_bench_multirank.py -x -q 2>&1 | tail -3
