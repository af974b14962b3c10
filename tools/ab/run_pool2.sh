mkdir -p gpurun_out/pool
for lib in "" tools/ab/libbsuite_amd_prev.so "" tools/ab/libbsuite_amd_prev.so; do
  echo "== lib=${lib:-new}"
  BSX_NATIVE_LIB=$lib timeout 120 python tools/lanes_sweep.py --mode rollout --T 16 --steps 640 cartpole mountain_car -- 2**17 2**18 2>&1 | grep workload
done 2>&1 | tee gpurun_out/pool/ab_rows_small_batches.log
timeout 200 python -m pytest tests/test_gpu_oracle_batch.py tests/test_gpu_golden.py tests/test_gpu_benched_sizes.py tests/test_gpu_all_ids.py -x -q -k "cartpole or mountain or physics" 2>&1 | tail -4 | tee gpurun_out/pool/pytest_physics.log
timeout 60 python tools/fuzz_gpu.py --seconds 40 2>&1 | tail -3 | tee gpurun_out/pool/fuzz40.log
