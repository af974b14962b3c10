mkdir -p gpurun_out/pool
for lib in tools/ab/libbsuite_amd_prev.so ""; do
  echo "== lib=${lib:-new}"
  BSX_NATIVE_LIB=$lib timeout 120 python tools/lanes_sweep.py --mode rollout --T 16 --steps 320 cartpole mountain_car -- 2**20 2**17 2>&1 | grep workload
  BSX_NATIVE_LIB=$lib timeout 120 python tools/lanes_sweep.py --mode rollout --T 64 --steps 320 cartpole mountain_car -- 2**20 2>&1 | grep workload
done 2>&1 | tee gpurun_out/pool/ab_rows_via_lds.log
timeout 300 python -m pytest tests/test_gpu_rollout.py -x -q 2>&1 | tail -5 | tee gpurun_out/pool/pytest_rollout.log
