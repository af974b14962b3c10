mkdir -p gpurun_out/pool
T=bsuite_amd/_lib/libbsuite_amd_tuning.so
for big in 1 1000000 1 1000000; do
  echo "== BSX_REGS_ROLLOUT_BIG_MIN_BLOCKS=$big"
  BSX_NATIVE_LIB=$T BSX_REGS_ROLLOUT_BIG_MIN_BLOCKS=$big timeout 120 python tools/lanes_sweep.py --mode eager --steps 400 cartpole -- 2**20 2>&1 | grep workload
done 2>&1 | tee gpurun_out/pool/ab_eager_pooled_resets.log
BSX_NATIVE_LIB=$T BSX_REGS_ROLLOUT_BIG_MIN_BLOCKS=1 timeout 100 python -m pytest tests/test_gpu_golden.py tests/test_gpu_oracle_batch.py -x -q -k cartpole 2>&1 | tail -2
