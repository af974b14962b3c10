mkdir -p gpurun_out/pool
for lib in bsuite_amd/_lib/libbsuite_amd_tuning.so "" bsuite_amd/_lib/libbsuite_amd_tuning.so ""; do
  echo "== lib=${lib:-product RUN=8} (tuning = RUN=16)"
  BSX_NATIVE_LIB=$lib timeout 120 python tools/lanes_sweep.py --mode rollout --T 16 --steps 480 cartpole mountain_car -- 2**20 2>&1 | grep workload
done 2>&1 | tee gpurun_out/pool/ab_run16.log
