mkdir -p gpurun_out/pool
for lib in "" tools/ab/libbsx_exp1.so tools/ab/libbsx_exp2.so tools/ab/libbsx_exp3.so; do
  echo "== lib=${lib:-new}"
  BSX_NATIVE_LIB=$lib timeout 100 python tools/lanes_sweep.py --mode rollout --T 16 --steps 320 cartpole mountain_car -- 2**20 2>&1 | grep workload
done 2>&1 | tee gpurun_out/pool/exp_store_ablation.log
