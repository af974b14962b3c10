out=$PWD/gpurun_out/final; mkdir -p $out
A="--no-cpu-baseline --no-also"
timeout 200 python tools/pmc.py sq cartpole_rollout16 $out/cartpole_rollout16_pmc_sq.json --kernels "small_obs_kernel<cartpole_env, true" --last 4 -- --workload cartpole --rollout 16 --steps 64 --warmup 16 $A 2>&1 | tail -1
timeout 300 python bench.py > $out/bench_default.json 2> $out/bench_default.err; wc -c $out/bench_default.json
