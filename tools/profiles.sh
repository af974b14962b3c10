#!/bin/bash
# Regenerates the round's evidence under gpurun_out/final/ from the current code (one gpurun call); copy what should
# be judged into profiles/rNN/.  rocprofv3 runs from /tmp; every PMC group in its own pass with --kernel-trace only
# (tools/pmc.py); kernel averages over the TIMED launches only (tools/kernel_stats.py).
#   bash tools/profiles.sh [quick]     quick: skip the test suite and the secondary workloads
set -u
mode=${1:-}
out=$PWD/gpurun_out/final; mkdir -p $out
B=1048576
A="--no-cpu-baseline --no-also"
if [ "$mode" != quick ] && [ -z "${SKIP_PYTEST:-}" ]; then
  ( time timeout 1700 python -m pytest tests -m gpu -q --durations=8 ) > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
fi
timeout 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err; wc -c $out/bench_default.json
# the headline under rocprofv3: kernel averages over the timed launches + the bench line of the same run
timeout 400 python tools/kernel_stats.py $out/bench_headline_kernel_stats.csv -- --steps 200 --warmup 20 $A > /dev/null 2>$out/kernel_stats.err
timeout 400 python tools/kernel_stats.py $out/bench_catch_kernel_stats.csv -- --workload catch --steps 200 --warmup 20 $A > /dev/null 2>>$out/kernel_stats.err
# the sweep, one schedule per trace (split = the default closed-loop cut: sweep_phase0_kernel is its first launch — the
# workgroups the stream waits for — and sweep_pipelined_kernel its second: the store stream beside the small families)
for sch in split closed pipelined; do
  timeout 400 python tools/kernel_stats.py $out/bench_sweep_${sch}_kernel_stats.csv --last 100 -- --workload sweep --sweep-schedule $sch --steps 100 --warmup 20 > /dev/null 2>>$out/kernel_stats.err
done
# a rank's share of an 8-GPU strong-scaled run (2^17 lanes): deep_sea is ONE launch there (deep_sea_step1_kernel), catch the fused tile step
timeout 300 python tools/kernel_stats.py $out/deep_sea_2p17_kernel_stats.csv -- --lanes 131072 --steps 200 --warmup 20 $A > /dev/null 2>>$out/kernel_stats.err
timeout 300 python tools/kernel_stats.py $out/catch_2p17_kernel_stats.csv -- --workload catch --lanes 131072 --steps 200 --warmup 20 $A > /dev/null 2>>$out/kernel_stats.err
# HBM traffic (WRITE_SIZE / FETCH_SIZE, separate passes) of every BASELINE config
pm() { timeout 240 python tools/pmc.py "$@" 2>&1 | tail -1; }
pm traffic deep_sea $out/deep_sea_pmc_traffic.json --kernels "bsx_advance2_kernel<deep_sea_fam" "bsx_advance_kernel<deep_sea_fam" "bsx_hot_stream_kernel<deep_sea_hot" --alg-bytes $((3621*B)) -- --steps 20 --warmup 4 $A --workload deep_sea
pm traffic catch $out/catch_pmc_traffic.json --kernels "bsx_advance2_kernel<catch_fam" "bsx_advance_kernel<catch_fam" "bsx_hot_stream_kernel<catch_hot" --alg-bytes $((221*B)) -- --steps 20 --warmup 4 $A --workload catch
pm traffic cartpole $out/cartpole_pmc_traffic.json --kernels "small_obs_kernel<cartpole_env" --alg-bytes $((85*B)) -- --steps 20 --warmup 4 $A --workload cartpole
# (mountain_car lock-step: staggering its 1001-call episodes is ~10^4 torch launches, minutes under a PMC pass; in 24
#  calls no lane resets either way)
pm traffic mountain_car $out/mountain_car_pmc_traffic.json --kernels "small_obs_kernel<mountain_car_env" "small_obs_eager2_kernel<mountain_car_env" --alg-bytes $((49*B)) -- --steps 20 --warmup 4 $A --no-stagger --workload mountain_car
pm traffic sweep_closed $out/sweep_closed_pmc_traffic.json --kernels sweep_phase0_kernel pair_mixed_stream_kernel --alg-bytes 885580000 --last 40 -- --workload sweep --sweep-schedule closed --steps 40 --warmup 10
pm traffic sweep_split $out/sweep_split_pmc_traffic.json --kernels sweep_phase0_kernel sweep_pipelined_kernel --alg-bytes 885580000 --last 40 -- --workload sweep --sweep-schedule split --steps 40 --warmup 10
pm traffic sweep_pipelined $out/sweep_pipelined_pmc_traffic.json --kernels sweep_pipelined_kernel --alg-bytes 885580000 --last 40 -- --workload sweep --sweep-schedule pipelined --steps 40 --warmup 10
# the other families north_star names, and the mnist bandit of the sweep: traffic + kernel averages
pm traffic bandit $out/bandit_pmc_traffic.json --kernels "small_obs_kernel<bandit_env" "small_obs_eager2_kernel<bandit_env" --alg-bytes $((25*B)) -- --steps 20 --warmup 4 $A --workload bandit
pm traffic discounting_chain $out/discounting_chain_pmc_traffic.json --kernels "small_obs_kernel<discounting_chain_env" "small_obs_eager2_kernel<discounting_chain_env" --alg-bytes $((29*B)) -- --steps 20 --warmup 4 $A --workload discounting_chain
pm traffic memory_len $out/memory_len_pmc_traffic.json --kernels "small_obs_kernel<memory_chain_env" "small_obs_eager2_kernel<memory_chain_env" --alg-bytes $((49*B)) -- --steps 20 --warmup 4 $A --workload memory_len
pm traffic umbrella_length $out/umbrella_length_pmc_traffic.json --kernels "small_obs_kernel<umbrella_chain_env" --alg-bytes $((113*B)) -- --steps 20 --warmup 4 $A --workload umbrella_length
pm traffic mnist $out/mnist_pmc_traffic.json --kernels mnist_advance_kernel mnist_observe_kernel --alg-bytes $((3157*B)) -- --steps 20 --warmup 4 $A --workload mnist
# ... and behind every rollout / wrapped record of the default line: a fused rollout launch runs 16 steps
# (per-launch bytes / 16); a pipelined rollout of deep_sea / catch is T+1 launches per call (the timed dispatches summed / their steps)
pm traffic catch_noise $out/catch_noise_pmc_traffic.json --kernels "catch_fam" "catch_hot" --alg-bytes $((221*B)) -- --steps 20 --warmup 4 $A --workload catch_noise
pm traffic deep_sea_logging $out/deep_sea_logging_pmc_traffic.json --kernels "deep_sea_fam" "deep_sea_hot" --alg-bytes $((3621*B)) -- --steps 20 --warmup 4 $A --workload deep_sea --logging
pm traffic catch_rollout32 $out/catch_rollout32_pmc_traffic.json --kernels "catch_fam" "catch_hot" --alg-bytes $((221*B)) --last 7 --last-total $((7*33)) --steps-total 224 -- --workload catch --rollout 32 --steps 224 --warmup 32 $A
pm traffic deep_sea_rollout16 $out/deep_sea_rollout16_pmc_traffic.json --kernels "deep_sea_fam" "deep_sea_hot" --alg-bytes $((3621*B)) --last 3 --last-total $((3*17)) --steps-total 48 -- --workload deep_sea --rollout 16 --steps 48 --warmup 16 $A
for wk in "cartpole cartpole_env 40" "mountain_car mountain_car_env 26.5" "bandit bandit_env 17.5" "discounting_chain discounting_chain_env 21.5" "memory_len memory_chain_env 26.5" "umbrella_length umbrella_chain_env 105.5"; do
  set -- $wk; ns=""; [ $1 = mountain_car ] && ns="--no-stagger"
  alg=$(python -c "print(int($3*$B))")
  pm traffic ${1}_rollout16 $out/${1}_rollout16_pmc_traffic.json --kernels "small_obs_lean_rollout_kernel<$2" "small_obs_kernel<$2" --alg-bytes $alg --last 4 --steps-per-launch 16 -- --workload $1 --rollout 16 --steps 64 --warmup 16 $A $ns
done
for w in bandit discounting_chain memory_len umbrella_length mnist; do
  timeout 300 python tools/kernel_stats.py $out/${w}_kernel_stats.csv -- --workload $w --steps 200 --warmup 20 $A > /dev/null 2>>$out/kernel_stats.err
done
# issue-side counters of the two store streams round 6 rebuilt: the mnist observation stream and the sweep's mixed stream
pm sq mnist_eager $out/mnist_eager_pmc_sq.json --kernels "mnist_observe_kernel" -- --workload mnist --steps 20 --warmup 4 $A
pm sq sweep_split $out/sweep_split_pmc_sq.json --kernels sweep_pipelined_kernel --last 40 -- --workload sweep --sweep-schedule split --steps 40 --warmup 10
pm sq deep_sea_eager $out/deep_sea_eager_pmc_sq.json --kernels "bsx_hot_stream_kernel<deep_sea_hot" -- --workload deep_sea --steps 20 --warmup 4 $A
pm sq umbrella_length_eager $out/umbrella_length_eager_pmc_sq.json --kernels "small_obs_kernel<umbrella_chain_env" -- --workload umbrella_length --steps 20 --warmup 4 $A
# issue-side counters of the fused rollouts (bound "valu") and of the eager physics steps
for w in cartpole mountain_car; do
  ns=""; [ $w = mountain_car ] && ns="--no-stagger"
  pm sq ${w}_rollout16 $out/${w}_rollout16_pmc_sq.json --kernels "small_obs_lean_rollout_kernel<${w}_env" --last 4 -- --workload $w --rollout 16 --steps 64 --warmup 16 $A $ns
  pm sq ${w}_eager $out/${w}_eager_pmc_sq.json --kernels "small_obs_kernel<${w}_env, false" "small_obs_eager2_kernel<${w}_env" -- --workload $w --steps 20 --warmup 4 $A $ns
done
if [ "$mode" != quick ]; then
  for w in bandit discounting_chain memory_len umbrella_length umbrella_distract memory_size cartpole mountain_car catch deep_sea mnist; do
    timeout 100 python bench.py --workload $w --steps 200 --warmup 40 $A 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-18s eager  %.3e env-steps/s  %.2f us/step  %.0f GB/s  frac %.3f' % ('$w', d['value'], r['kernel_ms']*1e3, r['achieved'], r['frac']))"
  done > $out/bench_all_workloads_eager.log
  timeout 900 python tools/strong_scaling_proxy.py $out/strong_scaling_proxy.json > $out/strong_scaling_proxy.log 2>&1
  BSX_BENCH_BACKEND=gloo BSX_BENCH_SINGLE_DEVICE=1 timeout 400 python bench.py --gpus 2 --no-cpu-baseline 2>/dev/null | grep '^{' > $out/bench_2ranks_on_one_gpu_gloo.json
  # the driver's 8-GPU command, rehearsed with eight ranks on the one GPU (the numbers mean nothing; the line's shape does)
  BSX_BENCH_BACKEND=gloo BSX_BENCH_SINGLE_DEVICE=1 timeout 900 python bench.py --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' > $out/bench_8ranks_on_one_gpu_gloo.json
  timeout 300 python tools/physics_error.py > $out/physics_error.log 2>&1; cp gpurun_out/physics_error.json $out/ 2>/dev/null
  timeout 300 python tools/fuzz_gpu.py --seconds ${FUZZ_SECONDS:-300} --seed 11 > $out/fuzz_gpu.log 2>&1; tail -1 $out/fuzz_gpu.log
  # the engine against the UNMODIFIED reference, live: 8 x LIVE_CASES random cases (families, kwargs, wrappers, resets, policies)
  ( BSX_LIVE_CASES=${LIVE_CASES:-1000} timeout 900 python -m pytest tests/test_gpu_vs_reference_live.py -q -m gpu ) > $out/live_reference_cases.log 2>&1; tail -2 $out/live_reference_cases.log
fi
ls -la $out
