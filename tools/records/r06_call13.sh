#!/bin/bash
# Round 6, last evidence call: the parts of the final-tree evidence the write-through catch tiles touch (the sweep's phase 0,
# catch at 2^19 lanes, the per-rank proxy) + the default bench line of the final tree
set -u
out=$PWD/gpurun_out/final2; mkdir -p $out
A="--no-cpu-baseline --no-also"
timeout 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err; wc -c $out/bench_default.json
for sch in split closed pipelined; do
  timeout 400 python tools/kernel_stats.py $out/bench_sweep_${sch}_kernel_stats.csv --last 100 -- --workload sweep --sweep-schedule $sch --steps 100 --warmup 20 > /dev/null 2>>$out/kernel_stats.err
done
timeout 300 python tools/kernel_stats.py $out/catch_2p19_kernel_stats.csv -- --workload catch --lanes 524288 --steps 200 --warmup 20 $A > /dev/null 2>>$out/kernel_stats.err
pm() { timeout 240 python tools/pmc.py "$@" 2>&1 | tail -1; }
pm traffic sweep_closed $out/sweep_closed_pmc_traffic.json --kernels sweep_phase0_kernel pair_mixed_stream_kernel --alg-bytes 885580000 --last 40 -- --workload sweep --sweep-schedule closed --steps 40 --warmup 10
pm traffic sweep_split $out/sweep_split_pmc_traffic.json --kernels sweep_phase0_kernel sweep_pipelined_kernel --alg-bytes 885580000 --last 40 -- --workload sweep --sweep-schedule split --steps 40 --warmup 10
timeout 900 python tools/strong_scaling_proxy.py $out/strong_scaling_proxy.json > $out/strong_scaling_proxy.log 2>&1
ls -la $out
