#!/bin/bash
# The part of tools/profiles.sh that a host-side change invalidates: GPU test log, default bench line (plain and
# under rocprofv3 --kernel-trace --stats), the two-rank functional run, one eager line per workload.
set -u
out=$PWD/gpurun_out/final; mkdir -p $out; repo=$PWD
( time timeout 1500 python -m pytest tests -m gpu -q ) > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
timeout 300 python bench.py > $out/bench_default.json 2> $out/bench_default.err
BSX_BENCH_BACKEND=gloo BSX_BENCH_SINGLE_DEVICE=1 timeout 400 python bench.py --gpus 2 --lanes 524288 --no-cpu-baseline 2>/dev/null | grep '^{' > $out/bench_2ranks_on_one_gpu_gloo.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_default
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_default -- python $repo/bench.py --no-cpu-baseline > $out/bench_default_under_rocprof.json 2>$out/rocprof_default.err
f=$(find /tmp/prof_default -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $out/bench_default_kernel_stats.csv
cd $repo
for w in bandit discounting_chain memory_len umbrella_length umbrella_distract memory_size cartpole mountain_car catch deep_sea mnist; do
  timeout 100 python bench.py --workload $w --steps 200 --warmup 40 --no-cpu-baseline --no-also 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-18s eager  %.3e env-steps/s  %.2f us/step  %.0f GB/s  frac %.3f' % ('$w', d['value'], r['kernel_ms']*1e3, r['achieved'], r['frac']))"
done > $out/bench_all_workloads_eager.log
cat $out/bench_all_workloads_eager.log
