#!/bin/bash
# Regenerates the headline evidence under gpurun_out/final/ from the current code (one gpurun call):
# rocprofv3 kernel stats of the default bench, PMC traffic (separate passes), sweep + delta + image lines.
set -u
out=$PWD/gpurun_out/final; mkdir -p $out
repo=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_default -- python $repo/bench.py > $out/deep_sea_bench_under_rocprof.json 2>$out/rocprof_default.err
f=$(find /tmp/prof_default -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $out/deep_sea_bench_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sweep -- python $repo/bench.py --workload sweep --steps 100 --warmup 10 > $out/bench_sweep_under_rocprof.json 2>/dev/null
f=$(find /tmp/prof_sweep -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $out/sweep_grouped_kernel_stats.csv
cd $repo
timeout 200 python bench.py > $out/bench_default_deep_sea_and_catch.json 2>/dev/null
timeout 200 python bench.py --workload sweep > $out/bench_sweep_config5.json 2>/dev/null
for w in deep_sea catch; do timeout 100 python bench.py --workload $w --observation-mode delta --no-cpu-baseline 2>/dev/null | tail -1; done > $out/bench_delta_mode.json
timeout 100 python tools/bench_image.py 2>/dev/null | tail -6 > $out/bench_image_adapter.log
B=1048576
timeout 700 python tools/pmc_traffic.py deep_sea $out/deep_sea_pmc_traffic.json $((3621*B)) "bsx_advance_kernel<deep_sea_fam>" "bsx_hot_stream_kernel<deep_sea_hot" -- --steps 20 --warmup 4 --no-cpu-baseline --workload deep_sea > $out/pmc_deep_sea.log 2>&1
timeout 700 python tools/pmc_traffic.py catch $out/catch_pmc_traffic.json $((221*B)) "bsx_advance_kernel<catch_fam>" "bsx_hot_stream_kernel<catch_hot" -- --steps 20 --warmup 4 --no-cpu-baseline --workload catch > $out/pmc_catch.log 2>&1
timeout 700 python tools/pmc_traffic.py deep_sea_delta $out/deep_sea_delta_pmc_traffic.json $((37*B)) "bsx_advance_delta_kernel<deep_sea_fam" -- --steps 20 --warmup 4 --no-cpu-baseline --workload deep_sea --observation-mode delta > $out/pmc_deep_sea_delta.log 2>&1
ls -la $out
