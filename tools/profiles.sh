#!/bin/bash
# Regenerates the round's evidence under gpurun_out/final/ from the current code (one gpurun call); copy what
# should be judged into profiles/rNN/.  rocprofv3 runs from /tmp; PMC counters in their own passes with
# --kernel-trace only (tools/pmc_traffic.py, tools/pmc_insts.py).
set -u
out=$PWD/gpurun_out/final; mkdir -p $out
repo=$PWD
B=1048576
( time timeout 1500 python -m pytest tests -m gpu -q ) > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
timeout 300 python bench.py > $out/bench_default.json 2> $out/bench_default.err
# the multi-rank path, executed for real on this ONE GPU (both ranks on cuda:0, gloo): functional evidence, NOT a scaling number
BSX_BENCH_BACKEND=gloo BSX_BENCH_SINGLE_DEVICE=1 timeout 400 python bench.py --gpus 2 --lanes 524288 --no-cpu-baseline 2>/dev/null | grep '^{' > $out/bench_2ranks_on_one_gpu_gloo.json
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_default -- python $repo/bench.py --no-cpu-baseline > $out/bench_default_under_rocprof.json 2>$out/rocprof_default.err
f=$(find /tmp/prof_default -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $out/bench_default_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_sweep -- python $repo/bench.py --workload sweep --steps 40 --warmup 10 > $out/bench_sweep_under_rocprof.json 2>/dev/null
f=$(find /tmp/prof_sweep -name "*kernel_trace.csv" | head -1)
python - "$f" > $out/sweep_whole_group_timeline.txt <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows=[r for r in rows if any(k in r['Kernel_Name'] for k in ('group_kernel', 'counter_add', 'pair_mixed', 'sweep_phase0', 'sweep_pipelined'))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'sweep_phase0' in r['Kernel_Name'] or 'sweep_pipelined' in r['Kernel_Name']]
for s in (-4,-3,-2):
    a,b=idx[s], idx[s+1]
    t0=int(rows[a]['Start_Timestamp'])
    print('--- sweep step')
    for r in rows[a:b]:
        n=r['Kernel_Name'].split('(')[0][-60:]
        print(f"{(int(r['Start_Timestamp'])-t0)/1e3:8.1f} -> {(int(r['End_Timestamp'])-t0)/1e3:8.1f} us  {n}")
PY
cd $repo
timeout 200 python bench.py --workload sweep --no-cpu-baseline > $out/bench_sweep_config5.json 2>/dev/null
BSX_SWEEP_PIPELINED=0 timeout 200 python bench.py --workload sweep --no-cpu-baseline > $out/bench_sweep_config5_two_launches.json 2>/dev/null
timeout 200 python tools/sweep_phase0_trace.py --out $out/sweep_phase0_trace.json > $out/sweep_phase0_trace.log 2>&1
for w in bandit discounting_chain memory_len umbrella_length umbrella_distract memory_size cartpole mountain_car catch deep_sea mnist; do
  timeout 100 python bench.py --workload $w --steps 200 --warmup 40 --no-cpu-baseline --no-also 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-18s eager  %.3e env-steps/s  %.2f us/step  %.0f GB/s  frac %.3f' % ('$w', d['value'], r['kernel_ms']*1e3, r['achieved'], r['frac']))"
done > $out/bench_all_workloads_eager.log
timeout 600 python tools/pmc_traffic.py deep_sea $out/deep_sea_pmc_traffic.json $((3621*B)) "bsx_advance_kernel<deep_sea_fam>" "bsx_hot_stream_kernel<deep_sea_hot" -- --steps 20 --warmup 4 --no-cpu-baseline --no-also --workload deep_sea > $out/pmc_deep_sea.log 2>&1
timeout 600 python tools/pmc_traffic.py catch $out/catch_pmc_traffic.json $((221*B)) "bsx_advance_kernel<catch_fam>" "bsx_hot_stream_kernel<catch_hot" -- --steps 20 --warmup 4 --no-cpu-baseline --no-also --workload catch > $out/pmc_catch.log 2>&1
timeout 600 python tools/pmc_traffic.py cartpole $out/cartpole_pmc_traffic.json $((85*B)) "small_obs_kernel<cartpole_env" -- --steps 20 --warmup 4 --no-cpu-baseline --no-also --workload cartpole > $out/pmc_cartpole.log 2>&1
timeout 600 python tools/pmc_traffic.py mountain_car $out/mountain_car_pmc_traffic.json $((49*B)) "small_obs_kernel<mountain_car_env" -- --steps 20 --warmup 4 --no-cpu-baseline --no-also --workload mountain_car > $out/pmc_mountain_car.log 2>&1
for w in cartpole mountain_car; do
  timeout 300 python tools/pmc_insts.py $out/${w}_pmc_insts.json "small_obs_kernel<${w}_env" -- --workload $w --steps 40 --warmup 8 --no-cpu-baseline --no-also > $out/pmc_insts_$w.log 2>&1
done
timeout 300 python tools/physics_error.py > $out/physics_error.log 2>&1; cp gpurun_out/physics_error.json $out/ 2>/dev/null
timeout 100 python tools/launch_floor.py > $out/launch_floor_fill_kernels.log 2>/dev/null
ls -la $out
