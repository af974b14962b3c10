mkdir -p gpurun_out
python tools/host_overhead.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/host_overhead_after.log
for w in bandit discounting_chain memory_len mountain_car cartpole catch bandit cartpole; do
  timeout 100 python bench.py --workload $w --steps 400 --warmup 50 --no-cpu-baseline --no-also 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-18s eager  %.3e env-steps/s  %.2f us/step  frac %.3f' % ('$w', d['value'], r['kernel_ms']*1e3, r['frac']))"
done | tee gpurun_out/bench_small_after_host_path.log
