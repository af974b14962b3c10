out=$PWD/gpurun_out/tl; mkdir -p $out; repo=$PWD
cd /tmp && export TMPDIR=/tmp
for hf in 1 0; do
rm -rf /tmp/prof_sweep
BSX_SWEEP_HEAVY_FIRST=$hf timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_sweep -- python $repo/bench.py --workload sweep --steps 40 --warmup 10 --no-cpu-baseline > $out/bench_sweep_under_rocprof_$hf.json 2>/dev/null
f=$(find /tmp/prof_sweep -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, statistics
rows=list(csv.DictReader(open(sys.argv[1])))
for key in ('sweep_phase0','pair_mixed_stream'):
    d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows if key in r['Kernel_Name']]
    d=d[len(d)//2:]
    print(key, 'n',len(d),'median us', round(statistics.median(d),2), 'min', round(min(d),2))
PY
done
