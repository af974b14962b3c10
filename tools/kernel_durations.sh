repo=$PWD; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
for w in bandit mountain_car; do
rm -rf /tmp/prof_k
timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_k -- python $repo/bench.py --workload $w --steps 300 --warmup 50 --no-cpu-baseline --no-also --no-stagger > /tmp/b.json 2>/dev/null
tail -1 /tmp/b.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w bench (under rocprof): %.2f us/step' % (d['roofline']['kernel_ms']*1e3))"
f=$(find /tmp/prof_k -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, statistics
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'small_obs_kernel' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
rows=rows[-300:]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
gap=[(int(b['Start_Timestamp'])-int(a['End_Timestamp']))/1e3 for a,b in zip(rows,rows[1:])]
per=[(int(b['Start_Timestamp'])-int(a['Start_Timestamp']))/1e3 for a,b in zip(rows,rows[1:])]
print(' kernel us: min %.2f median %.2f | gap between kernels: median %.2f min %.2f | start-to-start median %.2f' % (min(d), statistics.median(d), statistics.median(gap), min(gap), statistics.median(per)))
print(' VGPR', rows[0].get('VGPR_Count'), 'SGPR', rows[0].get('SGPR_Count'), 'LDS', rows[0].get('LDS_Block_Size'), 'grid', rows[0].get('Grid_Size_X'), 'wg', rows[0].get('Workgroup_Size_X'))
PY
done
rm -rf /tmp/prof_k
timeout 100 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_k -- $repo/tools/step_floor.bin > /dev/null 2>&1
f=$(find /tmp/prof_k -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, statistics, collections
by=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    by[(r['Kernel_Name'][:40], r['Grid_Size_X'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(by.items()):
    if k[1]=='1048576': print(' microbench', k, 'kernel us min %.2f median %.2f' % (min(v), statistics.median(v)))
PY
