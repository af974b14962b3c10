#!/bin/bash
# Round 6: cache policy of the EAGER step's outputs — plain (snt0) / non-temporal (ent) / write-through sc1 (product) — measured as
# the step alone AND as a closed loop with a device-side policy that reads every observation (examples/closed_loop_policy.py);
# and the 64-lane tiles of small catch batches with nt (product) / sc1 (t64wt) / plain stores (t64pl)
set -u
out=$PWD/gpurun_out/r06c; mkdir -p $out
{
for rep in 1 2 3; do for v in snt0 ent product; do
lib=tools/ab/libbsuite_amd_$v.so; [ $v = product ] && lib=""
for id in cartpole/0 mountain_car/0 bandit/0 discounting_chain/0 memory_len/10 umbrella_length/10 memory_size/16; do
  w=${id%/*}; ns=""; [ $w = mountain_car ] && ns="--no-stagger"
  e=$(BSX_NATIVE_LIB=$lib timeout 200 python bench.py --workload $w --steps 400 --warmup 40 --no-cpu-baseline --no-also $ns 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l); print('%.2f' % (d['roofline']['kernel_ms']*1e3))")
  c=$(BSX_NATIVE_LIB=$lib timeout 200 python examples/closed_loop_policy.py $id 1048576 400 2>/dev/null | tail -1 | python -c "import sys,json; print('%.1f' % (json.loads(sys.stdin.read())['ms_per_step']*1e3))")
  echo "$id lib=$v eager $e us | closed loop with policy $c us"
done; done; done
for rep in 1 2 3; do for v in t64pl t64wt product; do
lib=tools/ab/libbsuite_amd_$v.so; [ $v = product ] && lib=""
for lanes in 131072 262144; do
  e=$(BSX_NATIVE_LIB=$lib timeout 200 python bench.py --workload catch --lanes $lanes --steps 400 --warmup 40 --no-cpu-baseline --no-also 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l); print('%.2f' % (d['roofline']['kernel_ms']*1e3))")
  c=$(BSX_NATIVE_LIB=$lib timeout 200 python examples/closed_loop_policy.py catch/0 $lanes 400 2>/dev/null | tail -1 | python -c "import sys,json; print('%.1f' % (json.loads(sys.stdin.read())['ms_per_step']*1e3))")
  echo "catch/0@$lanes lib=$v eager $e us | closed loop with policy $c us"
done; done; done
} > $out/ab_eager_output_policy.log 2>&1
