# RECORD of a measurement: BSX_SWEEP_SMALL_ORDER (a hook in sweep_batch.prepare_groups) was not kept; BSX_SPLIT_ROUND is the tuning knob of the adopted top-up.
# split closed-loop sweep step: which small-observation workgroups top launch 1 up — the narrow-row families at the end of
# the wide-first order (default), or the wide-row ones (narrow-first order) — and how many.  Tuning build, BSX_SPLIT_ROUND.
out=$PWD/gpurun_out/r05f; mkdir -p $out
T=$(python -c "from bsuite_amd import build; print(build.build(tuning=True))")
one() { python -c "
import sys,json
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l); r=d['roofline']
    print('%-72s %8.2f us/step  frac %.3f' % (sys.argv[1], d['ms_per_step']*1e3, r['frac']))
" "$1"; }
{
for rep in 1 2 3; do
 for order in wide_first narrow_first; do
  for x in 0 2420 3000 3600; do
   BSX_NATIVE_LIB=$T BSX_SPLIT_ROUND=$x BSX_SWEEP_SMALL_ORDER=$order timeout 300 python bench.py --workload sweep --sweep-schedule split --steps 200 --warmup 40 2>/dev/null | one "sweep split, small segments $order, launch 1 = $x workgroups (rep $rep)"
  done
 done
done
} > $out/ab_sweep_split_order.log 2>&1; cat $out/ab_sweep_split_order.log
