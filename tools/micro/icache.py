import sys, json, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tools'))
import pmc
wanted = ['sweep_phase0_kernel', 'pair_mixed_stream_kernel', 'small_obs_kernel<', 'bsx_advance_kernel', 'bsx_hot_stream_kernel', 'small_obs_lean_rollout_kernel']
out = {}
for counters in (['SQC_ICACHE_REQ', 'SQC_ICACHE_MISSES', 'SQC_ICACHE_MISSES_DUPLICATE', 'SQ_IFETCH', 'SQ_WAVES'],
                 ['SQ_WAIT_INST_ANY', 'SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQC_TC_INST_REQ', 'SQC_TC_STALL']):
  r = pmc.one_pass(counters, sys.argv[1:], wanted, 30)
  for k, v in r.items():
    out.setdefault(pmc.short(k), {}).update(v)
print(json.dumps(out, indent=1))
