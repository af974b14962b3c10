"""CPU: register / LDS / scratch budgets of the built kernels, read from the code objects' metadata
(tools/kernel_resources.py: clang-offload-bundler + llvm-readelf, no GPU).  The hot kernels are latency- or
throughput-bound through their occupancy — a change that pushes one of them over a VGPR step (64 -> 7 waves per SIMD,
72 -> 6 at 80, ...) or into scratch shows up here before it shows up as a slower bench line."""
import os
import re
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import kernel_resources as kr  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(kr.LLVM, 'clang-offload-bundler')) or shutil.which('c++filt') is None,
                                reason='needs the ROCm LLVM tools')


@pytest.fixture(scope='module')
def kernels():
  from bsuite_amd import build
  ks = kr.kernels(build.build())
  return {k['name'].split('(')[0]: k for k in ks}


# kernel -> most VGPRs it may use (the step below the next occupancy loss, with a little slack)
BUDGET = {
    # BASELINE configs 1/2: deep_sea / catch lane advance + observation store stream (eager and pipelined rollout)
    'bsx_hot_stream_kernel<deep_sea_hot, 4, 256>': 32,
    'bsx_hot_stream_kernel<catch_hot, 2, 256>': 32,
    'bsx_advance_kernel<deep_sea_fam, true, -1>': 64,
    'bsx_advance_kernel<catch_fam, true, -1>': 32,
    'bsx_pipelined_kernel<deep_sea_fam, true, deep_sea_hot, 4>': 64,
    'bsx_pipelined_kernel<catch_fam, true, catch_hot, 2>': 32,
    'bsx_fused_tile_kernel<catch_fam, true, catch_hot>': 32,
    # configs 3/4: the physics families, eager and fused rollout (lean instantiations)
    'small_obs_kernel<cartpole_env, false, 0, 0, 0, true, false>': 40,
    'small_obs_kernel<mountain_car_env, false, 0, 0, 0, true, false>': 32,
    # ... and the lean eager step with two lanes per thread (2^19+ lanes; cartpole's exists, ungated: measured equal)
    'small_obs_eager2_kernel<bandit_env, 0, 2>': 32,
    'small_obs_eager2_kernel<discounting_chain_env, 0, 2>': 32,
    'small_obs_eager2_kernel<memory_chain_env, 0, 2>': 40,
    'small_obs_eager2_kernel<mountain_car_env, 0, 2>': 32,
    # ... their lean fused rollouts: <family, BIG (pooled resets, staged rows), variant, table in LDS>
    'small_obs_lean_rollout_kernel<cartpole_env, true, 0, true>': 64,       # 8 waves: 16 workgroups per CU at 2^20 lanes = 8 + 8
    'small_obs_lean_rollout_kernel<cartpole_env, true, 1, true>': 72,       # swing-up (8-float rows, per-step info): 7 waves
    'small_obs_lean_rollout_kernel<cartpole_env, false, 0, true>': 64,
    'small_obs_lean_rollout_kernel<cartpole_env, false, 1, true>': 64,
    'small_obs_lean_rollout_kernel<mountain_car_env, false, 0, true>': 64,
    # config 5: the whole sweep as one launch group
    'sweep_phase0_kernel': 64,
    'sweep_pipelined_kernel': 64,
    'pair_mixed_stream_kernel': 64,         # 8 waves (the straight-line mnist body holds 6 chunks' state words, offsets and pixels)
    'mnist_observe_kernel<4>': 64,
    # the chains' wide rows, opt-in pair path: lane advance (flat bit planes into the scratch) + the wide-row store stream
    'small_obs_kernel<umbrella_chain_env, false, 0, 0, 0, true, true>': 64,
    'small_obs_kernel<memory_chain_env, false, 0, 0, 0, true, true>': 64,
    'bsx_row_stream_kernel<umbrella_rows, 2>': 32,
    'bsx_row_stream_kernel<memory_rows, 2>': 32,
    'small_obs_mixed_group_kernel': 80,
}
# scratch that is not a spill: a dynamically indexed per-thread array in two non-lean instantiations
KNOWN_SCRATCH = {'small_obs_kernel<umbrella_chain_env, false, -1, -1, -1, true, false>',
                 'bsx_fused_rollout_kernel<deep_sea_fam, false, deep_sea_hot>'}


def test_hot_kernels_stay_inside_their_register_budgets(kernels):
  for name, most in BUDGET.items():
    assert name in kernels, f'{name} is not in the library any more: update BUDGET'
    k = kernels[name]
    assert k['vgpr_count'] <= most, f'{name}: {k["vgpr_count"]} VGPRs > {most} ({k["waves_per_simd"]} waves per SIMD)'
    assert k['agpr_count'] == 0 and k['private_segment_fixed_size'] == 0 and k['vgpr_spill_count'] == 0, (name, k)


def test_no_kernel_spills_vector_registers(kernels):
  assert 100 < len(kernels) < 190          # (the K x block-size matrix of stream kernels exists in the tuning build only)
  for name, k in kernels.items():
    assert k['vgpr_spill_count'] == 0, f'{name} spills {k["vgpr_spill_count"]} VGPRs'
    if name not in KNOWN_SCRATCH:
      assert k['private_segment_fixed_size'] == 0, f'{name} uses {k["private_segment_fixed_size"]} B of scratch'
    assert k['max_flat_workgroup_size'] in (64, 128, 256, 512, 1024)
    assert k['group_segment_fixed_size'] <= 16 * 1024, f'{name}: {k["group_segment_fixed_size"]} B of static LDS'


def test_lean_rollout_step_loops_reload_no_spilled_scalars():
  """the fused cartpole rollout carried ~90 v_readlane_b32 spill reloads in each copy of its step loop (119
  SGPR spills: every `2*k < numel` of the row store was a loop-invariant 64-bit mask).  The lean rollouts are now
  compiled per variant; what the allocator still spills belongs to the prologue / epilogue (state and info column
  pointers), and the step loop — the inner of the two loops — reloads nothing."""
  import kernel_isa as ki
  csrc = os.path.join(ROOT, 'bsuite_amd', 'csrc')
  # kernel -> most reloads in its step loop.  Swing-up (two more thresholds, an f64 move cost, per-step info columns in
  # registers) still reloads ~20 scalars per step; taking them out cost 6-15 VGPRs in every variant tried (small_obs.h)
  for want, most in (('small_obs_lean_rollout_kernel<cartpole_env, true, 0, true>', 0), ('small_obs_lean_rollout_kernel<cartpole_env, false, 0, true>', 0),
                     ('small_obs_lean_rollout_kernel<mountain_car_env, false, 0, true>', 0),
                     ('small_obs_lean_rollout_kernel<cartpole_env, true, 1, true>', 24), ('small_obs_lean_rollout_kernel<cartpole_env, false, 1, true>', 24)):
    name, text = ki.kernel_text(os.path.join(csrc, 'mountain_car.hip' if 'mountain_car' in want else 'cartpole.hip'), want)
    assert sum('v_' in l for l in text) > 300, name
    assert ki.loop_spill_reloads(text, min_depth=2) <= most, (name, ki.loop_spill_reloads(text, min_depth=2))


def test_last_barrier_of_a_workgroup_does_not_wait_for_its_stores():
  """bsx_final_barrier (csrc/bsx_device.h): the barrier in front of bsx_flush_counts orders the waves' LDS counter
  updates only — `s_waitcnt lgkmcnt(0)` + `s_barrier` — where __syncthreads() made every wave sit through the
  acknowledgements of its final stores (`s_waitcnt vmcnt(0)`) before it could retire."""
  import kernel_isa as ki
  for src, want in (('mountain_car.hip', 'small_obs_kernel<mountain_car_env, false, 0, 0, 0, true, false>'),
                    ('cartpole.hip', 'small_obs_kernel<cartpole_env, false, 0, 0, 0, true, false>'),
                    ('deep_sea.hip', 'bsx_advance_kernel<deep_sea_fam, true, -1>')):
    name, text = ki.kernel_text(os.path.join(ROOT, 'bsuite_amd', 'csrc', src), want)
    ins = [l.split(';')[0].strip() for l in text if l.strip() and not l.strip().startswith((';', '.'))]
    bars = [k for k, l in enumerate(ins) if l.startswith('s_barrier')]
    assert bars, name
    last = bars[-1]
    assert ins[last - 1].startswith('s_waitcnt') and 'lgkmcnt(0)' in ins[last - 1] and 'vmcnt' not in ins[last - 1], (name, ins[last - 3:last + 1])
    assert any(l.startswith('global_store') for l in ins[:last]), name      # ... with stores issued before it


def test_sweep_kernels_spill_nothing(kernels):
  """sweep_phase0_kernel spilled 157 scalar registers (sweep_pipelined_kernel 58) under the 80-SGPR cap that
  amdgpu_waves_per_eu(8) brings — the attribute held a kernel that compiled the MT19937-exact generators into every
  wrapped segment's body to 64 VGPRs.  The whole-sweep launches are now compiled without those generators (the group
  refuses MT19937-exact segments): 46 VGPRs without the attribute, no spill of either kind."""
  for n in ('sweep_phase0_kernel', 'sweep_pipelined_kernel'):
    k = kernels[n]
    assert k['sgpr_spill_count'] == 0 and k['vgpr_spill_count'] == 0 and k['private_segment_fixed_size'] == 0, (n, k)
    assert k['vgpr_count'] <= 64, (n, k['vgpr_count'])


def test_wrapped_rollouts_keep_four_waves_per_simd(kernels):
  """the fused rollouts a Logging- or RewardNoise-wrapped environment runs (every *_noise id, every
  recorded run) were 157-218 VGPRs = 2 waves per SIMD: the ~60 f64 constants of the normal transform hoisted out of the
  step loop into VGPR pairs, and the MT19937-exact generator compiled into every one of them.  The constants are now
  materialised on the scalar unit where they are used (BSX_K, include/bsx_stream.h) and the counter-based runs have
  instantiations of their own (MT = 0): at most 128 VGPRs, no scratch."""
  wrapped = {n: k for n, k in kernels.items()
             if n.startswith('small_obs_kernel<') and re.search(r', true, (1, 0|0, 1|1, 1), 0, (true|false), false>$', n)}
  assert len(wrapped) >= 15, sorted(wrapped)
  for n, k in wrapped.items():
    assert k['vgpr_count'] <= 128 and k['private_segment_fixed_size'] == 0, (n, k['vgpr_count'], k['private_segment_fixed_size'])
  for n in ('bsx_fused_rollout_kernel<catch_fam, false, catch_hot>', 'bsx_fused_rollout_kernel<deep_sea_fam, false, deep_sea_hot>'):
    assert kernels[n]['vgpr_count'] <= 128, (n, kernels[n]['vgpr_count'])


def test_lean_instantiations_keep_full_occupancy(kernels):
  """The common call (no Logging wrapper, no RewardNoise, counter-based draws) of every eager small-observation
  kernel fits 8 waves per SIMD."""
  lean = [k for n, k in kernels.items() if n.startswith('small_obs_kernel<') and ', false, 0, 0, 0,' in n]
  assert len(lean) >= 8
  for k in lean:
    assert k['vgpr_count'] <= 64, (k['name'], k['vgpr_count'])


def test_non_temporal_stores_are_where_they_were_measured_to_pay():
  """Round 6: non-temporal stores pay where a wave's store instruction covers one contiguous range of outputs that
  nobody reads again while the kernel's own INPUTS compete for the cache — and cost 8-40 % elsewhere.  From the ISA:
  the stand-alone one-hot stream and deep_sea's single-launch step (scattered writer threads) have none; the sweep's mixed
  stream, the fused rollouts' rows and cartpole's eager step (16-byte chunks through the wave's LDS) have them; cartpole never
  stores an 8-byte row piece non-temporally (partial lines: 18 -> 25 us)."""
  import kernel_isa as ki
  csrc = os.path.join(ROOT, 'bsuite_amd', 'csrc')

  def stores(src, want):
    _, text = ki.kernel_text(os.path.join(csrc, src), want)
    ins = [l.split(';')[0].strip() for l in text]
    return [l for l in ins if l.startswith(('global_store', 'flat_store'))]

  for src, want in (('deep_sea.hip', 'deep_sea_step1_kernel<4>'), ('deep_sea.hip', 'bsx_hot_stream_kernel<deep_sea_hot, 4, 256>'),
                    ('catch.hip', 'bsx_hot_stream_kernel<catch_hot, 2, 256>'), ('deep_sea.hip', 'bsx_advance_kernel<deep_sea_fam, true, -1>')):
    st = stores(src, want)
    assert st and not any(l.endswith(' nt') for l in st), (want, [l for l in st if l.endswith(' nt')])
  st = stores('pair_mixed.hip', 'pair_mixed_stream_kernel')
  assert sum(l.startswith('flat_store_dwordx4') and l.endswith(' nt') for l in st) >= 6          # deep_sea 4 + catch 2 chunks (flat: the pointers come from the argument table)
  st = stores('mountain_car.hip', 'small_obs_lean_rollout_kernel<mountain_car_env, false, 0, true>')
  assert any(l.startswith('global_store_dwordx3') and l.endswith(' nt') for l in st)
  assert not any(l.startswith('global_store_byte') and l.endswith(' nt') for l in st)              # its scalars stay ordinary
  st = stores('bandit.hip', 'small_obs_lean_rollout_kernel<bandit_env, false, 0, false>')
  assert any(l.startswith('global_store_byte') and l.endswith(' nt') for l in st)
  # an EAGER step's outputs are write-through (sc1: an agent reads them next), never non-temporal
  st = stores('cartpole.hip', 'small_obs_kernel<cartpole_env, false, 0, 0, 0, true, false>')
  assert any(l.startswith('global_store_dwordx4') and l.endswith(' sc1') for l in st)
  assert not any(l.endswith(' nt') for l in st)
  st = stores('bandit.hip', 'small_obs_eager2_kernel<bandit_env, 0, 2>')
  assert any(l.endswith(' sc1') for l in st) and not any(l.endswith(' nt') for l in st)
  # catch's single-launch steps (64- and 256-lane tiles): write-through observation chunks; its fused rollout: non-temporal
  for want in ('bsx_fused_tile_kernel<catch_fam, true, catch_hot>', 'bsx_fused_tile64_kernel<catch_fam, true, catch_hot>'):
    st = stores('catch.hip', want)
    assert any(l.startswith('global_store_dwordx4') and l.endswith(' sc1') for l in st) and not any(l.endswith(' nt') for l in st), want
  st = stores('catch.hip', 'bsx_fused_rollout_kernel<catch_fam, true, catch_hot>')
  assert any(l.startswith('global_store_dwordx4') and l.endswith(' nt') for l in st)
  st = stores('sweep_mixed.hip', 'sweep_phase0_kernel')
  assert any(l.endswith(' sc1') for l in st)                                                        # the sweep's catch tiles
