"""GPU: rollout(actions[T,B]) == T consecutive step() calls, bit for bit, for every family —
including the fused T-step kernels of the small-observation families, odd observation widths
(unaligned [t] slices), wrappers and the Logging bookkeeping.

This is ENGINE vs ENGINE on purpose: `step()` is what the oracle, the golden fixtures and the live reference pin
(tests/test_gpu_oracle_batch.py, test_gpu_golden.py, test_gpu_vs_reference_live.py), and bit-equality of the fused
rollouts with T such calls is a stronger statement than any tolerance against the f64 oracle could be — for the
physics families in particular, whose free-running f32 trajectories leave the 1e-6 band of a free-running f64
reference within a few dozen calls."""
import numpy as np
import pytest
import torch

from bsuite_amd.utils import wrappers
from tests import engine_util as eu
from tests import golden_util as gu

pytestmark = pytest.mark.gpu

CASES = [
    ('bandit', dict(mapping_seed=2), None, 11),
    ('bandit', dict(mapping_seed=2), ('noise', 0.5), 11),
    ('memory_chain', dict(memory_length=3, num_bits=3), None, 2),      # numel 5: unaligned slices
    ('memory_chain', dict(memory_length=2, num_bits=40), None, 2),     # packed records, 42-float rows
    ('memory_chain', dict(memory_length=12, num_bits=1), None, 2),     # memory_len/10: the register-resident rollout, variant 0
    ('memory_chain', dict(memory_length=9, num_bits=1), ('noise', 0.3), 2),
    ('memory_chain', dict(memory_length=1, num_bits=6), ('scale', 2.0), 2),     # 8 floats: the widest row a thread stores itself
    ('umbrella_chain', dict(chain_length=4, n_distractor=20), ('scale', 3.0), 2),
    ('discounting_chain', dict(mapping_seed=1), None, 5),
    ('cartpole', dict(), None, 3),
    ('cartpole_swingup', dict(), None, 3),
    ('mountain_car', dict(max_steps=15), None, 3),
    ('deep_sea', dict(size=6, deterministic=False, mapping_seed=1), None, 2),
    ('catch', dict(rows=5, columns=3), None, 3),
    ('catch', dict(), None, 3),        # 50 cells x odd B: slices t*B*50 floats in are not 16-byte aligned
    ('mnist', dict(), None, 10),
]


@pytest.mark.parametrize('family,kwargs,wrap,na', CASES)
@pytest.mark.parametrize('batch', [1, 1000, 333])
def test_rollout_equals_steps(family, kwargs, wrap, na, batch):
  kw = dict(kwargs)
  if family == 'mnist':
    kw['images'], kw['labels'] = gu.mnist_dataset()
  T, seed = 37, 21
  g = torch.Generator(device='cuda'); g.manual_seed(3)
  acts = torch.randint(na, (T, batch), generator=g, device='cuda', dtype=torch.int32)
  a = eu.make_env(family, kw, batch=batch, lane_offset=9, seed=seed, wrap=wrap)
  b = eu.make_env(family, kw, batch=batch, lane_offset=9, seed=seed, wrap=wrap)
  la = wrappers.Logging(a, None)
  lb = wrappers.Logging(b, None)
  warm = acts[:5]
  for t in range(5):                          # some ordinary steps first (state carried into the rollout)
    la.step(warm[t]); lb.step(warm[t])
  ro = la.rollout(acts)
  for t in range(T):
    ts = lb.step(acts[t])
    for x, y in zip((ro.step_type[t], ro.reward[t], ro.discount[t], ro.observation[t]),
                    (ts.step_type, ts.reward, ts.discount, ts.observation)):
      np.testing.assert_array_equal(x.cpu().numpy(), y.cpu().numpy(), err_msg=f'{family} t={t}')
  after_a, after_b = la.step(acts[0]), lb.step(acts[0])     # and the state left behind is the same
  np.testing.assert_array_equal(after_a.observation.cpu().numpy(), after_b.observation.cpu().numpy())
  for k, v in eu.raw(a).bsuite_info().items():
    torch.testing.assert_close(v, eu.raw(b).bsuite_info()[k], rtol=0, atol=0)
  for k, v in la.counters().items():
    torch.testing.assert_close(v, lb.counters()[k], rtol=0, atol=0)
  torch.testing.assert_close(la.num_rows(), lb.num_rows(), rtol=0, atol=0)
  torch.testing.assert_close(eu.raw(a).episode_counters(), eu.raw(b).episode_counters(), rtol=0, atol=0)
  assert eu.raw(a).step_index == eu.raw(b).step_index == 5 + T + 1


@pytest.mark.parametrize('family,na', [('cartpole', 3), ('cartpole_swingup', 3), ('mountain_car', 3)])
@pytest.mark.parametrize('batch,T', [(1, 40), (333, 300), (5000, 300), (70000, 64), ((1 << 19) + 1000, 12), ((1 << 19) + 333, 12),
                                     (1 << 20, 16)])        # the last one: as benched (`cartpole/0 r16`)
def test_lean_fused_rollout_equals_steps(family, na, batch, T):
  """No wrapper at all: the lean instantiation of the fused rollout.  From 2^19 lanes up (2048 workgroups) the launch
  takes the BIG variant, in which cartpole's resetting lanes hand their draws to the workgroup's pool and full waves
  stage their rows in LDS (small_obs.h; an odd batch keeps row-per-lane stores, both have a ragged last workgroup);
  test_big_launch_variant_at_small_shapes runs the small shapes through that variant too.  The first rollout starts
  on a fresh environment — all 256 lanes of every workgroup reset at once — the second one carries state, and
  episodes end at every step of it."""
  seed = 5
  g = torch.Generator(device='cuda'); g.manual_seed(11)
  acts = torch.randint(na, (2 * T, batch), generator=g, device='cuda', dtype=torch.int32)
  a = eu.make_env(family, {}, batch=batch, lane_offset=7, seed=seed)
  b = eu.make_env(family, {}, batch=batch, lane_offset=7, seed=seed)
  n_first = 0
  for half in range(2):
    part = acts[half * T:(half + 1) * T]
    ro = a.rollout(part)
    got = [x.clone() for x in (ro.step_type, ro.reward, ro.discount, ro.observation)]
    for t in range(T):
      ts = b.step(part[t])
      for name, x, y in zip(('step_type', 'reward', 'discount', 'observation'), got,
                            (ts.step_type, ts.reward, ts.discount, ts.observation)):
        assert torch.equal(x[t], y), f'{family} {name} half={half} t={t}'
    n_first += int((got[0][1:] == 0).sum())
  if family == 'cartpole' and T >= 300:
    assert n_first > batch                                    # episodes did end, all over the rollout
  for k, v in a.bsuite_info().items():
    torch.testing.assert_close(v, b.bsuite_info()[k], rtol=0, atol=0)
  torch.testing.assert_close(a.episode_counters(), b.episode_counters(), rtol=0, atol=0)
  ta, tb = a.step(acts[0]), b.step(acts[0])
  assert torch.equal(ta.observation, tb.observation)


@pytest.mark.parametrize('family', ['cartpole', 'cartpole_swingup', 'mountain_car'])
@pytest.mark.parametrize('batch,T', [(1000, 40), ((1 << 19) + 256, 19)])
def test_lean_fused_rollout_with_actions_outside_the_action_spec(family, batch, T):
  """Any int32 is a legal action of these families (the reference computes (action - 1) * force with whatever it is
  given, cartpole.py:48 / mountain_car.py:79).  The lean fused rollout keeps a run of eight actions packed 4 bits each
  and re-reads a run from memory step by step as soon as some lane of the wave holds an action outside 0..15
  (small_obs_regs_rollout): here a third of the waves see such runs — negative, large and INT_MIN / INT_MAX actions
  sprinkled over [T, B] — and the result still equals T step() calls, which never pack."""
  g = torch.Generator(device='cuda'); g.manual_seed(4)
  acts = torch.randint(3, (T, batch), generator=g, device='cuda', dtype=torch.int32)
  odd = torch.tensor([-1, -7, 16, 15, 1000, -2**31, 2**31 - 1, 255], device='cuda', dtype=torch.int64)
  where = torch.rand((T, batch), generator=g, device='cuda') < 0.002
  pick = torch.randint(len(odd), (T, batch), generator=g, device='cuda')
  acts = torch.where(where, odd[pick].to(torch.int32), acts)
  acts[:, : batch // 2] = acts[:, : batch // 2].clamp(0, 2)         # ... and half of the batch stays in-spec (packed path)
  a = eu.make_env(family, {}, batch=batch, lane_offset=3, seed=8)
  b = eu.make_env(family, {}, batch=batch, lane_offset=3, seed=8)
  ro = a.rollout(acts)
  for t in range(T):
    ts = b.step(acts[t])
    for name, x, y in zip(('step_type', 'reward', 'discount', 'observation'),
                          (ro.step_type[t], ro.reward[t], ro.discount[t], ro.observation[t]),
                          (ts.step_type, ts.reward, ts.discount, ts.observation)):
      assert torch.equal(x, y, ), f'{family} {name} t={t}'
  for k, v in a.bsuite_info().items():
    torch.testing.assert_close(v, b.bsuite_info()[k], rtol=0, atol=0)
  torch.testing.assert_close(a.episode_counters(), b.episode_counters(), rtol=0, atol=0)


@pytest.mark.timeout(900)
def test_big_launch_variant_at_small_shapes():
  """The BIG variant of the register-resident fused rollout (pooled resets, rows staged in LDS) is what a launch of
  2048+ workgroups runs; here the small and ragged shapes of the parity tests go through it as well: a subprocess
  against the tuning build of the library with BSX_REGS_ROLLOUT_BIG_MIN_BLOCKS=1."""
  import os, subprocess, sys
  from bsuite_amd import build as _build
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ, BSX_NATIVE_LIB=_build.build(tuning=True), BSX_REGS_ROLLOUT_BIG_MIN_BLOCKS='1', PYTHONPATH=root)
  p = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', '-m', 'gpu', '-p', 'no:cacheprovider',
                      'tests/test_gpu_rollout.py', 'tests/test_gpu_oracle_batch.py', 'tests/test_gpu_golden.py',
                      '-k', '(cartpole or mountain_car) and not big_launch'],
                     cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=800)
  tail = p.stdout[-3000:]
  assert p.returncode == 0, tail
  assert ' passed' in tail and 'failed' not in tail, tail


@pytest.mark.parametrize('family,kwargs,na', [('deep_sea', dict(size=6, deterministic=False, mapping_seed=1), 2),
                                               ('deep_sea', dict(size=10, mapping_seed=42), 2),
                                               ('catch', dict(), 3)])
@pytest.mark.parametrize('T', [2, 3, 8])
@pytest.mark.parametrize('logging', [False, True])
def test_pipelined_rollout_equals_steps(family, kwargs, na, T, logging):
  """deep_sea / catch rollouts are software-pipelined (bsx_call_t.state_alt): after the first advance every
  launch is {observation stream of step t, lane advance of step t+1}, the advances alternating between two
  state columns.  Even and odd T (the parity decides which column the first advance writes), with and without
  the Logging wrapper (lean / full advance kernel); the outputs, the state left behind and the counters are
  those of T step() calls."""
  batch, seed = 1024, 5                      # B*cells % 4 == 0: the pipelined path (else it falls back)
  g = torch.Generator(device='cuda'); g.manual_seed(T)
  acts = torch.randint(na, (3 * T + 1, batch), generator=g, device='cuda', dtype=torch.int32)
  a = eu.make_env(family, kwargs, batch=batch, lane_offset=3, seed=seed)
  b = eu.make_env(family, kwargs, batch=batch, lane_offset=3, seed=seed)
  if logging:
    a, b = wrappers.Logging(a, None), wrappers.Logging(b, None)
  assert eu.raw(a)._pipelined_rollout
  for r in range(3):                          # consecutive rollouts: the state column handed over is the right one
    ro = a.rollout(acts[r * T:(r + 1) * T])
    for t in range(T):
      ts = b.step(acts[r * T + t])
      for x, y in zip((ro.step_type[t], ro.reward[t], ro.discount[t], ro.observation[t]),
                      (ts.step_type, ts.reward, ts.discount, ts.observation)):
        np.testing.assert_array_equal(x.cpu().numpy(), y.cpu().numpy(), err_msg=f'{family} rollout {r} t={t}')
  assert eu.raw(a)._state_alt is not None
  np.testing.assert_array_equal(eu.raw(a).state_dict()['state'].cpu().numpy(), eu.raw(b).state_dict()['state'].cpu().numpy())
  ta, tb = a.step(acts[-1]), b.step(acts[-1])
  np.testing.assert_array_equal(ta.observation.cpu().numpy(), tb.observation.cpu().numpy())
  for k, v in eu.raw(a).bsuite_info().items():
    torch.testing.assert_close(v, eu.raw(b).bsuite_info()[k], rtol=0, atol=0)
  torch.testing.assert_close(eu.raw(a).episode_counters(), eu.raw(b).episode_counters(), rtol=0, atol=0)


@pytest.mark.parametrize('place', ['1', '2'])
def test_pipelined_rollout_other_placements(place):
  """BSX_PIPELINED_PLACE (an A/B knob: only the tuning build of the library reads it, bsuite_amd/build.py) = where the
  advance workgroups sit in the fused grid: the last workgroups, or spread evenly.  The default (first) is what every other test runs; the A/B placements must
  produce the same TimeSteps (catch at a batch whose stream has fewer / more workgroups than its advance)."""
  import subprocess
  import sys
  import os
  code = r'''
import numpy as np, torch
from tests import engine_util as eu
for family, kwargs, na, batch in (('catch', dict(), 3, 1024), ('catch', dict(rows=40, columns=40), 3, 2048),
                                  ('deep_sea', dict(size=9, mapping_seed=3), 2, 4096)):
  g = torch.Generator(device='cuda'); g.manual_seed(1)
  acts = torch.randint(na, (7, batch), generator=g, device='cuda', dtype=torch.int32)
  a = eu.make_env(family, kwargs, batch=batch, lane_offset=5, seed=11)
  b = eu.make_env(family, kwargs, batch=batch, lane_offset=5, seed=11)
  ro = a.rollout(acts)
  for t in range(7):
    ts = b.step(acts[t])
    for x, y in zip((ro.step_type[t], ro.reward[t], ro.discount[t], ro.observation[t]),
                    (ts.step_type, ts.reward, ts.discount, ts.observation)):
      np.testing.assert_array_equal(x.cpu().numpy(), y.cpu().numpy(), err_msg=f'{family} t={t}')
  np.testing.assert_array_equal(eu.raw(a).state_dict()['state'].cpu().numpy(), eu.raw(b).state_dict()['state'].cpu().numpy())
print('placements ok')
'''
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  from bsuite_amd import build as _build
  env = dict(os.environ, BSX_PIPELINED_PLACE=place, PYTHONPATH=root, BSX_NATIVE_LIB=_build.build(tuning=True),
             BSX_FUSED_TILE_MAX_CELLS='0')          # (small batches would otherwise take the fused one-launch step)
  p = subprocess.run([sys.executable, '-c', code], cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                     text=True, timeout=300)
  assert p.returncode == 0 and 'placements ok' in p.stdout, p.stdout[-2000:]


@pytest.mark.parametrize('kwargs', [dict(memory_length=12, num_bits=1), dict(memory_length=3, num_bits=1), dict(memory_length=5000, num_bits=1),
                                    dict(memory_length=4, num_bits=2), dict(memory_length=2, num_bits=5), dict(memory_length=7, num_bits=6)])
@pytest.mark.parametrize('batch,T', [(1, 40), (333, 90), (5000, 64), (1 << 20, 16)])      # the last one: as benched (`memory_len/10 r16`)
def test_memory_chain_register_resident_rollout_equals_steps(kwargs, batch, T):
  """No wrapper: the lean fused rollout of memory_chain with a short row (num_bits <= 6) keeps the lane's state word and
  context in registers for the T steps, the time fractions in an LDS table (memory_length 5000: no table, the division
  stays) and the bsuite_info columns in registers; == T step() calls bit for bit, state, info and counters included."""
  if batch == 1 << 20 and kwargs != dict(memory_length=12, num_bits=1):
    pytest.skip('full size: the benched shape only')
  g = torch.Generator(device='cuda'); g.manual_seed(5)
  acts = torch.randint(2, (T, batch), generator=g, device='cuda', dtype=torch.int32)
  a = eu.make_env('memory_chain', kwargs, batch=batch, lane_offset=3, seed=8)
  b = eu.make_env('memory_chain', kwargs, batch=batch, lane_offset=3, seed=8)
  for t in range(3):
    a.step(acts[t]); b.step(acts[t])
  for rep in range(2):                                       # twice: the state a rollout leaves is what the next one loads
    ro = a.rollout(acts)
    for t in range(T):
      ts = b.step(acts[t])
      if batch <= 5000 or t in (0, 1, T - 1):
        for x, y in zip((ro.step_type[t], ro.reward[t], ro.discount[t], ro.observation[t]),
                        (ts.step_type, ts.reward, ts.discount, ts.observation)):
          assert torch.equal(x, y), f'{kwargs} rep={rep} t={t}'
  for k, v in a.bsuite_info().items():
    torch.testing.assert_close(v, b.bsuite_info()[k], rtol=0, atol=0)
  for k in ('state', 'context'):
    assert torch.equal(eu.raw(a)._state[k], eu.raw(b)._state[k]), k
  torch.testing.assert_close(eu.raw(a).episode_counters(), eu.raw(b).episode_counters(), rtol=0, atol=0)


@pytest.mark.parametrize('family,kwargs,na', [('bandit', dict(mapping_seed=4), 11), ('discounting_chain', dict(mapping_seed=3), 5)])
@pytest.mark.parametrize('batch,T', [(1, 40), (333, 230), (5000, 64), (1 << 20, 16)])          # the last one: as benched (`... r16`)
def test_bandit_and_discounting_chain_register_resident_rollouts(family, kwargs, na, batch, T):
  """No wrapper: the lean fused rollouts of the bandit and of discounting_chain keep the lane's state word (and the
  bandit's regret column) in registers for the T steps; == T step() calls bit for bit — out-of-spec actions included,
  which both clamp and count (bandit.py:61, discounting_chain.py:80)."""
  g = torch.Generator(device='cuda'); g.manual_seed(6)
  acts = torch.randint(na, (T, batch), generator=g, device='cuda', dtype=torch.int32)
  if batch == 333:
    acts[3, ::7] = -2
    acts[4, 1::5] = 1 << 20                                   # outside 0..15: the run re-reads its actions step by step
    acts[T - 1, ::3] = na
  a = eu.make_env(family, kwargs, batch=batch, lane_offset=11, seed=2)
  b = eu.make_env(family, kwargs, batch=batch, lane_offset=11, seed=2)
  a.step(acts[0]); b.step(acts[0])
  for rep in range(2):
    ro = a.rollout(acts)
    for t in range(T):
      ts = b.step(acts[t])
      if batch <= 5000 or t in (0, 1, T - 1):
        for x, y in zip((ro.step_type[t], ro.reward[t], ro.discount[t], ro.observation[t]),
                        (ts.step_type, ts.reward, ts.discount, ts.observation)):
          assert torch.equal(x, y), f'{family} rep={rep} t={t}'
  for k, v in a.bsuite_info().items():
    torch.testing.assert_close(v, b.bsuite_info()[k], rtol=0, atol=0)
  assert torch.equal(eu.raw(a)._state['state'], eu.raw(b)._state['state'])
  torch.testing.assert_close(eu.raw(a).episode_counters(), eu.raw(b).episode_counters(), rtol=0, atol=0)
  torch.testing.assert_close(eu.raw(a).invalid_action_count(), eu.raw(b).invalid_action_count(), rtol=0, atol=0)
