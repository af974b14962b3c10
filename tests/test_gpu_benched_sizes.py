"""GPU: parity AT THE SIZES bench.py measures.

* BASELINE config 5 exactly as benched — all 468 bsuite_ids, 2^20 lanes, whole-sweep group, the
  software-pipelined one-launch schedule AND both closed-loop two-launch schedules (phase 0 | stream, and the split cut
  of bsx_group_step_split), the action ring the bench feeds
  (bsx_call_t.action_ring) — 50 sweep steps: 8 sampled lanes of every segment against the C oracle
  (bit-exact for the integer / grid families), every lane of every segment against a stand-alone
  environment of that bsuite_id stepped eagerly, and `summary()` against per-segment stand-alone
  summaries, exactly.  The big-grid paths are what the 257-lane tests cannot reach: a workgroup map of
  7000+ entries, a retirement ticket over 4213 workgroups, 64-bit magic division at float offsets > 2^32.
* The pipelined rollout() of deep_sea N=30 / B=2^20 / T=16 and catch / B=2^20 / T=32 (the `rollout16` /
  `rollout32` records): all-lane invariants + a 4096-lane subsample against the oracle on every step,
  across an episode end, twice in a row (the state-column parity).
* cartpole at B=2^20 for more than one full episode, teacher-forced on a 65 536-lane subsample.
"""
import numpy as np
import pytest
import torch

import bsuite_amd
from bsuite_amd import sweep
from bsuite_amd import sweep_batch as sb
from oracle import coracle
from tests import engine_util as eu
from tests import golden_util as gu
from tests.test_gpu_all_ids import oracle_config

pytestmark = pytest.mark.gpu
B = 1 << 20
PHYSICS = ('cartpole', 'cartpole_swingup', 'mountain_car')


@pytest.mark.timeout(900)
@pytest.mark.parametrize('pipelined', [True, False, 'split'])
def test_config5_full_sweep_as_benched(tmp_path, pipelined):
  split, pipelined = pipelined == 'split', pipelined is True
  from bsuite_amd.utils import datasets
  from bsuite_amd import distributed as bdist
  imgs, labels = gu.mnist_dataset()
  datasets.write_idx_files(str(tmp_path), imgs.view(np.uint8), labels)
  mn = dict(data_dir=str(tmp_path))
  kw = dict(mnist=mn, mnist_noise=mn, mnist_scale=mn)
  seed, steps, ring, n_samp = 42, 50, 16, 8
  batch = sb.SweepBatch(None, B, seed=seed, env_kwargs=kw)
  assert len(batch.envs) == 468 and batch.lanes() == B
  acts = batch.random_actions(seed=1, ring=ring)
  outs = batch.prepare_groups(acts, pipelined=pipelined, split=split)
  assert len(batch._groups) == (2 if pipelined else 1) and batch._split == split

  # the checker: 8 lanes of every segment through the C oracle, configured from the reference's experiment files
  rng = np.random.default_rng(5)
  orcs = []
  for (bid, begin, lanes), env in zip(batch.segments, batch.envs):
    fam, okw, wrap, _ = oracle_config(bid)     # (a SweepBatch keys every segment's draw stream by ITS seed)
    okw = dict(okw)
    if fam == 'mnist':
      okw.update(images=imgs, labels=labels)
    idx = np.unique(np.concatenate([[0, lanes - 1], rng.integers(0, lanes, size=n_samp - 2)]))
    orc = coracle.OracleEnv(fam, okw, (begin + idx).astype(np.uint64), seed=seed, wrap=wrap)
    chk = eu.PhysicsChecker(fam, okw, len(idx)) if fam in PHYSICS else None
    orcs.append((fam, idx, torch.from_numpy(idx).to(batch.device), orc, np.ones(len(idx), bool), chk))

  for s in range(steps):
    out = batch.step_grouped()
    if pipelined:
      assert out is outs[s & 1]
    for (bid, _, _), a, ts, env, (fam, idx, idx_t, orc, same, chk) in zip(batch.segments, acts, out, batch.envs, orcs):
      st, r, d, o = orc.call(a[s % ring][idx_t].cpu().numpy(), s)
      gst, gr, gd, go = (x[idx_t].cpu().numpy() for x in (ts.step_type, ts.reward, ts.discount, ts.observation))
      live = st != 0
      if fam in PHYSICS and not pipelined:
        # closed loop: the state columns are plain tensors the group's argument slots point at, so the sampled lanes
        # are TEACHER-FORCED between step_grouped() calls like every other physics test — f32(reference f64 state)
        # in, one call, 1e-6 * max(1, |b|) on every continuous component (a verified threshold tie waives the
        # discrete fields only: PhysicsChecker)
        chk.check((gst, gr, gd, go), (st, r, d, o), eu.oracle_physics_state(orc, fam), msg=f'{bid} s={s}')
        eu.teacher_force(eu.raw(env), orc, fam, lanes=idx_t)
      elif fam in PHYSICS:
        # pipelined: the lanes are already one advance ahead when a step's TimeSteps come back, nothing can be forced;
        # free-running f32 engine vs the f64 oracle — the lanes whose step types still agree stay within the drift
        # profiles/HISTORY.md §5 reports for a few dozen calls (the all-lane bit-equality with a stand-alone
        # environment below is the pin)
        same &= gst == st
        assert same.mean() > 0.7, bid
        assert np.abs(go[same].astype(np.float64) - o[same]).max() <= 2e-4, (bid, s)
      else:
        np.testing.assert_array_equal(gst, st, err_msg=f'{bid} s={s}')
        np.testing.assert_array_equal(eu.f32_bits(go), eu.f32_bits(o), err_msg=f'{bid} obs s={s}')
        np.testing.assert_array_equal(eu.f32_bits(gr[live]), eu.f32_bits(r[live].astype(np.float32)), err_msg=f'{bid} s={s}')
        np.testing.assert_array_equal(gd[live], d[live].astype(np.float32), err_msg=f'{bid} s={s}')
  batch.sync()
  ahead = 1 if pipelined else 0
  assert all(eu.raw(e).step_index == steps + ahead for e in batch.envs)
  summ = batch.summary()

  # every lane of every segment == the stand-alone environment of that id (same kernels, another launch path)
  total_last = 0
  forced = {bid: idx for (bid, _, _), (fam, idx, _, _, _, _) in zip(batch.segments, orcs) if fam in PHYSICS and not pipelined}
  for _, _, _, _, _, chk in orcs:
    if chk is not None and not pipelined:
      chk.assert_few_ties()
  for (bid, begin, lanes), a, ts, env in zip(batch.segments, acts, out, batch.envs):
    name = bid.split('/')[0]
    ekw = dict(kw.get(name, {}))
    if sweep.SETTINGS[bid].get('seed', 0) is None or 'seed' not in sweep.SETTINGS[bid]:
      ekw['seed'] = seed
    ref = bsuite_amd.load_from_id(bid, batch=lanes, lane_offset=begin, num_buffers=1, **ekw)
    for s in range(steps):
      rts = ref.step(a[s % ring])
    keep = np.ones(lanes, bool)
    if forced.get(bid) is not None:
      keep[forced[bid]] = False               # (the teacher-forced sample lanes no longer follow the free-running engine)
    for x, y in zip(eu.to_np(ts), eu.to_np(rts)):
      np.testing.assert_array_equal(x[keep], y[keep], err_msg=bid)
    for s in range(steps, steps + ahead):
      ref.step(a[s % ring])
    keep_t = torch.from_numpy(keep).to(batch.device)
    for k, v in ref.bsuite_info().items():
      torch.testing.assert_close(env.bsuite_info()[k][..., keep_t], v[..., keep_t], rtol=0, atol=0)
    if forced.get(bid) is None:
      # (a segment with teacher-forced sample lanes: its sums are those of slightly different trajectories; the
      # pipelined parametrisation of this test holds every id to these two exact comparisons)
      torch.testing.assert_close(eu.raw(env).episode_counters(), eu.raw(ref).episode_counters(), rtol=0, atol=0)
      vec, names = bdist.local_summary(ref)
      assert summ[bid] == dict(zip(names, vec.tolist())), bid        # exact, per segment
    total_last += summ[bid]['episodes_finished']
    del ref
  assert total_last > B                                              # the bandits alone finish 25 episodes per lane
  batch.release_groups()


def _subsample(rng, n=4096, B=B):
  idx = np.unique(np.concatenate([rng.integers(0, B, size=n), [0, 1, 63, 64, 255, 256, B - 1]]))
  return idx.astype(np.int64)


@pytest.mark.timeout(900)
@pytest.mark.parametrize('family,kwargs,na,T,warm,B', [('deep_sea', dict(size=30, mapping_seed=42), 2, 16, 20, 1 << 20),
                                                       ('catch', dict(), 3, 32, 3, 1 << 20),
                                                       ('catch', dict(), 3, 32, 3, 1 << 19)])
def test_pipelined_rollout_as_benched(family, kwargs, na, T, warm, B):
  """`rollout(actions[T, B])` of the two-kernel families at the benched sizes: software-pipelined at 2^20 lanes (T+1
  launches, two state columns); catch at 2^19 lanes (a rank's share of a 2-GPU strong-scaled run) is the largest batch
  that takes the fused ONE-launch rollout (bsx_fused_rollout_kernel).  After `warm` eager steps, two consecutive
  rollouts — deep_sea's cross the episode end at call 31 — compared on a 4096-lane subsample with the oracle on every
  step, plus size-independent invariants over all lanes."""
  seed = 11
  env = eu.make_env(family, kwargs, batch=B, lane_offset=0, seed=seed, num_buffers=1)
  assert eu.raw(env)._pipelined_rollout
  rng = np.random.default_rng(3)
  idx = _subsample(rng, B=B)
  idx_t = torch.from_numpy(idx).cuda()
  orc = coracle.OracleEnv(family, kwargs, idx.astype(np.uint64), seed=seed)
  g = torch.Generator(device='cuda'); g.manual_seed(9)
  cells = int(np.prod(env.observation_spec().shape))
  call = 0
  for _ in range(warm):
    a = torch.randint(na, (B,), generator=g, device='cuda', dtype=torch.int32)
    env.step(a)
    orc.call(a[idx_t].cpu().numpy(), call)
    call += 1
  lasts = 0
  for _ in range(2):
    acts = torch.randint(na, (T, B), generator=g, device='cuda', dtype=torch.int32)
    ro = env.rollout(acts)
    for t in range(T):
      obs = ro.observation[t].view(B, cells)
      s = obs.sum(dim=1)
      last = ro.step_type[t] == 2
      assert bool(((obs == 0) | (obs == 1)).all())
      if family == 'deep_sea':
        assert bool(((s == 1) ^ last).all())                         # one-hot, all-zero terminal board
      else:
        assert bool(((s == 1) | (s == 2)).all()) and bool((obs[:, 45:].sum(dim=1) >= 1).all())
      assert bool((ro.discount[t] == (~last).float()).all())
      lasts += int(last.sum())
      st, r, d, o = orc.call(acts[t][idx_t].cpu().numpy(), call)
      call += 1
      np.testing.assert_array_equal(ro.step_type[t][idx_t].cpu().numpy(), st, err_msg=f't={t}')
      live = st != 0
      np.testing.assert_array_equal(eu.f32_bits(ro.reward[t][idx_t].cpu().numpy()[live]), eu.f32_bits(r[live].astype(np.float32)))
      np.testing.assert_array_equal(ro.observation[t][idx_t].cpu().numpy().reshape(len(idx), -1), o.reshape(len(idx), -1),
                                    err_msg=f't={t}')
    del ro
  assert lasts >= B                                                  # every lane ended an episode inside the rollouts
  for k, v in orc.bsuite_info().items():
    np.testing.assert_array_equal(env.bsuite_info()[k][idx_t].cpu().numpy(), v)
  # the state left behind (in the environment's own column) is the oracle's: one more eager step agrees
  a = torch.randint(na, (B,), generator=g, device='cuda', dtype=torch.int32)
  ts = env.step(a)
  st, r, d, o = orc.call(a[idx_t].cpu().numpy(), call)
  np.testing.assert_array_equal(ts.step_type[idx_t].cpu().numpy(), st)
  np.testing.assert_array_equal(ts.observation[idx_t].cpu().numpy().reshape(len(idx), -1), o.reshape(len(idx), -1))


@pytest.mark.timeout(900)
@pytest.mark.parametrize('family,horizon', [('cartpole', 260), ('mountain_car', 40)])
def test_physics_full_batch_full_episode_teacher_forced(family, horizon):
  """BASELINE config 4 at its size: B = 2^20 lanes stepped eagerly; the first 65 536 lanes are teacher-forced from
  the f64 oracle on every call and compared at 1e-6*max(1,|b|) — for cartpole until (nearly) every one of them has
  finished an episode and been auto-reset (random-policy episodes last ~80 calls), mountain_car for 40 calls."""
  n, seed = 1 << 16, 3
  env = eu.make_env(family, {}, batch=B, lane_offset=0, seed=seed, num_buffers=1)
  raw = eu.raw(env)
  orc = coracle.OracleEnv(family, {}, np.arange(n, dtype=np.uint64), seed=seed)
  chk = eu.PhysicsChecker(family, {}, n)
  g = torch.Generator(device='cuda'); g.manual_seed(5)
  finished = np.zeros(n, bool)
  restarted = np.zeros(n, bool)
  for t in range(horizon):
    a = torch.randint(3, (B,), generator=g, device='cuda', dtype=torch.int32)
    if t > 0:           # device f32 state of the first n lanes := f32(reference f64 state)
      if family == 'mountain_car':
        st32 = np.stack([orc.s['position'], orc.s['velocity']]).astype(np.float32)
        k = orc.s['timestep'].astype(np.int32)
      else:
        st32 = orc.s['state'][:, :4].T.astype(np.float32)
        k = np.rint(orc.s['state'][:, 4] / orc.cfg.timescale).astype(np.int32)
      raw._state['state'][:, :n].copy_(torch.from_numpy(np.ascontiguousarray(st32)).cuda())
      raw._state['steps'][:n].copy_(torch.from_numpy(k | (orc.reset_next.astype(np.int32) << 30)).cuda())
    ts = env.step(a)
    want = tuple(x.copy() for x in orc.call(a[:n].cpu().numpy(), t))
    got = tuple(x[:n].cpu().numpy() for x in (ts.step_type, ts.reward, ts.discount, ts.observation))
    chk.check(got, want, eu.oracle_physics_state(orc, family), msg=f'{family} t={t}')
    restarted |= finished & (want[0] == 0)
    finished |= want[0] == 2
    if t % 16 == 15:
      assert bool(torch.isfinite(ts.observation).all())
  chk.assert_few_ties()
  assert chk.max_err['observation'] <= eu.PHYS_TOL
  if family == 'cartpole':
    assert restarted.mean() > 0.97, restarted.mean()                 # a full episode + auto-reset on (nearly) all of them
    info = env.bsuite_info()
    clean = ~chk.tainted
    for k_, v in orc.bsuite_info().items():
      np.testing.assert_array_equal(info[k_][:n].cpu().numpy()[clean], v[clean], err_msg=k_)


@pytest.mark.timeout(1500)
def test_config5_eight_shards_equal_the_unsharded_sweep(tmp_path):
  """BASELINE config 5 as the driver's 8-GPU run shards it: SweepBatch(rank=r, world_size=8) for r = 0..7, one after
  another on the one GPU there is here, each stepping ITS bin-packed eighth of the 468 ids as one whole-sweep group —
  every TimeStep field, bsuite_info column and episode counter of every segment bit-identical to the unsharded sweep
  at 2^20 lanes (draws and actions are keyed by global lane id / segment index: bsuite_amd/sweep_batch.py).  The
  shards partition the ids, and their loads (lanes x bytes per step) are within 1 % of each other."""
  from bsuite_amd.utils import datasets
  imgs, labels = gu.mnist_dataset()
  datasets.write_idx_files(str(tmp_path), imgs.view(np.uint8), labels)
  mn = dict(data_dir=str(tmp_path))
  kw = dict(mnist=mn, mnist_noise=mn, mnist_scale=mn)
  seed, steps, ring, world = 42, 21, 16, 8

  def run(rank, world_size):
    batch = sb.SweepBatch(None, B, seed=seed, env_kwargs=kw, rank=rank, world_size=world_size)
    acts = batch.random_actions(seed=1, ring=ring)
    outs = batch.prepare_groups(acts)
    for _ in range(steps):
      batch.step_grouped()
    batch.sync()
    res = {}
    for (bid, _, lanes), ts, env in zip(batch.segments, outs, batch.envs):
      info = env.bsuite_info()
      res[bid] = dict(ts=tuple(x.clone() for x in (ts.step_type, ts.reward, ts.discount, ts.observation)),
                      info={k: v.clone() for k, v in info.items()}, counters=eu.raw(env).episode_counters().clone(), lanes=lanes)
    load = sum(l * sb.bytes_per_step(int(np.prod(e.observation_spec().shape))) for e, (_, _, l) in zip(batch.envs, batch.segments))
    batch.release_groups()
    return res, load

  whole, total_load = run(0, 1)
  assert len(whole) == 468
  seen, loads = set(), []
  for r in range(world):
    part, load = run(r, world)
    loads.append(load)
    assert part and not (set(part) & seen)
    seen |= set(part)
    for bid, got in part.items():
      want = whole[bid]
      for x, y in zip(got['ts'], want['ts']):
        assert torch.equal(x, y), (r, bid)
      for k in want['info']:
        assert torch.equal(got['info'][k], want['info'][k]), (r, bid, k)
      assert torch.equal(got['counters'], want['counters']), (r, bid)
    del part
    torch.cuda.empty_cache()
  assert seen == set(whole) and sum(loads) == total_load
  assert max(loads) / (total_load / world) < 1.01, loads


@pytest.mark.timeout(900)
def test_split_step_of_another_family_mix_equals_the_unsplit_step_and_is_not_slower():
  """The split closed-loop step (bsx_group_step_split) tops its first launch up to one dispatch round DERIVED from the
  device and the group (CUs x resident workgroups of sweep_phase0_kernel with this group's LDS, + 18 %: csrc/sweep_mixed.hip),
  not to a constant tuned on BASELINE config 5: a sweep of a different composition — no mnist, no deep_sea above size 20,
  every wide-row chain id — at 2^20 lanes (phase 0 = 4100+ workgroups, two dispatch rounds) gives the unsplit step's
  TimeSteps, bsuite_info and counters bit for bit, and does not take longer than it."""
  ids = [b for b in sweep.SWEEP if not b.startswith('mnist') and not (b.startswith('deep_sea') and int(b.split('/')[1]) > 5)]
  assert 360 < len(ids) < 400
  seed, steps, ring = 7, 19, 16

  def run(split):
    batch = sb.SweepBatch(ids, B, seed=seed)
    acts = batch.random_actions(seed=3, ring=ring)
    outs = batch.prepare_groups(acts, split=split)
    assert batch._split == split
    for _ in range(steps):
      batch.step_grouped()
    batch.sync()
    assert batch._split == split                      # (the split cut applies: no fallback to the ordinary step)
    res = {}
    for (bid, _, _), ts, env in zip(batch.segments, outs, batch.envs):
      res[bid] = (tuple(x.clone() for x in (ts.step_type, ts.reward, ts.discount, ts.observation)),
                  {k: v.clone() for k, v in env.bsuite_info().items()}, eu.raw(env).episode_counters().clone())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = float('inf')
    for _ in range(5):
      e0.record()
      for _ in range(40):
        batch.step_grouped()
      e1.record()
      torch.cuda.synchronize()
      best = min(best, e0.elapsed_time(e1) / 40)
    batch.release_groups()
    return res, best

  plain, t_plain = run(False)
  torch.cuda.empty_cache()
  cut, t_cut = run(True)
  for bid, (ts, info, counters) in plain.items():
    for x, y in zip(cut[bid][0], ts):
      assert torch.equal(x, y), bid
    for k, v in info.items():
      assert torch.equal(cut[bid][1][k], v), (bid, k)
    assert torch.equal(cut[bid][2], counters), bid
  assert t_cut <= 1.03 * t_plain, (t_cut, t_plain)
