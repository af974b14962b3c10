"""GPU: the single-launch deep_sea step (deep_sea_step1_kernel, csrc/deep_sea.hip; ABI v11 BSX_CALL_STATE_TAGGED) against
the C oracle — the path a deterministic, un-wrapped DeepSea of N >= 28 (N even) takes up to 2^18 lanes: the threads
of the observation store stream recompute the transition of the lane whose row they write, and bit 18 of the packed
state word (the parity of the next call's index) tells an already-advanced word from one that is not.

Covered: the smallest and largest eligible boards and the benched one, ragged / one-lane / several-thousand-lane
batches with a 64-bit lane offset, explicit reset() calls in mid-episode, a state_dict() taken at an ODD call index
and loaded into an environment at an EVEN one (the tags must be rewritten), lanes put at mixed episode phases through
load_state_dict (bench.stagger_phases does that), the device-resident call counter, and the fallback cases that must
NOT take the single launch (stochastic, odd N, a reward wrapper) — all bit-exact, `bsuite_info()` and the episode
counters included."""
import numpy as np
import pytest
import torch

from oracle import coracle
from tests import engine_util as eu

pytestmark = pytest.mark.gpu


def _check(ts, want, msg):
  st, r, d, o = want
  gst, gr, gd, go = eu.to_np(ts)
  np.testing.assert_array_equal(gst, st, err_msg=msg)
  live = st != 0
  np.testing.assert_array_equal(eu.f32_bits(gr[live]), eu.f32_bits(r[live].astype(np.float32)), err_msg='reward ' + msg)
  np.testing.assert_array_equal(gd[live], d[live].astype(np.float32), err_msg='discount ' + msg)
  assert (gr[~live] == 0).all() and (gd[~live] == 1).all(), msg
  np.testing.assert_array_equal(eu.f32_bits(go), eu.f32_bits(o), err_msg='observation ' + msg)


@pytest.mark.parametrize('size', [28, 30, 46, 64])
@pytest.mark.parametrize('batch,lane_offset', [(1, 0), (333, 5), (4099, (1 << 32) - 17)])
def test_single_launch_step_bit_exact(size, batch, lane_offset):
  kw = dict(size=size, mapping_seed=7)
  seed = 99
  env = eu.make_env('deep_sea', kw, batch=batch, lane_offset=lane_offset, seed=seed)
  orc = coracle.OracleEnv('deep_sea', kw, np.arange(lane_offset, lane_offset + batch, dtype=np.uint64), seed=seed)
  rng = np.random.default_rng(size * 1000 + batch)
  T = 2 * size + 9                                                 # two whole episodes and a bit
  n_last = n_first = 0
  for t in range(T):
    a = rng.integers(0, 2, size=batch).astype(np.int32)
    if t % 7 == 3:
      a[:] = 1 - (t & 1)                                           # runs of equal actions: lanes travel together
    force = t in (size // 2, size // 2 + 1, size + 5)
    ts = env.reset() if force else env.step(torch.from_numpy(a).cuda())
    want = orc.call(a, t, force_reset=force)
    _check(ts, want, f'N={size} B={batch} t={t}')
    n_last += int((want[0] == 2).sum()); n_first += int((want[0] == 0).sum())
  assert eu.raw(env)._call_desc.flags == 1                         # the Python class vouches for the tags
  info = env.bsuite_info()
  for k, v in orc.bsuite_info().items():
    np.testing.assert_array_equal(info[k].cpu().numpy(), v, err_msg=k)
  c = eu.raw(env).episode_counters().cpu().numpy()
  assert (int(c[0]), int(c[1])) == (n_last, n_first)              # the writers' wave ballots counted every LAST / FIRST once


def test_state_dict_moves_between_call_parities_and_phases():
  """A dict taken at an odd call index, loaded at an even one; and lanes assembled from different call counts."""
  kw = dict(size=30, mapping_seed=42)
  B, seed = 2000, 3
  a_env = eu.make_env('deep_sea', kw, batch=B, lane_offset=0, seed=seed)
  rng = np.random.default_rng(1)
  acts = [torch.from_numpy(rng.integers(0, 2, size=B).astype(np.int32)).cuda() for _ in range(80)]
  for t in range(7):                                               # 7 calls: the next index is odd
    a_env.step(acts[t])
  sd = a_env.state_dict()
  assert int((sd['state'] >> 18 & 1).sum()) == 0                   # a dict carries no tag
  b_env = eu.make_env('deep_sea', kw, batch=B, lane_offset=0, seed=seed)
  for t in range(4):                                               # b is at an even index with other states
    b_env.step(acts[40 + t])
  b_env.load_state_dict(sd)
  for t in range(7, 45):
    x, y = a_env.step(acts[t]), b_env.step(acts[t])
    for u, v in zip(eu.to_np(x), eu.to_np(y)):
      np.testing.assert_array_equal(u, v, err_msg=f't={t}')
  # mixed phases: lane i keeps the state it had after (i % 5) + 1 calls — what bench.stagger_phases does
  c_env = eu.make_env('deep_sea', kw, batch=B, lane_offset=0, seed=seed)
  final = None
  phase = torch.arange(B, device='cuda') % 5
  for k in range(5):
    c_env.step(acts[k])
    sdk = c_env.state_dict()
    if final is None:
      final = sdk
    else:
      for key, val in sdk.items():
        if torch.is_tensor(val) and val.dim() >= 1 and val.shape[-1] == B and key != '__counters':
          final[key] = torch.where(phase == k, val, final[key])
        else:
          final[key] = val
  c_env.load_state_dict(final)
  orc = coracle.OracleEnv('deep_sea', kw, np.arange(B, dtype=np.uint64), seed=seed)
  # the oracle is replayed lane group by lane group: group k has made k + 1 calls, then continues at index 5
  want_state = {}
  for k in range(5):
    o2 = coracle.OracleEnv('deep_sea', kw, np.arange(B, dtype=np.uint64), seed=seed)
    for t in range(k + 1):
      o2.call(acts[t].cpu().numpy(), t)
    want_state[k] = o2
  for t in range(5, 40):
    ts = c_env.step(acts[t])
    got = eu.to_np(ts)
    for k in range(5):
      want = want_state[k].call(acts[t].cpu().numpy(), t)
      sel = (np.arange(B) % 5) == k
      np.testing.assert_array_equal(got[0][sel], want[0][sel], err_msg=f'step_type group {k} t={t}')
      np.testing.assert_array_equal(eu.f32_bits(got[3][sel]), eu.f32_bits(want[3][sel]), err_msg=f'obs group {k} t={t}')
  del orc


def test_single_launch_under_a_device_step_counter_and_hip_graph():
  import bsuite_amd
  B = 3000
  env = bsuite_amd.load_from_id('deep_sea/10', batch=B, seed=5, device_step_counter=True, num_buffers=2)
  ref = bsuite_amd.load_from_id('deep_sea/10', batch=B, seed=5, num_buffers=2)
  g = torch.Generator(device='cuda'); g.manual_seed(2)
  acts = torch.randint(2, (6, B), generator=g, device='cuda', dtype=torch.int32)
  env.step(acts[0]); ref.step(acts[0])
  side = torch.cuda.Stream()
  side.wait_stream(torch.cuda.current_stream())
  graph = torch.cuda.CUDAGraph()
  with torch.cuda.stream(side):
    with torch.cuda.graph(graph, stream=side):
      with env.step_counter_deferred():
        for t in range(1, 6):                                      # FIVE steps per replay: the parity flips every replay
          out = env.step(acts[t])
  torch.cuda.current_stream().wait_stream(side)
  for rep in range(7):
    graph.replay()
    for t in range(1, 6):
      want = ref.step(acts[t])
    torch.cuda.synchronize()
    for u, v in zip(eu.to_np(out), eu.to_np(want)):
      np.testing.assert_array_equal(u, v, err_msg=f'replay {rep}')


@pytest.mark.parametrize('kw,wrap', [(dict(size=30, deterministic=False, mapping_seed=42), None),      # draws per step
                                     (dict(size=29, mapping_seed=42), None),                            # chunks straddle lanes
                                     (dict(size=30, mapping_seed=42), ('scale', 0.5))])                 # lean, scaled: single launch too
def test_neighbours_of_the_single_launch_path(kw, wrap):
  B, seed = 1500, 11
  env = eu.make_env('deep_sea', kw, batch=B, lane_offset=3, seed=seed, wrap=wrap)
  orc = coracle.OracleEnv('deep_sea', kw, np.arange(3, 3 + B, dtype=np.uint64), seed=seed, wrap=wrap)
  rng = np.random.default_rng(0)
  for t in range(70):
    a = rng.integers(0, 2, size=B).astype(np.int32)
    _check(env.step(torch.from_numpy(a).cuda()), orc.call(a, t), f'{kw} t={t}')
  info = env.bsuite_info()
  for k, v in orc.bsuite_info().items():
    np.testing.assert_array_equal(info[k].cpu().numpy(), v, err_msg=k)


def test_action_ring_with_tagged_states_takes_the_two_launch_path():
  """the single-launch kernel reads `action` as a plain [B] column, so a TAGGED call that walks an action
  ring (C-ABI callers, captured graphs) must not take it: row (call index mod R) of the ring is what every lane acts on."""
  kw = dict(size=30, mapping_seed=42)
  B, R, seed = 2500, 4, 8
  env = eu.make_env('deep_sea', kw, batch=B, lane_offset=0, seed=seed)
  orc = coracle.OracleEnv('deep_sea', kw, np.arange(B, dtype=np.uint64), seed=seed)
  g = torch.Generator(device='cuda'); g.manual_seed(5)
  ring = torch.randint(2, (R, B), generator=g, device='cuda', dtype=torch.int32)
  ring_np = ring.cpu().numpy()
  raw = eu.raw(env)
  raw._ensure_allocated()
  assert raw._call_desc.flags == 1
  raw._call_desc.action_ring = R
  try:
    for t in range(2 * 30 + 5):
      b = raw._call(ring.data_ptr(), False)                        # host call index t -> row t mod R
      _check(raw._wrap_output(b), orc.call(ring_np[t % R], t), f'ring t={t}')
  finally:
    raw._call_desc.action_ring = 0


def test_tag_flag_is_only_set_when_call_indices_are_consecutive():
  """a graph captured with the HOST call count replays one index over and over, and a segment of a shared
  counter may be stepped twice between bumps — neither may take the single-launch step, whose readers would take an
  already-tagged word for an already-advanced one."""
  import bsuite_amd
  B = 3000
  shared = torch.zeros(1, dtype=torch.int64, device='cuda')
  seg = bsuite_amd.load_from_id('deep_sea/10', batch=B, seed=5, shared_step_counter=shared)
  ref = bsuite_amd.load_from_id('deep_sea/10', batch=B, seed=5)
  g = torch.Generator(device='cuda'); g.manual_seed(3)
  acts = torch.randint(2, (40, B), generator=g, device='cuda', dtype=torch.int32)
  for t in range(6):                                               # the counter is never bumped: every call repeats index 0
    x, y = seg.step(acts[t]), ref.step(acts[t])
    for u, v in zip(eu.to_np(x), eu.to_np(y)):
      np.testing.assert_array_equal(u, v, err_msg=f'shared counter t={t}')
  assert eu.raw(seg)._call_desc.flags == 0
  # host count + capture: the captured call is the two-launch step; eager calls before and after stay single-launch
  env = bsuite_amd.load_from_id('deep_sea/10', batch=B, seed=5, num_buffers=1)
  ref = bsuite_amd.load_from_id('deep_sea/10', batch=B, seed=5, num_buffers=1)
  static = acts[0].clone()
  env.step(static); ref.step(acts[0])
  assert eu.raw(env)._call_desc.flags == 1
  side = torch.cuda.Stream()
  side.wait_stream(torch.cuda.current_stream())
  graph = torch.cuda.CUDAGraph()
  with torch.cuda.stream(side):
    with torch.cuda.graph(graph, stream=side):
      out = env.step(static)
  torch.cuda.current_stream().wait_stream(side)
  for rep in range(1, 36):                                         # across an episode end
    static.copy_(acts[rep])
    graph.replay()
    want = ref.step(acts[rep])
    torch.cuda.synchronize()
    for u, v in zip(eu.to_np(out), eu.to_np(want)):
      np.testing.assert_array_equal(u, v, err_msg=f'replay {rep}')
  for t in range(36, 40):                                          # eager again: tagged, single launch
    x, y = env.step(acts[t]), ref.step(acts[t])
    assert eu.raw(env)._call_desc.flags == 1
    for u, v in zip(eu.to_np(x), eu.to_np(y)):
      np.testing.assert_array_equal(u, v, err_msg=f'eager after replays t={t}')
