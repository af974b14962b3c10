"""GPU: engine-level behaviours around the kernels — output ring buffers, state_dict round trip,
device-resident call counter and HIP-graph replay, episode counters."""
import numpy as np
import pytest
import torch

from bsuite_amd.environments import catch, deep_sea, umbrella_chain
from tests import engine_util as eu

pytestmark = pytest.mark.gpu


def _acts(n, b, seed=0, na=3):
  g = torch.Generator(device='cuda')
  g.manual_seed(seed)
  return torch.randint(na, (n, b), generator=g, device='cuda', dtype=torch.int32)


def test_output_ring_keeps_previous_timestep_valid():
  env = catch.Catch(seed=1, batch=512, num_buffers=2)
  a = _acts(3, 512)
  ts0 = env.step(a[0])
  snap = ts0.observation.clone()
  ts1 = env.step(a[1])
  assert ts1.observation.data_ptr() != ts0.observation.data_ptr()
  torch.testing.assert_close(ts0.observation, snap, rtol=0, atol=0)      # still intact
  ts2 = env.step(a[2])
  assert ts2.observation.data_ptr() == ts0.observation.data_ptr()        # depth-2 ring wraps


def test_state_dict_round_trip_resumes_bit_exactly():
  a = _acts(40, 777, na=2)
  env = umbrella_chain.UmbrellaChain(chain_length=7, n_distractor=20, seed=3, batch=777)
  for t in range(13):
    env.step(a[t])
  saved = env.state_dict()
  want = [eu.to_np(env.step(a[t])) for t in range(13, 40)]
  info_want = {k: v.clone() for k, v in env.bsuite_info().items()}
  env2 = umbrella_chain.UmbrellaChain(chain_length=7, n_distractor=20, seed=999, batch=777)
  env2.load_state_dict(saved)
  for t, w in zip(range(13, 40), want):
    got = eu.to_np(env2.step(a[t]))
    for x, y in zip(got, w):
      np.testing.assert_array_equal(x, y)
  for k, v in env2.bsuite_info().items():
    torch.testing.assert_close(v, info_want[k], rtol=0, atol=0)


@pytest.mark.parametrize('deferred', [False, True])
def test_hip_graph_replay_equals_eager(deferred):
  """deferred=True: the captured steps take their call index as counter + position and the device
  call counter is bumped once per replay (Environment.step_counter_deferred)."""
  import contextlib
  B, T, reps, seed = 4096, 8, 5, 11
  a = _acts(T, B, na=2)
  eager = deep_sea.DeepSea(12, deterministic=False, mapping_seed=42, seed=seed, batch=B, num_buffers=1)
  graphed = deep_sea.DeepSea(12, deterministic=False, mapping_seed=42, seed=seed, batch=B, num_buffers=1,
                             device_step_counter=True)
  graphed.step(a[0])                              # allocate + call 0 outside capture
  eager.step(a[0])
  side = torch.cuda.Stream()
  side.wait_stream(torch.cuda.current_stream())
  g = torch.cuda.CUDAGraph()
  with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side):
      with (graphed.step_counter_deferred() if deferred else contextlib.nullcontext()):
        for t in range(T):
          last = graphed.step(a[t])
  torch.cuda.current_stream().wait_stream(side)
  for _ in range(reps):
    g.replay()
    for t in range(T):
      ref = eager.step(a[t])
  torch.cuda.synchronize()
  for x, y in zip(eu.to_np(last), eu.to_np(ref)):
    np.testing.assert_array_equal(x, y)
  assert graphed.device_step_index() == 1 + T * reps == eager.step_index
  for k, v in graphed.bsuite_info().items():
    torch.testing.assert_close(v, eager.bsuite_info()[k], rtol=0, atol=0)
  torch.testing.assert_close(graphed.episode_counters(), eager.episode_counters(), rtol=0, atol=0)


def test_episode_counters_count_ballots_exactly():
  B, N = 10007, 9
  env = deep_sea.DeepSea(N, mapping_seed=1, seed=0, batch=B)
  a = _acts(2 * (N + 1), B, na=2)
  n_last = n_first = 0
  for t in range(2 * (N + 1)):
    ts = env.step(a[t])
    n_last += int((ts.step_type == 2).sum().item())
    n_first += int((ts.step_type == 0).sum().item())
  c = env.episode_counters().cpu().numpy()
  assert (c[0], c[1]) == (n_last, n_first) == (2 * B, 2 * B)


@pytest.mark.gpu
def test_invalid_actions_are_clamped_and_counted():
  """SURVEY §8b error word: where the reference raises IndexError the batched kernels clamp/clip and
  count the lane-step; in-spec runs leave the counter at zero."""
  from bsuite_amd.environments import bandit, catch, discounting_chain
  B = 1000
  bad = torch.zeros(B, dtype=torch.bool, device='cuda')
  bad[::7] = True
  n_bad = int(bad.sum())
  # bandit: every non-reset call indexes the reward table
  env = bandit.SimpleBandit(mapping_seed=1, batch=B, seed=0)
  env.reset()
  a = torch.where(bad, torch.tensor(11, device='cuda'), torch.tensor(3, device='cuda')).to(torch.int32)
  ts = env.step(a)
  assert int(env.invalid_action_count()) == n_bad
  top = max(env._cfg.rewards[i] for i in range(11))
  assert torch.equal(ts.reward[bad], torch.full((n_bad,), env._cfg.rewards[10], device='cuda', dtype=torch.float32))
  del top
  env.step(a)                                                  # auto-reset call: actions ignored, nothing counted
  assert int(env.invalid_action_count()) == n_bad
  env.step(torch.full((B,), -5, dtype=torch.int32, device='cuda'))
  assert int(env.invalid_action_count()) == n_bad + B
  # catch: the paddle moves by (action - 1) and is clipped
  env = catch.Catch(batch=B, seed=0)
  env.reset()
  env.step(torch.where(bad, torch.tensor(3, device='cuda'), torch.tensor(1, device='cuda')).to(torch.int32))
  assert int(env.invalid_action_count()) == n_bad
  env.step(torch.full((B,), 2, dtype=torch.int32, device='cuda'))
  assert int(env.invalid_action_count()) == n_bad
  # discounting_chain: only the first action of an episode selects the chain
  env = discounting_chain.DiscountingChain(mapping_seed=0, batch=B)
  env.reset()
  env.step(torch.where(bad, torch.tensor(9, device='cuda'), torch.tensor(2, device='cuda')).to(torch.int32))
  env.step(torch.full((B,), 77, dtype=torch.int32, device='cuda'))      # later actions are not read
  assert int(env.invalid_action_count()) == n_bad
  # an in-spec random rollout never touches the word
  env = catch.Catch(batch=B, seed=1)
  g = torch.Generator(device='cuda').manual_seed(0)
  for _ in range(30):
    env.step(torch.randint(3, (B,), device='cuda', generator=g, dtype=torch.int32))
  assert int(env.invalid_action_count()) == 0
  assert int(env.episode_counters()[0]) == 3 * B               # three finished episodes per lane


@pytest.mark.parametrize('batch', [5, 6, 257, 4097])
@pytest.mark.parametrize('custom_table,big_dataset', [(False, False), (True, False), (False, True), (True, True)])
def test_mnist_observation_stream_tables_and_ragged_ends(batch, custom_table, big_dataset):
  """The mnist observation stream (csrc/mnist_fam.h): the reference's pixel table np.float32(int8) / 255 (mnist.py:64) is
  recognised and computed per wave; any other table in bsx_mnist_t.pixel_lut is read from the arguments.  Both forms, on
  batches whose last workgroup is partial (16 KiB = 5.22 rows), against the table applied on the host — on the small
  dataset and on one beyond 32 MiB (the real MNIST's class: six KiB-runs per wave, csrc/mnist.hip)."""
  from bsuite_amd.environments import mnist
  from tests import golden_util as gu
  images, labels = gu.mnist_dataset()
  if big_dataset:
    rng = np.random.default_rng(1)
    images = rng.integers(-128, 128, size=(43000, 28, 28), dtype=np.int8)
    labels = rng.integers(0, 10, size=43000).astype(np.uint8)
  env = mnist.MNISTBandit(seed=5, batch=batch, images=images, labels=labels, num_buffers=1)
  table = np.arange(256, dtype=np.uint8).view(np.int8).astype(np.float32) / 255
  if custom_table:
    table = (np.arange(256, dtype=np.float32) * 0.5 - 3.25).astype(np.float32)
    for b in range(256):
      env._cfg.pixel_lut[b] = float(table[b])
  flat = images.reshape(len(labels), -1).view(np.uint8)
  a = _acts(6, batch, na=10)
  for t in range(6):
    ts = env.step(a[t])
    obs = ts.observation.cpu().numpy().reshape(batch, -1)
    first = ts.step_type.cpu().numpy() == 0
    idx = (env._state['state'].cpu().numpy() & 0x00FFFFFF)
    want = np.where(first[:, None], table[flat[idx]], np.float32(0))
    np.testing.assert_array_equal(obs.view(np.uint32), want.astype(np.float32).view(np.uint32))
    assert first.all() if t % 2 == 0 else not first.any()
