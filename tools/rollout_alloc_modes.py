"""Is a fused rollout's bimodal time per step (catch at 2^20 lanes 34 or 42 us, umbrella_length r16 17.6-24: profiles/r03, r06) a
property of WHERE its [T, B, ...] output arrays lie?  One process, one environment, the same actions; per round (a spacer
allocation of a different size in front each time, so that the driver hands out different memory) the outputs are
  sep    four separate torch allocations (what `rollout()` does),
  arena  ONE allocation, observation | reward | discount | step_type at 2 MiB-aligned offsets,
  arena' the same with the three columns in front of the observations,
and, in the last arena, the whole group moved by byte offsets.  Prints us per step per case.
  BSX_NATIVE_LIB=bsuite_amd/_lib/libbsuite_amd_tuning.so BSX_FUSED_ROLLOUT_MAX_MIB=256 python tools/rollout_alloc_modes.py catch/0 16
"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bsuite_amd  # noqa: E402
from bsuite_amd import _native  # noqa: E402

MIB2 = 2 << 20


def main():
  bsuite_id, T = sys.argv[1], int(sys.argv[2])
  B = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 20
  rounds = int(os.environ.get('ROUNDS', '5'))
  dev = torch.device('cuda:0')
  env = bsuite_amd.load_from_id(bsuite_id, batch=B, seed=5, device=dev)
  n_act = int(env.action_spec().num_values)
  g = torch.Generator(device=dev); g.manual_seed(1)
  actions = torch.randint(0, n_act, (T, B), dtype=torch.int32, device=dev, generator=g)
  env.reset()
  obs_shape = tuple(env.observation_spec().shape)
  numel = int(np.prod(obs_shape))
  n_obs, n_col = T * B * numel * 4, T * B * 4
  up = lambda n: (n + MIB2 - 1) // MIB2 * MIB2  # noqa: E731

  def timed(reps=5):
    for _ in range(2):
      env.rollout(actions)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
      env.rollout(actions)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps / T * 1e3

  def install(o):
    env.__dict__['_rollout_out'] = {T: (o, _native.TimeStepPtrs(o['reward'].data_ptr(), o['discount'].data_ptr(),
                                                                 o['step_type'].data_ptr(), o['observation'].data_ptr()))}

  def drop():
    env.__dict__['_rollout_out'] = {}
    torch.cuda.empty_cache()

  def separate():
    return dict(reward=torch.empty((T, B), dtype=torch.float32, device=dev), discount=torch.empty((T, B), dtype=torch.float32, device=dev),
                step_type=torch.empty((T, B), dtype=torch.int8, device=dev),
                observation=torch.empty((T, B) + obs_shape, dtype=torch.float32, device=dev))

  def arena(obs_first=True, off=0, buf=None):
    if buf is None:
      buf = torch.empty(up(n_obs) + 3 * up(n_col) + (64 << 20), dtype=torch.uint8, device=dev)
    p = off
    def carve(nbytes, dtype, shape):
      nonlocal p
      t = buf[p:p + nbytes].view(dtype).view(shape)
      p += up(nbytes)
      return t
    o = {}
    if obs_first:
      o['observation'] = carve(n_obs, torch.float32, (T, B) + obs_shape)
    o['reward'], o['discount'] = carve(n_col, torch.float32, (T, B)), carve(n_col, torch.float32, (T, B))
    o['step_type'] = carve(T * B, torch.int8, (T, B))
    if not obs_first:
      o['observation'] = carve(n_obs, torch.float32, (T, B) + obs_shape)
    o['_buf'] = buf
    return o

  print(f'{bsuite_id} T={T} B={B} obs {obs_shape}: slice stride {B * numel * 4} bytes')
  res = {'sep': [], 'arena': [], "arena'": []}
  spacers = []
  for k in range(rounds):
    for name, make in (('sep', separate), ('arena', lambda: arena(True)), ("arena'", lambda: arena(False))):
      drop()
      spacers.append(torch.empty((len(spacers) * 37 + 5) << 20, dtype=torch.uint8, device=dev))     # shifts the next allocation
      o = make(); install(o)
      res[name].append(timed())
      del o
  for name, v in res.items():
    print(f'{name:7s} ' + ' '.join(f'{x:7.2f}' for x in v) + f'   median {np.median(v):.2f}  min {min(v):.2f}  max {max(v):.2f}')
  drop()
  buf = torch.empty(up(n_obs) + 3 * up(n_col) + (64 << 20), dtype=torch.uint8, device=dev)
  line = []
  for off in (0, 4096, 1 << 20, 17 << 20, 0):
    install(arena(True, off, buf)); line.append(timed())
  print('one arena, group moved by 0 / 4 KiB / 1 MiB / 17 MiB / 0: ' + ' '.join(f'{x:.2f}' for x in line))


if __name__ == '__main__':
  main()
