"""Closed loop on the device: a batched environment stepped by a (random-weight) linear policy that
reads the observation tensor the engine just wrote — no host round trip anywhere in the loop.

    python examples/closed_loop_policy.py [bsuite_id] [lanes] [steps]

This is the batched counterpart of the reference run loop (bsuite/baselines/experiment.py:43-57):
`timestep = env.step(agent.select_action(timestep))`, with 2^20 environments per call.
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout
import bsuite_amd  # noqa: E402


def main():
  bsuite_id = sys.argv[1] if len(sys.argv) > 1 else 'deep_sea/10'
  lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
  steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
  env = bsuite_amd.load_from_id(bsuite_id, batch=lanes, seed=0)
  n_obs = int(torch.tensor(env.observation_spec().shape).prod())
  n_act = env.action_spec().num_values
  g = torch.Generator(device='cuda').manual_seed(0)
  weights = torch.randn((n_obs, n_act), device='cuda', generator=g)

  def policy(timestep):                           # greedy over a linear read-out of the observation
    logits = timestep.observation.reshape(lanes, n_obs) @ weights
    return logits.argmax(dim=1).to(torch.int32)

  ts = env.reset()
  for _ in range(20):
    ts = env.step(policy(ts))
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(steps):
    ts = env.step(policy(ts))
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  info = {k: float(v.sum()) for k, v in env.bsuite_info().items()}
  print(json.dumps(dict(bsuite_id=bsuite_id, lanes=lanes, steps=steps, ms_per_step=round(dt / steps * 1e3, 4),
                        env_steps_per_s=round(lanes * steps / dt), episodes_finished=int(env.episode_counters()[0]),
                        bsuite_info_sums=info)))


if __name__ == '__main__':
  main()
