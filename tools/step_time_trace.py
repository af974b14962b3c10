"""HIP-event time of EVERY eager step of the headline workload after the bench's own prelude (allocation, staggered phases,
W warm-up steps, synchronize): does a 20-step region measure the steady state?  Prints the steps' durations in groups of ten.
  python tools/step_time_trace.py [steps=120] [warmup=5]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import bsuite_amd  # noqa: E402


def main():
  steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
  warmup = int(sys.argv[2]) if len(sys.argv) > 2 else 5
  dev = torch.device('cuda:0')
  B = 1 << 20
  env = bsuite_amd.load_from_id('deep_sea/10', batch=B, device=dev, seed=42, num_buffers=2)
  actions = bench.synthetic_actions(torch, env.action_spec().num_values, 32, 0, B, dev)
  bench.stagger_phases(env, actions, 31)
  for t in range(warmup):
    env.step(actions[t % 32])
  torch.cuda.synchronize(dev)
  ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
  ev[0].record()
  for t in range(steps):
    env.step(actions[t % 32])
    ev[t + 1].record()
  torch.cuda.synchronize(dev)
  ms = [ev[t].elapsed_time(ev[t + 1]) for t in range(steps)]
  for g in range(0, steps, 10):
    print(f'steps {g:4d}-{g + 9:4d}: ' + ' '.join(f'{x * 1e3:6.1f}' for x in ms[g:g + 10]) + f'   mean {sum(ms[g:g + 10]) / len(ms[g:g + 10]) * 1e3:.1f} us')


if __name__ == '__main__':
  main()
