"""CPU, world_size 2, gloo: the multi-process path — lane sharding by rank and the end-of-rollout
all-gather of per-rank summaries (the one collective; RCCL on the GPU box, gloo here)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bsuite_amd import distributed as bdist


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


class _Shard:
  """The part of the batched Environment surface local_summary() reads (environments/base.py: batch_size,
  episode_counters(), bsuite_info()), on CPU tensors: lane i of the GLOBAL batch has finished an episode iff
  i % 3 == 0 and carries total_regret i."""

  def __init__(self, off, n):
    self.batch_size = n
    self._lanes = torch.arange(off, off + n, dtype=torch.float64)

  def episode_counters(self):
    return torch.tensor([int((self._lanes % 3 == 0).sum()), self.batch_size], dtype=torch.int64)

  def bsuite_info(self):
    return {'total_regret': self._lanes.clone()}


class _Wrapped:
  """... and behind a wrapper (utils/wrappers.py: raw_env + delegation), as bench.py hands it over."""

  def __init__(self, raw):
    self.raw_env = raw

  def bsuite_info(self):
    return self.raw_env.bsuite_info()


def _worker(rank, world, port, total, q):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  off, n = bdist.shard_lanes(total, rank, world)
  env = _Shard(off, n)
  vec, names = bdist.local_summary(_Wrapped(env) if rank else env)   # the function the GPU path calls (bench.py, sweep_batch.py)
  assert names == ('lanes', 'episodes_finished', 'episodes_started', 'total_regret')
  g = bdist.all_gather_summary(vec)
  red = bdist.reduce_summary(g, names)
  q.put((rank, off, n, g.tolist(), red))
  dist.barrier()
  dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_sharding_and_summary_allgather():
  world, total = 2, 1001
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
  for p in procs:
    p.start()
  res = sorted(q.get(timeout=100) for _ in range(world))
  for p in procs:
    p.join(30)
    assert p.exitcode == 0
  (r0, o0, n0, g0, red0), (r1, o1, n1, g1, red1) = res
  assert (o0, n0, o1, n1) == (0, 501, 501, 500)
  assert g0 == g1 and len(g0) == 2                       # every rank holds every rank's block
  assert red0 == red1
  assert red0['lanes'] == total
  assert red0['total_regret'] == sum(range(total))
  assert red0['episodes_finished'] == len([x for x in range(total) if x % 3 == 0])


def test_identity_without_process_group():
  v = torch.tensor([1.0, 2.0], dtype=torch.float64)
  assert bdist.all_gather_summary(v).tolist() == [[1.0, 2.0]]
