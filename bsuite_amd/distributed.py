"""Multi-GPU sharding of a batch of environments: one process per GPU, lanes split into contiguous
ranges, zero communication per step.

The reference's only parallelism is a process pool over bsuite_ids
(bsuite/baselines/utils/pool.py:28-54, `map_mpi`); environments never interact, so the batched
engine shards lanes embarrassingly.  Because a lane's random draws are keyed by its GLOBAL lane id
(include/bsx_stream.h), a run sharded G ways is bit-identical, lane for lane, to the unsharded run.

The one collective on the path is the end-of-rollout reduction of per-rank summaries (episodes
finished, bsuite_info sums): an all-gather of a few dozen bytes per rank over RCCL/xGMI
(`torch.distributed` backend "nccl" on ROCm), gloo on CPU in the tests.
"""
from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist


def shard_lanes(total_lanes: int, rank: int, world_size: int) -> Tuple[int, int]:
  """Contiguous lane range of `rank`: returns (lane_offset, n_lanes).  Remainder lanes go to the
  lowest ranks so shard sizes differ by at most one."""
  if not 0 <= rank < world_size:
    raise ValueError(f'rank {rank} outside world of {world_size}')
  base, extra = divmod(int(total_lanes), int(world_size))
  n = base + (1 if rank < extra else 0)
  offset = rank * base + min(rank, extra)
  return offset, n


def local_summary(env) -> Tuple[torch.Tensor, Tuple[str, ...]]:
  """f64 vector [lanes, episodes_finished, episodes_started, sum(info_k)...] for this shard."""
  raw = env.raw_env if hasattr(env, 'raw_env') else env
  counters = raw.episode_counters().to(torch.float64)
  info = env.bsuite_info()
  keys = tuple(sorted(info))
  parts = [torch.tensor([float(raw.batch_size)], dtype=torch.float64, device=counters.device), counters]
  if keys:
    parts.append(torch.stack([info[k].sum() for k in keys]))
  return torch.cat(parts), ('lanes', 'episodes_finished', 'episodes_started') + keys


def all_gather_summary(vec: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
  """All-gathers each rank's summary vector -> [world, k] on every rank.  With no process group initialised it is the
  identity.  ONE collective into ONE pre-allocated [world, k] tensor (`all_gather_into_tensor`): on device
  tensors under the "nccl" backend that is a single ncclAllGather over RCCL/xGMI, no per-rank allocations and no
  stack afterwards.  Under gloo (the CPU tests) a device tensor is gathered through host memory."""
  if not (dist.is_available() and dist.is_initialized()):
    return vec.unsqueeze(0)
  world = dist.get_world_size(group)
  src = vec.contiguous()
  if vec.is_cuda and dist.get_backend(group) != 'nccl':   # gloo (tests): gather through host memory
    src = src.cpu()
  # (the concatenated form of the output — every backend takes it; gloo refuses the stacked one)
  out = torch.empty(world * src.numel(), dtype=src.dtype, device=src.device)
  dist.all_gather_into_tensor(out, src.reshape(-1), group=group)
  return out.view((world,) + tuple(src.shape)).to(vec.device)


def reduce_summary(gathered: torch.Tensor, names: Tuple[str, ...]) -> Dict[str, float]:
  total = gathered.sum(dim=0).tolist()
  return dict(zip(names, total))
