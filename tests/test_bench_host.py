"""CPU: the host-side pieces of bench.py — byte accounting, the shardable synthetic action
stream, the reference CPU baseline record, and the self-launch refusal without devices."""
import json
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_algorithmic_bytes_match_the_survey():
  # deep_sea N=30 3621 B, catch 10x5 221 B, cartpole 85 B, mountain_car 49 B
  for w, want in (('deep_sea', 3621), ('catch', 221), ('cartpole', 85), ('mountain_car', 49)):
    _, _, _, numel, state_bytes, _ = bench.WORKLOADS[w]
    assert bench.algorithmic_bytes_per_step(numel, state_bytes) == want


def test_synthetic_actions_are_a_function_of_global_lane_and_step():
  dev = torch.device('cpu')
  full = bench.synthetic_actions(torch, 3, 32, 0, 4096, dev)
  assert full.dtype == torch.int32 and full.shape == (32, 4096)
  assert int(full.min()) == 0 and int(full.max()) == 2
  # any sharding sees the same per-lane sequences
  a = bench.synthetic_actions(torch, 3, 32, 0, 1000, dev)
  b = bench.synthetic_actions(torch, 3, 32, 1000, 3096, dev)
  assert torch.equal(torch.cat([a, b], dim=1), full)
  # and they are close to uniform, per step and per lane
  counts = torch.bincount(full.flatten().long(), minlength=3).double() / full.numel()
  assert float((counts - 1 / 3).abs().max()) < 0.01
  per_step = torch.stack([torch.bincount(full[t].long(), minlength=3) for t in range(32)]).double() / 4096
  assert float((per_step - 1 / 3).abs().max()) < 0.04
  big = bench.synthetic_actions(torch, 11, 4, (1 << 20) * 7, 8, dev)          # lane offsets of an 8-GPU run
  assert int(big.min()) >= 0 and int(big.max()) <= 10


def test_cpu_baseline_times_the_reference_live():
  """`cpu_baseline` is measured in the run, on the box: the unmodified reference (from /root/reference here, from the
  staged oracle/_ref byte-code on the GPU box) on one core and on one process per core, the C port beside it."""
  from oracle import replay
  if not replay.reference_available():
    import pytest
    pytest.skip('no reference on this box')
  rec = bench.cpu_baseline('catch/0', 'catch', {}, 3, single_s=1.0, all_s=1.0)
  assert rec['kind'] == 'reference' and rec['cores'] == 1 and rec['unit'] == 'env-steps/s'
  assert 2e4 < rec['value'] < 2e6                                      # the numpy reference: ~1e5 env-steps/s per core
  assert rec['all_cores']['cores'] == bench._host_cores() and rec['all_cores']['value'] > 0.5 * rec['value']
  assert rec['port']['kind'] == 'port' and rec['port']['value'] > rec['value']
  assert 'load_from_id' in rec['sample']


def test_the_line_is_rounded_and_the_staged_reference_imports():
  line = bench.sig({'a': 1.23456789e9, 'b': [0.000123456789, 3], 'c': {'d': float('nan'), 'e': 'text'}})
  assert line == {'a': 1.2346e9, 'b': [0.00012346, 3], 'c': {'d': None, 'e': 'text'}}
  from oracle import stage_reference
  if stage_reference.reference_present():
    assert stage_reference.stage()
    env = dict(os.environ, BSX_REFERENCE_STAGED='1')
    code = ('import sys; sys.path.insert(0, %r); from oracle import replay; bs = replay.import_reference(); '
            'assert replay.reference_origin() == "staged" and bs.__file__.endswith(".pyc"), bs.__file__; '
            'print(bs.load_from_id("deep_sea/10").reset().observation.shape)' % ROOT)
    out = subprocess.run([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE, text=True, timeout=120, check=True).stdout
    assert '(30, 30)' in out


def test_self_launch_refuses_when_devices_are_missing():
  if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
    return
  env = dict(os.environ)
  env.pop('WORLD_SIZE', None)
  env.pop('BSX_BENCH_SINGLE_DEVICE', None)
  p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], env=env, stdout=subprocess.PIPE,
                     stderr=subprocess.PIPE, text=True, timeout=300)
  assert p.returncode != 0 and 'HIP device' in (p.stderr + p.stdout)


def test_committed_pmc_passes_feed_the_roofline_fields():
  """`roofline.traffic` / the `valu` roofline of the bench line come from committed rocprofv3 PMC passes
  (tools/pmc.py → profiles/rNN/): every BASELINE config has one, each with a non-empty FETCH pass, and the HBM traffic
  stays within 15 % of the algorithmic bytes (more would mean wasted re-reads, less a broken pass)."""
  for w, alg in (('deep_sea', 3621), ('catch', 221), ('cartpole', 85), ('mountain_car', 49)):
    hbm, src = bench.pmc_traffic(w, 1 << 20)
    assert hbm is not None and src.startswith('profiles/r'), w
    assert 0.9 < hbm / (alg * (1 << 20)) < 1.15, (w, hbm)
  for w in ('sweep_closed', 'sweep_pipelined'):
    hbm, src = bench.pmc_traffic(w, 1 << 20)
    assert hbm is not None and 0.95 < hbm / 885.58e6 < 1.1, (w, hbm)
  assert bench.pmc_traffic('deep_sea', 1 << 19) == (None, None)          # measured at 2^20 lanes only
  for w in ('cartpole', 'mountain_car'):
    v = bench.pmc_valu(w, 'rollout16', 1 << 20)
    assert v is not None and v['insts'] > 1e7 and 0.2 < v['busy'] < 1.0, (w, v)
  # VALU issue peak: 256 CUs x 4 SIMDs, one wave64 instruction per 2 cycles at 2.4 GHz
  assert abs(bench.VALU_PEAK_GINSTR - 1228.8) < 1e-9
