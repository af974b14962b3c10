"""GPU: the TWO-launch step (lane advance + observation store stream) and the software-pipelined rollout of deep_sea /
catch at SMALL boards.  In the product library a board of at most 128 floats (catch 10x5, deep_sea N <= 11) takes the
fused one-launch step / rollout (bsx_fused_tile_kernel, bsx_fused_rollout_kernel), so the in-process parity tests at such shapes — ragged and 1-lane batches,
boards whose 16-byte chunks straddle two lanes, odd slice alignments, both parities of T — exercise the FUSED kernel;
the pair path runs in-process only for bigger boards (deep_sea N >= 12, incl. the benched N=30) and batches.  Here
the same small-shape parity tests run once more in a subprocess against the tuning build of the library (-DBSX_TUNING,
bsuite_amd/build.py) with BSX_FUSED_TILE_MAX_CELLS=0 and BSX_DEEP_SEA_STEP1=0, i.e. with the fused step and the single-launch
deep_sea step (N >= 28, up to 2^18 lanes) switched off: every edge case of the stream kernel stays covered against the
golden fixtures and the C oracle."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(1200)
def test_pair_path_parity_at_small_shapes():
  from bsuite_amd import build as _build
  env = dict(os.environ, BSX_NATIVE_LIB=_build.build(tuning=True), BSX_FUSED_TILE_MAX_CELLS='0', BSX_DEEP_SEA_STEP1='0', PYTHONPATH=ROOT)
  p = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', '-m', 'gpu', '-p', 'no:cacheprovider',
                      'tests/test_gpu_golden.py', 'tests/test_gpu_oracle_batch.py', 'tests/test_gpu_rollout.py',
                      'tests/test_gpu_delta_obs.py', 'tests/test_gpu_engine_features.py',
                      '-k', 'deep_sea or catch or pipelined or delta or engine'],
                     cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1100)
  tail = p.stdout[-3000:]
  assert p.returncode == 0, tail
  assert ' passed' in tail and 'failed' not in tail, tail


@pytest.mark.timeout(900)
def test_256_lane_fused_tiles_at_small_shapes():
  """Up to 2^18 lanes the fused one-launch step uses 64-lane tiles (bsx_fused_tile64_kernel), so the in-process tests at
  small shapes no longer reach the 256-lane tile kernel that 2^17 < B <= 2^19 lanes take: here they do
  (BSX_FUSED_TILE64_MAX_LANES=0 in the tuning build)."""
  from bsuite_amd import build as _build
  env = dict(os.environ, BSX_NATIVE_LIB=_build.build(tuning=True), BSX_FUSED_TILE64_MAX_LANES='0', PYTHONPATH=ROOT)
  p = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', '-m', 'gpu', '-p', 'no:cacheprovider',
                      'tests/test_gpu_golden.py', 'tests/test_gpu_oracle_batch.py', 'tests/test_gpu_engine_features.py',
                      '-k', 'deep_sea or catch or engine'],
                     cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=800)
  tail = p.stdout[-3000:]
  assert p.returncode == 0, tail
  assert ' passed' in tail and 'failed' not in tail, tail


@pytest.mark.timeout(900)
def test_two_lanes_per_thread_advance_at_small_shapes():
  """the two-lanes-per-thread lane advance (bsx_advance2_kernel) is size-gated at 4096 workgroups, so no
  in-process test at a small shape reaches it: here every deep_sea / catch / mnist parity test of the pair path runs
  with BSX_ADVANCE_LPT2_MIN_BLOCKS=1 (tuning build) — one-lane, ragged and several-thousand-lane batches, the `mine[h]`
  guards and the (blocks + 1) / 2 grid included (fused steps and the single-launch deep_sea step switched off)."""
  from bsuite_amd import build as _build
  env = dict(os.environ, BSX_NATIVE_LIB=_build.build(tuning=True), BSX_ADVANCE_LPT2_MIN_BLOCKS='1', BSX_FUSED_TILE_MAX_CELLS='0',
             BSX_DEEP_SEA_STEP1='0', PYTHONPATH=ROOT)
  p = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', '-m', 'gpu', '-p', 'no:cacheprovider',
                      'tests/test_gpu_golden.py', 'tests/test_gpu_oracle_batch.py', 'tests/test_gpu_engine_features.py',
                      '-k', 'deep_sea or catch or mnist or engine'],
                     cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=800)
  tail = p.stdout[-3000:]
  assert p.returncode == 0, tail
  assert ' passed' in tail and 'failed' not in tail, tail


@pytest.mark.timeout(900)
def test_wide_row_stream_chunk_counts():
  """The wide-row store stream with 1 and 4 chunks per thread (tuning build; the product library has K = 2)."""
  from bsuite_amd import build as _build
  for k in ('1', '4'):
    env = dict(os.environ, BSX_NATIVE_LIB=_build.build(tuning=True), BSX_ROW_STREAM_K=k, PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', '-m', 'gpu', '-p', 'no:cacheprovider',
                        'tests/test_gpu_wide_rows.py', '-k', 'bit_exact'],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=400)
    tail = p.stdout[-3000:]
    assert p.returncode == 0, tail
    assert ' passed' in tail and 'failed' not in tail, tail


@pytest.mark.timeout(900)
def test_two_lanes_per_thread_eager_step_at_small_shapes():
  """small_obs_eager2_kernel (the lean eager step of bandit / discounting_chain / memory_len / cartpole / mountain_car
  with two lanes per thread) is size-gated at 2048+ workgroups: every parity test of those families once more with
  BSX_EAGER2_MIN_BLOCKS=1 (tuning build) — one-lane and ragged batches, both variants of cartpole; and with four lanes
  per thread, the form that was measured and not adopted."""
  from bsuite_amd import build as _build
  for lpt in ('2', '4'):
    env = dict(os.environ, BSX_NATIVE_LIB=_build.build(tuning=True), BSX_EAGER2_MIN_BLOCKS='1', BSX_EAGER_LPT=lpt, PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', '-m', 'gpu', '-p', 'no:cacheprovider',
                        'tests/test_gpu_golden.py', 'tests/test_gpu_oracle_batch.py', 'tests/test_gpu_dm_env_conformance.py',
                        'tests/test_gpu_engine_features.py',
                        '-k', 'bandit or discounting or memory or cartpole or mountain_car or swingup'],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=400)
    tail = p.stdout[-3000:]
    assert p.returncode == 0, tail
    assert ' passed' in tail and 'failed' not in tail, tail


@pytest.mark.timeout(900)
def test_split_sweep_step_with_a_topped_up_first_launch():
  """bsx_group_step_split tops launch 1 up with small-observation workgroups when phase 0 is more than one dispatch round
  (BSX_SPLIT_ROUND_DEFAULT workgroups, csrc/sweep_mixed.hip) — in-process only the 2^20-lane tests get there
  (test_gpu_benched_sizes.py).  The sweep-group tests once more through the tuning build with the round at 3 and 20
  workgroups: launch 1 = the lane advance alone / + a few / + all of the small segments' workgroups, the call counter
  bumped from whichever launch retires the last of them."""
  from bsuite_amd import build as _build
  for r in ('3', '20'):
    env = dict(os.environ, BSX_NATIVE_LIB=_build.build(tuning=True), BSX_SPLIT_ROUND=r, PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', '-m', 'gpu', '-p', 'no:cacheprovider',
                        'tests/test_gpu_sweep_batch.py', '-k', 'split or sweep'],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=400)
    tail = p.stdout[-3000:]
    assert p.returncode == 0, tail
    assert ' passed' in tail and 'failed' not in tail, tail
