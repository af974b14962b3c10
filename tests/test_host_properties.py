"""CPU: property tests (hypothesis) of the host-side partitioning and table logic."""
import numpy as np
import torch  # noqa: F401
from hypothesis import given, settings, strategies as st

from bsuite_amd import distributed as bdist
from bsuite_amd import sweep_batch as sb
from bsuite_amd.utils import wrappers


@settings(max_examples=200, deadline=None)
@given(st.integers(0, 1 << 40), st.integers(1, 64))
def test_shard_lanes_is_a_contiguous_partition(total, world):
  shards = [bdist.shard_lanes(total, r, world) for r in range(world)]
  assert shards[0][0] == 0
  for (o0, n0), (o1, _) in zip(shards, shards[1:]):
    assert o0 + n0 == o1                                          # contiguous, in rank order
  assert shards[-1][0] + shards[-1][1] == total
  sizes = [n for _, n in shards]
  assert max(sizes) - min(sizes) <= 1


@settings(max_examples=200, deadline=None)
@given(st.integers(1, 500), st.integers(0, 1 << 30))
def test_segment_table_covers_every_lane_once(n_ids, extra):
  ids = [f'x/{i}' for i in range(n_ids)]
  total = n_ids + extra
  table = sb.segment_table(ids, total)
  assert [t[0] for t in table] == ids
  begin = 0
  for _, b, lanes in table:
    assert b == begin and lanes >= 1
    begin += lanes
  assert begin == total
  assert len({lanes for _, _, lanes in table[:-1]}) <= 1          # even split, remainder to the last id


@settings(max_examples=200, deadline=None)
@given(st.lists(st.floats(0.0, 1e9, allow_nan=False), min_size=1, max_size=200), st.integers(1, 16))
def test_assign_segments_balances_within_one_largest_item(costs, world):
  rank_of = sb.assign_segments(costs, world)
  assert len(rank_of) == len(costs) and all(0 <= r < world for r in rank_of)
  load = np.zeros(world)
  for c, r in zip(costs, rank_of):
    load[r] += c
  # greedy LPT: no rank exceeds the lightest one by more than the largest single item
  assert load.max() - load.min() <= max(costs) + 1e-6 * max(1.0, load.max())


@settings(max_examples=50, deadline=None)
@given(st.integers(0, 2_000_000))
def test_log_point_table_is_exactly_the_reference_predicate(max_count):
  pts = wrappers.logarithmic_logging_points(max_count)
  assert pts == sorted(set(pts)) and all(0 <= p <= max_count for p in pts)
  assert all(wrappers._logarithmic_logging(p) for p in pts)
  # no point is missed: probe the neighbourhood of every tabulated point and a random sample
  probe = set()
  for p in pts:
    probe.update((p - 1, p + 1))
  rng = np.random.RandomState(max_count % 1000)
  probe.update(int(x) for x in rng.randint(0, max_count + 1, size=200))
  listed = set(pts)
  for q in probe:
    if 0 <= q <= max_count and q not in listed:
      assert not wrappers._logarithmic_logging(q), q


@settings(max_examples=200, deadline=None)
@given(st.integers(1, 4), st.integers(1, 64), st.integers(1, 64), st.integers(1, 6))
def test_image_rule_selection(size, h, w, tail):
  cfg = wrappers._image_cfg((h, w, tail), (1, size))
  assert (cfg.mode, cfg.in_rows * cfg.in_cols, cfg.out_rows, cfg.out_cols, cfg.tail) == (0, size, h, w, tail)
