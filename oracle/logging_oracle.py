"""ORACLE (test infrastructure): numpy restatement of the reference `Logging` wrapper's bookkeeping
(bsuite/utils/wrappers.py:85-125 `_track` / `_log_bsuite_data`, :140-147 `_logarithmic_logging`),
vectorised over lanes, to sit on top of oracle/coracle.OracleEnv outputs.  Pinned by the
`logging_*` fixtures under tests/golden/, which hold the rows the unmodified reference wrapper
wrote."""
import numpy as np

RATIOS = (1., 1.2, 1.4, 1.7, 2., 2.5, 3., 4., 5., 6., 7., 8., 9., 10.)


def logarithmic_logging(count):
  """Vectorised `_logarithmic_logging` (wrappers.py:140-147)."""
  count = np.asarray(count)
  exponent = np.floor(np.log10(np.maximum(1, count)))
  hit = np.zeros(count.shape, bool)
  for r in RATIOS:
    hit |= (count == 10 ** exponent * r)
  return hit


class TrackOracle:
  """Per-lane steps / episode / returns and the rows `logger.write` would receive."""

  def __init__(self, n_lanes, info_keys, log_by_step=False, log_every=False):
    self.steps = np.zeros(n_lanes, np.int64)
    self.episode = np.zeros(n_lanes, np.int64)
    self.total_return = np.zeros(n_lanes, np.float64)
    self.episode_len = np.zeros(n_lanes, np.int64)
    self.episode_return = np.zeros(n_lanes, np.float64)
    self.info_keys = list(info_keys)
    self.by_step, self.every = log_by_step, log_every
    self.rows = [[] for _ in range(n_lanes)]

  def track(self, step_type, reward_f64, info):
    """step_type int8 [B]; reward_f64 [B] with NaN where the reference returns None; info dict."""
    first, last = step_type == 0, step_type == 2
    r = np.where(np.isnan(reward_f64), 0.0, reward_f64)          # `timestep.reward or 0.0`
    self.steps += ~first
    self.episode_len += ~first
    self.episode += last
    self.episode_return = self.episode_return + r
    self.total_return = self.total_return + r
    if self.by_step:
      log = logarithmic_logging(self.steps) | self.every
    else:
      log = last & (logarithmic_logging(self.episode) | self.every)
    for i in np.nonzero(log)[0]:
      self.rows[i].append([float(self.steps[i]), float(self.episode[i]), self.total_return[i],
                           float(self.episode_len[i]), self.episode_return[i]] +
                          [float(info[k][i]) for k in self.info_keys])
    self.episode_len[last] = 0
    self.episode_return[last] = 0.0
