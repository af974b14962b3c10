"""CSV logging in bsuite's wire format (counterpart of bsuite/logging/csv_logging.py:29-89).

One file per bsuite_id named `bsuite_id_-_<name>-<index>.csv`, one row per log point with the
columns of `bsuite_amd.utils.wrappers.Logging` (STANDARD_KEYS + bsuite_info keys), so the untouched
reference loader `bsuite.logging.csv_load.load_bsuite(results_dir)` and its analysis read runs of
this engine exactly like its own.  Batched runs write one results_dir per lane
(`write_lane_csvs`), mirroring "one run = one results directory".
"""
import os
from typing import Any, Mapping

import pandas as pd

from bsuite_amd import sweep
from bsuite_amd.logging import base
from bsuite_amd.utils import wrappers

SAFE_SEPARATOR = '-'
INITIAL_SEPARATOR = '_-_'
BSUITE_PREFIX = 'bsuite_id' + INITIAL_SEPARATOR


def csv_path(bsuite_id: str, results_dir: str) -> str:
  safe_bsuite_id = bsuite_id.replace(sweep.SEPARATOR, SAFE_SEPARATOR)
  return os.path.join(results_dir, f'{BSUITE_PREFIX}{safe_bsuite_id}.csv')


def wrap_environment(env, bsuite_id: str, results_dir: str, overwrite: bool = False,
                     log_by_step: bool = False):
  """Returns a wrapped environment that logs using CSV (csv_logging.py:34-41)."""
  logger = Logger(bsuite_id, results_dir, overwrite)
  return wrappers.Logging(env, logger, log_by_step=log_by_step)


class Logger(base.Logger):
  """Row sink for `wrappers.Logging`: accumulates the rows of one bsuite_id and mirrors them to its
  CSV file after every write (the reference does the same — writes are log-spaced, hence rare)."""

  def __init__(self, bsuite_id: str, results_dir: str = '/tmp/bsuite', overwrite: bool = False):
    os.makedirs(results_dir, exist_ok=True)          # tolerant of concurrent creators
    target = csv_path(bsuite_id, results_dir)
    if not overwrite and os.path.exists(target):
      raise ValueError(f'File {target} already exists. Specify a different '
                       'directory, or set overwrite=True to overwrite existing data.')
    self._save_path = target
    self._rows = []

  @property
  def save_path(self) -> str:
    return self._save_path

  def write(self, data: Mapping[str, Any]):
    """Appends one row and rewrites the CSV (column order = first-seen key order, like pandas)."""
    self._rows.append(dict(data))
    pd.DataFrame(self._rows).to_csv(self._save_path, index=False)


def write_lane_csvs(logging_env: wrappers.Logging, bsuite_id: str, results_root: str, lanes,
                    overwrite: bool = False):
  """Batched runs: writes lane k's rows to `<results_root>/lane_<k>/bsuite_id_-_….csv`."""
  paths = []
  for lane in lanes:
    d = os.path.join(results_root, f'lane_{int(lane)}')
    os.makedirs(d, exist_ok=True)
    path = csv_path(bsuite_id, d)
    if os.path.exists(path) and not overwrite:
      raise ValueError(f'File {path} already exists.')
    logging_env.dataframe(int(lane)).to_csv(path, index=False)
    paths.append(path)
  return paths
