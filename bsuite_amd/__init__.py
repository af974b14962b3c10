"""bsuite_amd — MI355X-native batched engine for the bsuite environment step()/reset() path.

Drop-in surface of `bsuite` for that path (bsuite/__init__.py:21-24): `load`, `load_from_id`,
`load_and_record*`, `sweep`; the dynamics run in hand-written HIP kernels (bsuite_amd/csrc) behind
the C ABI of include/bsuite_amd.h.
"""
from bsuite_amd import sweep
from bsuite_amd.bsuite import load
from bsuite_amd.bsuite import load_and_record
from bsuite_amd.bsuite import load_and_record_to_csv
from bsuite_amd.bsuite import load_and_record_to_terminal
from bsuite_amd.bsuite import load_from_id

__version__ = '0.1.0'
