"""ORACLE shim for `absl` (only absl.logging is touched, bsuite/utils/datasets.py:27)."""
