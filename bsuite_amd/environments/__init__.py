"""Batched bsuite environments (mirror of bsuite/environments/__init__.py: exposes the base class)."""
from bsuite_amd.environments.base import Environment
