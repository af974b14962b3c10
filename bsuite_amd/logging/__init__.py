"""Loggers for the batched `Logging` wrapper (counterpart of bsuite/logging/)."""
