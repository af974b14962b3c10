"""ORACLE shim for `skimage` (bsuite/utils/wrappers.py:26 imports skimage.transform; only the
out-of-scope ImageObservation wrapper calls it)."""
