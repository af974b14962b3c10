import logging as _l

info = _l.info
warning = _l.warning
PythonHandler = _l.StreamHandler
