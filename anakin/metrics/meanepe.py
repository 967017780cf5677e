from artiboost_amd.metrics import Mean2DEPE, Mean3DEPE  # noqa: F401  (anakin/metrics/meanepe.py:103,108)
