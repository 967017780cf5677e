from artiboost_amd.metrics import AR  # noqa: F401  (anakin/metrics/bopAR.py:16)
