from artiboost_amd.metrics import LossesMetric  # noqa: F401  (anakin/metrics/lossesmetric.py:12)
