from artiboost_amd.metrics import ValMetricAR2, ValMetricMean3DEPE2  # noqa: F401  (anakin/metrics/val_metric.py:55,146)
