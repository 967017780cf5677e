from artiboost_amd.metrics import AverageMeter, Metric  # noqa: F401  (anakin/metrics/metric.py:7,55)
