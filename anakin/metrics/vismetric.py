from artiboost_amd.metrics import Vis2DMetric, VisHand2DMetric, VisMetric  # noqa: F401  (anakin/metrics/vismetric.py:18,71,361)
