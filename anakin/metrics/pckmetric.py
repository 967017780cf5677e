from artiboost_amd.metrics import Hand2DPCKMetric, Hand3DPCKMetric, Obj2DPCKMetric, Obj3DPCKMetric, PCKMetric  # noqa: F401  (anakin/metrics/pckmetric.py)
