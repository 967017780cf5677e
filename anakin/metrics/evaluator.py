from artiboost_amd.metrics import Evaluator  # noqa: F401  (anakin/metrics/evaluator.py:12)
