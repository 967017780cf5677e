from artiboost_amd.criterions import ManoLoss  # noqa: F401  (anakin/criterions/honetloss.py:12)
