from artiboost_amd.criterions import JointsLoss  # noqa: F401  (anakin/criterions/jointloss.py:14)
