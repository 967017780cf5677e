from artiboost_amd.criterions import SymCornerLoss  # noqa: F401  (anakin/criterions/symcornerloss.py:18)
