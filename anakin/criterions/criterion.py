from artiboost_amd.criterions import Criterion  # noqa: F401  (anakin/criterions/criterion.py:30)
