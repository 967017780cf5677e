from artiboost_amd.criterions import HandOrdLoss, SceneOrdLoss  # noqa: F401  (anakin/criterions/ordinal.py:126,222)
