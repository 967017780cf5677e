from artiboost_amd.criterions import Criterion, HandOrdLoss, JointsLoss, SceneOrdLoss, SymCornerLoss  # noqa: F401  (registers the LOSS types)
