from artiboost_amd.hpregnet import ManoBranch  # noqa: F401  (anakin/models/mano.py:17)
