from artiboost_amd.hpregnet import HOPRegNet  # noqa: F401  (anakin/models/hpregnet.py:19)
