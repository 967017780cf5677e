from artiboost_amd.synth import ArtiBoostLoader  # noqa: F401  (anakin/artiboost/artiboost_loader.py:33)
