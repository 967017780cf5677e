from artiboost_amd.models import HybridBaseline  # noqa: F401  (anakin/models/hybridbaseline.py:18)
