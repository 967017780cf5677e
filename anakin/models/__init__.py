from artiboost_amd.hpregnet import HOPRegNet, ManoBranch, ResNet18, ResNet34  # noqa: F401
from artiboost_amd.models import Arch, HybridBaseline  # noqa: F401  (registers the MODEL types; builder.py:82 imports them from here)
