from artiboost_amd.models import Arch  # noqa: F401  (anakin/models/arch.py:11)
