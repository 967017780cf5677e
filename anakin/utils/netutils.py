from artiboost_amd.netutils import build_optimizer, build_scheduler  # noqa: F401  (anakin/utils/netutils.py:26-63)
