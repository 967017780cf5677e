from artiboost_amd.registry import Registry, build_from_cfg  # noqa: F401  (anakin/utils/registry.py)
