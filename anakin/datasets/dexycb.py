from artiboost_amd.datasets import DexYCB  # noqa: F401  (anakin/datasets/dexycb.py)
