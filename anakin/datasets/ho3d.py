from artiboost_amd.datasets import HO3D, HO3DV3  # noqa: F401  (anakin/datasets/ho3d.py)
