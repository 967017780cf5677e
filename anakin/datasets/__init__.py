from artiboost_amd.datasets import HO3D, HO3DV3, DexYCB, SynthOnly  # noqa: F401  (registers the DATASET types)
