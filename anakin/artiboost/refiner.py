from artiboost_amd.refiner import HORefiner, Refiner  # noqa: F401  (anakin/artiboost/refiner.py:21,227)
