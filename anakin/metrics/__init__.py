from artiboost_amd import metrics as _m  # noqa: F401  (registers the METRIC types)
