from artiboost_amd.submit import HOSubmitEpochPass, SubmitEpochPass  # noqa: F401  (anakin/submit/__init__.py)
