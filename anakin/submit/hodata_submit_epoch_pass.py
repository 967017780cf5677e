from artiboost_amd.submit import HOSubmitEpochPass  # noqa: F401  (anakin/submit/hodata_submit_epoch_pass.py:21)
