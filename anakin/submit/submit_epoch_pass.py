from artiboost_amd.submit import SubmitEpochPass  # noqa: F401  (anakin/submit/submit_epoch_pass.py)
