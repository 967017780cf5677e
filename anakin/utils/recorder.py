from artiboost_amd.recorder import Recorder  # noqa: F401  (anakin/utils/recorder.py:28)
