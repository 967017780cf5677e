from artiboost_amd.recorder import Summarizer  # noqa: F401  (anakin/utils/summarizer.py:12)
