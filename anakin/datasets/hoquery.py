from artiboost_amd.registry import Queries, SynthQueries  # noqa: F401  (anakin/datasets/hoquery.py:6-56)
