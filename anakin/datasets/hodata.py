from artiboost_amd.datasets import ho_collate  # noqa: F401  (anakin/datasets/hodata.py:17)
from artiboost_amd.realdata import HOdataSource as HOdata  # noqa: F401
