from artiboost_amd.recorder import RandomState  # noqa: F401  (anakin/utils/misc.py; pickled in random_state.pkl)
from artiboost_amd.registry import CONST, TrainMode, camel_to_snake, enable_lower_param, update_config  # noqa: F401
