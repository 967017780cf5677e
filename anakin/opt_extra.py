from anakin.opt import custom_arg_string
from artiboost_amd import opt as _opt


def data_generation_manager_parse():
    return _opt.data_generation_manager_parse(custom_arg_string)
