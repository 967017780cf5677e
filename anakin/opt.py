"""anakin/opt.py: parses sys.argv at import, like the reference."""
from artiboost_amd.opt import parse

arg, cfg, custom_arg_string = parse()
