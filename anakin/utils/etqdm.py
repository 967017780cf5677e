"""anakin/utils/etqdm.py: tqdm on rank 0, a plain iterable elsewhere."""
from types import MethodType


def etqdm(iterable, rank=None, **kwargs):
    if rank:
        iterable.set_description = MethodType(lambda self, _: None, iterable)
        return iterable
    from tqdm import tqdm
    return tqdm(iterable, bar_format="{l_bar}{bar:3}{r_bar}", **kwargs)
