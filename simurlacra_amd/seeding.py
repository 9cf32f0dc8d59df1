"""pyrado.set_seed (P/__init__.py:135-183): MD5("{base}-{sub}-{subsub}") crushed to 32 bit -> random, numpy, torch."""
import hashlib
import random
from typing import Optional

import numpy as np

_BASE_SEED = None


def derive_seed(base_seed, sub_seed=None, sub_sub_seed=None) -> Optional[int]:
    if sub_seed is None:
        sub_seed = 0
    if sub_sub_seed is None:
        sub_sub_seed = 0
    if not isinstance(base_seed, int):
        return None
    return int(hashlib.md5(f"{base_seed}-{sub_seed}-{sub_sub_seed}".encode()).hexdigest(), 16) % (2 ** 32)


def set_seed(seed: Optional[int], sub_seed: int = None, sub_sub_seed: int = None, verbose: bool = False):
    global _BASE_SEED
    s = derive_seed(seed, sub_seed, sub_sub_seed)
    if s is None:
        if verbose:
            print(f"Base seed {seed} is not an integer -- the random number generators' seeds were not set.")
        return None
    import torch

    random.seed(s)
    np.random.seed(s)
    torch.manual_seed(s)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(s)
    _BASE_SEED = seed
    if verbose:
        print(f"Set the random number generators' seed to {s} (base seed {seed}, sub-seed {sub_seed}, "
              f"sub-sub-seed {sub_sub_seed}).")
    return s


def get_base_seed():
    return _BASE_SEED
