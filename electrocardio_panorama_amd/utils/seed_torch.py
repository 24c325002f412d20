"""One seed for every generator the train step consumes: Python `random` (the two Standin lead choices per forward),
numpy (synthetic data), torch CPU + HIP (initialisation, dropout counter base).  Same effect as the reference's
codes/utils/seed_torch.py:7-17, minus its cuDNN flags (there is no cuDNN on this path)."""
import os
import random

import numpy as np
import torch


def seed_torch(seed=123):
    os.environ['PYTHONHASHSEED'] = str(seed)
    for seeder in (random.seed, np.random.seed, torch.manual_seed):
        seeder(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
