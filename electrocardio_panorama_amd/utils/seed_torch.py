"""Seeding, as reference codes/utils/seed_torch.py:7-17 (python / numpy / torch generators)."""
import os
import random

import numpy as np
import torch


def seed_torch(seed=123):
    random.seed(seed)
    os.environ['PYTHONHASHSEED'] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
