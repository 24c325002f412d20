"""Packaged equivalent of reference codes/train_net.py:10-32: seed, data loaders, Solver(cfg).train(...).

Data: when the configured label lists exist (`cfg.DATA.train_label_path` / `test_label_path`, reference on-disk
format) the packaged Tianchi per-beat dataset (`dataset/tianchi.py`) feeds torch DataLoaders exactly as the reference
does; otherwise seeded synthetic `meta` batches with the same schema are generated."""
import os

import numpy as np
import torch

from . import parallel, synth
from .solver import Solver
from .utils import seed_torch


class SyntheticLoader:
    """Iterable of `meta` dicts (reference codes/dataset/tianchi.py:212-224 schema), `n_batches` per epoch."""

    def __init__(self, cfg, batch_size=32, n_batches=8, length=512, seed=0, Q=4):
        self.V, self.B, self.n, self.L, self.seed, self.Q = cfg.DATA.lead_num, batch_size, n_batches, length, seed, Q

    def __len__(self):
        return self.n

    def __iter__(self):
        for i in range(self.n):
            meta = synth.make_batch(self.B, self.V, self.L, seed=self.seed + i, Q=self.Q)
            yield {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in meta.items()}


def build_loaders(cfg, batch_size=32):
    if os.path.exists(cfg.DATA.train_label_path) and os.path.exists(cfg.DATA.test_label_path):
        from torch.utils.data import DataLoader
        from .dataset import build_dataset
        train = DataLoader(build_dataset(cfg, phase='train'), batch_size=batch_size, shuffle=True, num_workers=16,
                           drop_last=True, collate_fn=_collate)
        test = DataLoader(build_dataset(cfg, phase='test'), batch_size=batch_size, num_workers=8, drop_last=True,
                          collate_fn=_collate)
        return train, test
    return SyntheticLoader(cfg, batch_size, seed=cfg.seed), SyntheticLoader(cfg, batch_size, 2, seed=cfg.seed + 10 ** 6)


def _collate(items):
    """Stack the array fields of the `meta` dicts (ids and lead-name lists stay lists)."""
    out = {}
    for k in items[0]:
        v = [it[k] for it in items]
        out[k] = torch.from_numpy(np.stack(v)) if isinstance(v[0], np.ndarray) else v
    return out


def main(cfg):
    parallel.init_from_env()
    seed_torch(seed=cfg.seed)
    os.makedirs(os.path.join(cfg.output_dir, cfg.desc), exist_ok=True)
    train_dl, test_dl = build_loaders(cfg)
    solver = Solver(cfg)
    solver.train(train_dl, test_dl)
