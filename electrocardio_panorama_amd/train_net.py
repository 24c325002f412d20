"""Packaged equivalent of reference codes/train_net.py:10-32: seed, data loaders, Solver(cfg).train(...).

Data.  The configured label lists (`cfg.DATA.train_label_path` / `test_label_path`, reference on-disk format) feed the
packaged Tianchi per-beat dataset (`dataset/tianchi.py`) through torch DataLoaders exactly as the reference does; a
missing list is an error, as it is in the reference.  `cfg.DATA.synthetic = True` (a key the reference does not have)
selects seeded synthetic `meta` batches with the same schema instead.

Data parallelism (torchrun, one process per GPU).  `batch_size` is the GLOBAL batch, as under the reference's
nn.DataParallel (solver.py:32-34): every rank draws its 1/world share -- a DistributedSampler over the real dataset,
or the contiguous shard of each synthetic global batch (`parallel.ShardedLoader`).  All ranks keep the same seeds, so
parameters and Standin lead choices agree; the test set is evaluated by every rank after rank 0's BatchNorm buffers
have been broadcast, so all ranks see the same metrics."""
import os

import numpy as np
import torch

from . import parallel, synth
from .prefetch import DevicePrefetcher
from .solver import Solver
from .utils import seed_torch


class SyntheticLoader:
    """Iterable of `meta` dicts (reference codes/dataset/tianchi.py:212-224 schema), `n_batches` per epoch."""

    def __init__(self, cfg, batch_size=32, n_batches=8, length=512, seed=0, Q=None):
        self.V, self.B, self.n, self.L, self.seed = cfg.DATA.lead_num, batch_size, n_batches, length, seed
        self.Q = max(12 - self.V, 5) if Q is None else Q      # rest views: the leads that are not inputs (tianchi.py:191-195)

    def __len__(self):
        return self.n

    def __iter__(self):
        for i in range(self.n):
            meta = synth.make_batch(self.B, self.V, self.L, seed=self.seed + i, Q=self.Q)
            yield {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in meta.items()}


def build_loaders(cfg, batch_size=32, phases=('train', 'test')):
    rank, world = parallel.rank_world()
    if batch_size % world:
        raise ValueError(f"global batch {batch_size} is not divisible by world size {world}")
    out = []
    if cfg.DATA.get('synthetic', False):
        for ph in phases:
            if ph == 'train':
                out.append(parallel.ShardedLoader(SyntheticLoader(cfg, batch_size, seed=cfg.seed), rank, world))
            else:
                out.append(SyntheticLoader(cfg, batch_size, 2, seed=cfg.seed + 10 ** 6))
        return out
    for path in ([cfg.DATA.train_label_path] if 'train' in phases else []) + [cfg.DATA.test_label_path]:
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} (set DATA.synthetic True to train on synthetic meta batches)")
    from torch.utils.data import DataLoader
    from torch.utils.data.distributed import DistributedSampler
    from .dataset import build_dataset
    for ph in phases:
        ds = build_dataset(cfg, phase=ph)
        if ph == 'train':
            if cfg.DATA.get('weighted_sample', False):
                # reference train_net.py:21-25: WeightedRandomSampler(train_dataset.get_label_weight(), num_samples=5000),
                # num_workers=0.  None of the reference's dataset classes defines get_label_weight(), so the option only
                # works with a dataset that brings it; never fall back to uniform shuffling silently.
                if world > 1:
                    raise NotImplementedError("DATA.weighted_sample under data parallelism: the weighted draw is not sharded")
                if not hasattr(ds, 'get_label_weight'):
                    raise NotImplementedError(f"DATA.weighted_sample needs {type(ds).__name__}.get_label_weight() "
                                              "(the reference's datasets do not define it either)")
                from torch.utils.data import WeightedRandomSampler
                dl = DataLoader(ds, batch_size=batch_size, sampler=WeightedRandomSampler(ds.get_label_weight(), num_samples=5000),
                                num_workers=0, drop_last=True, collate_fn=_collate, pin_memory=True)
            elif world > 1:
                sampler = DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True, seed=cfg.seed, drop_last=True)
                dl = DataLoader(ds, batch_size=batch_size // world, sampler=sampler, num_workers=16, drop_last=True,
                                collate_fn=_collate_train, pin_memory=True, persistent_workers=True, prefetch_factor=4)
            else:
                dl = DataLoader(ds, batch_size=batch_size, shuffle=True, num_workers=16, drop_last=True,
                                collate_fn=_collate_train, pin_memory=True, persistent_workers=True, prefetch_factor=4)
        else:
            dl = DataLoader(ds, batch_size=batch_size, num_workers=8, drop_last=True, collate_fn=_collate, pin_memory=True)
        out.append(dl)
    return out


# what Solver.run_one_epoch(phase='train') reads of a `meta` batch (solver.py:157-189); the per-item dict of the dataset keeps
# the reference's full schema, but shipping the unused 12-lead float64 fields (`ori_data`, `rest_view`: 100 KB per item)
# from the workers to the trainer halves the loader's throughput
_TRAIN_FIELDS = ('data', 'rois', 'input_theta', 'target_view', 'target_theta', 'noise', 'rest_theta')


def _collate_train(items):
    return _collate([{k: it[k] for k in _TRAIN_FIELDS if k in it} for it in items])


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
    solver.train(DevicePrefetcher(train_dl, solver.device), DevicePrefetcher(test_dl, solver.device))
