"""`python -m electrocardio_panorama_amd.val_net --config-file config/nef_net.yml [--epoch N]` -- reference
codes/val_net.py:9-48: build the test loader, load `best_valid.pkl` (or `epoch_N.pkl`) and print PSNR / SSIM."""
import argparse
import os

import torch

from . import parallel
from .config import cfg, resolve_config_path
from .prefetch import DevicePrefetcher
from .solver import Solver
from .train_net import build_loaders
from .utils import seed_torch


def main(cfg, epoch=-1):
    parallel.init_from_env()
    seed_torch(seed=cfg.seed)
    os.makedirs(os.path.join(cfg.output_dir, cfg.desc), exist_ok=True)
    test_dl, = build_loaders(cfg, phases=('test',))
    solver = Solver(cfg, use_tensorboardx=False)
    with torch.no_grad():
        return solver.val(DevicePrefetcher(test_dl, solver.device), epoch=epoch)


def run(argv=None):
    parser = argparse.ArgumentParser(description='ecg generation')
    parser.add_argument('--config-file', default="", metavar="FILE", help="path to config file", type=str)
    parser.add_argument('--epoch', default=-1, type=int)
    parser.add_argument('--ds', default='tianchi', type=str)
    parser.add_argument('opts', nargs=argparse.REMAINDER, help="KEY VALUE overrides")
    args = parser.parse_args(argv)
    if args.config_file != '':
        cfg.merge_from_file(resolve_config_path(args.config_file))
    if args.opts:
        cfg.merge_from_list(args.opts)
    cfg.desc = args.config_file.split('/')[-1].replace('.yml', '') or cfg.desc
    cfg.output_dir = os.path.join(cfg.output_dir, cfg.desc)
    return main(cfg, epoch=args.epoch)


if __name__ == '__main__':
    run()
