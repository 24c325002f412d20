"""`python -m electrocardio_panorama_amd.main --config-file config/nef_net.yml` -- reference codes/main.py:1-30."""
import argparse
import os

from .config import cfg, resolve_config_path
from .train_net import main


def run(argv=None):
    parser = argparse.ArgumentParser(description='ecg generation')
    parser.add_argument('--config-file', default="", metavar="FILE", help="path to config file", type=str)
    parser.add_argument('opts', nargs=argparse.REMAINDER, help="KEY VALUE overrides, e.g. SOLVER.epochs 2")
    args = parser.parse_args(argv)
    if args.config_file != '':
        cfg.merge_from_file(resolve_config_path(args.config_file))
    if args.opts:
        cfg.merge_from_list(args.opts)
    print('Using config: ', cfg)
    cfg.desc = args.config_file.split('/')[-1].replace('.yml', '') or cfg.desc
    cfg.output_dir = os.path.join(cfg.output_dir, cfg.desc)
    main(cfg)


if __name__ == '__main__':
    run()
