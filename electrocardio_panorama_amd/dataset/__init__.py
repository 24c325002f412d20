"""`build_dataset(cfg, phase)` of reference codes/dataset/__init__.py:5-16."""
from .ptbv2 import PTBV2, HeartBeat
from .tianchi import EcgTianChiInterval

_PTB_PATHS = dict(train_pkl_path='data/ptb/ptb_pkl_data/train_ptb.pkl', test_pkl_path='data/ptb/ptb_pkl_data/test_ptb.pkl',
                  train_label_path='data/ptb/ptb_train.txt', test_label_path='data/ptb/ptb_test.txt',
                  train_data_root='data/ptb/ptb-diag_preprocess')


def build_dataset(cfg, phase, ptb_paths=None):
    """`ptb_paths`: optional overrides of the PTB locations the reference hard-codes (dataset/__init__.py:10-14)."""
    if cfg.DATA.dataset == 'tianchi':
        return EcgTianChiInterval(cfg, phase)
    if cfg.DATA.dataset == 'ptbv2':
        for k, v in {**_PTB_PATHS, **(ptb_paths or {})}.items():
            cfg.DATA[k] = v
        return PTBV2(cfg, phase)
    raise NotImplementedError("{} is not support".format(cfg.DATA.dataset))
