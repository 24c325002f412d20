"""`build_dataset(cfg, phase)` of reference codes/dataset/__init__.py:5-16 (Tianchi per-beat dataset only; the PTB
loader needs pickles that are not part of the reference repo)."""
from .tianchi import EcgTianChiInterval


def build_dataset(cfg, phase):
    if cfg.DATA.dataset == 'tianchi':
        return EcgTianChiInterval(cfg, phase)
    raise NotImplementedError("{} is not support".format(cfg.DATA.dataset))
