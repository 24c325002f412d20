"""`build_model(cfg)` / `build_loss(cfg)`: the factory surface of reference codes/network/__init__.py:7-24
(same accepted names, same ValueError on anything else)."""
from torch.nn import CrossEntropyLoss, MSELoss

from .loss import MSELead, losswrapper  # noqa: F401
from .model_nefnet import Model_nefnet, Model_nefnet2

_MODELS = {"model_nefnet": lambda cfg: Model_nefnet(theta_encoder_len=cfg.MODEL.theta_L, lead_num=cfg.DATA.lead_num),
           # not in the reference's factory (its model_nefnet2.py is unreachable from config); an extension name
           "model_nefnet2": lambda cfg: Model_nefnet2(theta_encoder_len=cfg.MODEL.theta_L, lead_num=cfg.DATA.lead_num)}
_LOSSES = {"v1": lambda: losswrapper, "ce": CrossEntropyLoss, "mse": MSELoss}


def build_model(cfg):
    try:
        make = _MODELS[cfg.MODEL.model]
    except KeyError:
        raise ValueError('build model: model name error') from None
    return make(cfg)


def build_loss(cfg):
    try:
        make = _LOSSES[cfg.MODEL.loss]
    except KeyError:
        raise ValueError('build loss: loss name error') from None
    return make()
