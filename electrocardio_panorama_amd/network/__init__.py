"""Factory functions of reference codes/network/__init__.py:7-24."""
from torch.nn import MSELoss, CrossEntropyLoss

from .model_nefnet import Model_nefnet
from .loss import losswrapper, MSELead  # noqa: F401


def build_model(cfg):
    model_name = cfg.MODEL.model
    if model_name == 'model_nefnet':
        return Model_nefnet(theta_encoder_len=cfg.MODEL.theta_L, lead_num=cfg.DATA.lead_num)
    raise ValueError('build model: model name error')


def build_loss(cfg):
    loss_name = cfg.MODEL.loss
    if loss_name == 'v1':
        return losswrapper
    if loss_name == 'ce':
        return CrossEntropyLoss()
    if loss_name == 'mse':
        return MSELoss()
    raise ValueError('build loss: loss name error')
