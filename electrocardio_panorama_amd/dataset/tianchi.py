"""Per-heartbeat Tianchi samples with the reference's `meta` schema (host side; not on the train-step path).

Restates what reference codes/dataset/tianchi.py:84-225 (`EcgTianChiInterval.__getitem__`) produces -- same fields,
same value conventions and the same consumption order of Python `random` / `numpy.random`, so a seeded run yields the
same batches: an `(8, 5000)` recording is extended by the four derived limb leads (:88-93), one annotated beat is drawn
(:97), cropped, min-max normalised to [0, 1] (:109-111), described by seven contiguous ROIs relative to its P-onset
(:103-106) and padded/cropped to 512 samples (:199-211); input / target / rest views follow the lead tables of
:127-190.  On-disk format: `<data_root>/<id>.npy` + `<label_root>/<id>.json` with the six P/R/T on/off index lists
(reference codes/README.md:13); `.npz` bundles with the same arrays are accepted too.
"""
import json
import os
import random

import numpy as np

from ..synth import LEAD_THETA

BEAT_LEN = 512
_LIMB, _CHEST = [2, 4, 6, 7], [0, 1, 8, 9]          # "supervision_lead_lamb" / "_chest" of the reference


def _others(*taken):
    used = set().union(*taken)
    return [i for i in range(12) if i not in used]


def _lead_plan(lead_num, super_mode, data_mode):
    """(select, supervised, unsupervised, keep_supervised_as_rest) for the configured input-lead scheme (:127-191)."""
    sup, unsup = _LIMB + _CHEST, _others(_LIMB + _CHEST)
    sel = None
    if lead_num == 3:
        n_limb = random.randint(1, 2)                 # drawn in every mode, as the reference does (:128)
        if data_mode == 'input_fix':
            if super_mode == 'IIv2v5_v4I_372':
                sel, unsup = [1, 3, 6], [5, 0]
                sup = _others(sel, unsup)
        else:
            sel = random.sample(_LIMB, n_limb) + random.sample(_CHEST, 3 - n_limb)
    elif lead_num == 12 and super_mode == '_12120':
        sel, sup, unsup = list(range(12)), list(range(12)), []
    elif lead_num == 9:
        sup = [0, 1, 3]
        sel, unsup = _others(sup), []
    elif lead_num == 8 and super_mode == '_8120':
        sel, sup, unsup = list(range(8)), list(range(12)), []
    elif lead_num == 4:
        sel = [2, 6, 0, 8]
        if super_mode == '_480':
            sup, unsup = _others(sel), []
        elif super_mode == '_462':
            unsup = [4, 11]
            sup = _others(sel, unsup)
    elif lead_num == 5:
        table = {'_552': [4, 11], '_561': [4], '_570': []}
        if super_mode in table:
            sel, unsup = [2, 6, 0, 8, 10], table[super_mode]
            sup = _others(sel, unsup)
    elif lead_num == 2:
        sel = [1, 6]
        if super_mode == '_228':
            sup = [1, 6, 9, 3]
            unsup = _others(sup)
        elif super_mode == '_2100':
            sup, unsup = _others(sel), []
    elif lead_num == 1:
        sel = [1]
        table = {'_1110': [], '_1101': [4], '_192': [4, 11]}
        if super_mode in table:
            unsup = table[super_mode]
            sup = _others(sel, unsup)
    else:
        raise KeyError("WORANG lead num: {}".format(lead_num))
    if sel is None:
        raise KeyError("no input-lead scheme for lead_num={} super_mode={}".format(lead_num, super_mode))
    return sel, sup, unsup, super_mode in ('_12120', '_3120', '_8120')


def _fit(a, n=BEAT_LEN):
    """Zero-pad or crop the last axis to n samples."""
    if a.shape[-1] >= n:
        return a[..., :n]
    pad = [(0, 0)] * (a.ndim - 1) + [(0, n - a.shape[-1])]
    return np.pad(a, pad, mode='constant')


class EcgTianChiInterval:
    """Map-style dataset (usable with torch.utils.data.DataLoader); `cfg` is the config tree of config/default.py."""

    def __init__(self, cfg, phase, transform=None):
        self.cfg, self.phase, self.transform = cfg, phase, transform
        self.theta = LEAD_THETA.copy()
        label_path = cfg.DATA.train_label_path if phase == 'train' else cfg.DATA.test_label_path
        with open(label_path) as f:
            self.dataset = f.read().splitlines()
        self.data_root, self.label_dir = cfg.DATA.train_data_root, cfg.DATA.train_label_root

    def __len__(self):
        return len(self.dataset)

    def _load(self, name):
        stem = name.replace('.json', '')
        npz = os.path.join(self.data_root, stem + '.npz')
        if os.path.exists(npz):
            z = np.load(npz)
            return z['signal'].astype(np.float64), {k.replace('_', ' '): z[k].tolist() for k in z.files if k != 'signal'}
        sig = np.load(os.path.join(self.data_root, stem + '.npy')).astype(np.float64)
        with open(os.path.join(self.label_dir, name)) as f:
            return sig, json.load(f)

    def __getitem__(self, index):
        sig, label = self._load(self.dataset[index])
        lead_i, lead_ii = sig[0:1], sig[1:2]
        sig = np.concatenate([sig, lead_ii - lead_i, -0.5 * (lead_i + lead_ii), lead_i - 0.5 * lead_ii,
                              lead_ii - 0.5 * lead_i], axis=0)                       # + III, aVR, aVL, aVF
        beat = random.sample(range(len(label['P on']) - 1), k=1)[0]
        p_on, p_off, r_on, r_off, t_on, t_off = (label[k][beat] for k in ('P on', 'P off', 'R on', 'R off', 'T on', 'T off'))
        end = label['P on'][beat + 1] if beat + 1 < len(label['P on']) else sig.shape[-1]
        rois = np.array([[p_on, p_off], [p_off, r_on], [r_on, r_off], [r_off, t_on], [t_on, t_off], [t_off, end],
                         [end, BEAT_LEN + p_on]]) - p_on
        sig = sig[:, p_on:end]
        lo, hi = np.min(sig), np.max(sig)
        sig = (sig - lo) / (hi - lo)
        quiet = sig[:, (rois[5][0] + rois[5][1]) // 2: rois[5][1]]                   # second half of the T-P segment
        noise = np.random.normal(loc=0, scale=np.std(quiet, axis=1), size=(sig.shape[-1], 12))
        theta = self.theta
        if self.cfg.MODEL.jitter_factor > 0 and self.phase == 'train':
            theta = theta + np.random.normal(scale=self.cfg.MODEL.jitter_factor / 180 * np.pi, size=theta.shape)
        sel, sup, unsup, keep = _lead_plan(self.cfg.DATA.lead_num, self.cfg.DATA.super_mode, self.cfg.DATA.train_data_mode)
        rest = list(sup) if keep else [x for x in sup if x not in sel]
        target = random.sample(rest, 1)[0]
        rest = rest + list(unsup)                                                    # unsupervised leads last
        return {
            'data': _fit(sig[sel]).astype(np.float32),
            'rois': rois.astype(np.int64),
            'input_theta': theta[sel].astype(np.float32),
            'target_view': _fit(sig[target]).astype(np.float32),
            'target_theta': theta[target].astype(np.float32),
            'id': self.dataset[index],
            'ori_data': _fit(sig),
            'rest_view': _fit(sig[rest]),
            'rest_theta': theta[rest].astype(np.float32),
            'noise': _fit(noise[:, target]).astype(np.float32),
            'unsupervision_lead_name': list(unsup),
        }
