"""Per-heartbeat PTB samples with the reference's `meta` schema (host side).

Restates reference codes/dataset/ptbv2.py:39-157 (`PTBV2.__getitem__`) and :169-212 (`HeartBeatList`): PTB recordings
are 12-lead `(12, n)` arrays in the order I, II, III, aVR, aVL, aVF, V1..V6, one directory per patient holding
`<record>.npy` + `<record>.json` (the six P/R/T on/off index lists).  Every annotated beat but the last of a record
becomes one `HeartBeat` (signal cropped from its P onset to the next P onset, seven contiguous ROIs relative to the P
onset, the last one running to sample 512: :186-196).  A sample re-orders the leads to the model's table order
(I, II, V1..V6, III, aVR, aVL, aVF: :42), min-max normalises, draws the noise / angle jitter / lead plan / target in the
reference's order of `numpy.random` and `random` calls, and pads or crops everything to 512 samples.

Differences by design: the beat cache is a single `.npz` (arrays only) next to where the reference would put its
pickle; a pickle written by the reference is still readable (`load_reference_pickle`), and records of one patient are
visited in sorted order (the reference uses the directory order of `os.listdir`, which is file-system dependent);
a beat longer than 512 samples has its noise cropped like every other field (the reference's pad at :142 raises).
"""
import io
import json
import logging
import os
import pickle
import random

import numpy as np

from ..synth import LEAD_THETA
from .tianchi import BEAT_LEN, _fit, _lead_plan

_KEYS = ('P on', 'P off', 'R on', 'R off', 'T on', 'T off')


class HeartBeat:
    """One beat: `data` float64 [12, len] in PTB lead order, `rois_list` int [7, 2] relative to the P onset."""
    def __init__(self, data, rois_list):
        self.data = data
        self.rois_list = rois_list


def split_record(signal, label):
    """All beats of one annotated record (ptbv2.py:183-199)."""
    beats = []
    on = label['P on']
    for i in range(len(on) - 1):
        p_on, p_off, r_on, r_off, t_on, t_off = (label[k][i] for k in _KEYS)
        end = on[i + 1]
        rois = np.array([[p_on, p_off], [p_off, r_on], [r_on, r_off], [r_off, t_on], [t_on, t_off], [t_off, end],
                         [end, BEAT_LEN + p_on]]) - p_on
        beats.append(HeartBeat(signal[:, p_on:end], rois))
    return beats


def read_heartbeats(txt_path, data_root):
    """Walk the patient list (ptbv2.py:176-199)."""
    with open(txt_path) as f:
        patients = f.read().splitlines()
    beats = []
    for patient in patients:
        pdir = os.path.join(data_root, patient)
        for name in sorted(x for x in os.listdir(pdir) if x.endswith('.json')):
            signal = np.load(os.path.join(pdir, name.replace('.json', '.npy'))).astype(np.float64)
            with open(os.path.join(pdir, name)) as f:
                beats.extend(split_record(signal, json.load(f)))
    return beats


class _RefUnpickler(pickle.Unpickler):
    """Resolves the reference's `dataset.ptbv2.HeartBeat` (whatever module path it was pickled under) to HeartBeat."""

    def find_class(self, module, name):
        if name == "HeartBeat":
            return HeartBeat
        return super().find_class(module, name)


def load_reference_pickle(path):
    with open(path, 'rb') as f:
        return _RefUnpickler(io.BytesIO(f.read())).load()


class HeartBeatList:
    """`heart_beats` of a patient list, cached on disk (ptbv2.py:169-212)."""

    def __init__(self, txt_path, data_root, pkl_path):
        cache = os.path.splitext(pkl_path)[0] + '.npz'
        if os.path.exists(cache):
            self.heart_beats = self._load_npz(cache)
        elif os.path.exists(pkl_path):
            logging.info("Loading PTB heartbeats from the reference's pickle...")
            self.heart_beats = load_reference_pickle(pkl_path)
        else:
            self.heart_beats = read_heartbeats(txt_path, data_root)
            self._save_npz(cache)

    @staticmethod
    def _load_npz(path):
        z = np.load(path)
        flat, off, rois = z['signal'], z['offsets'], z['rois']
        return [HeartBeat(flat[:, off[i]:off[i + 1]], rois[i]) for i in range(len(rois))]

    def _save_npz(self, path):
        os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
        lens = [hb.data.shape[-1] for hb in self.heart_beats]
        flat = np.concatenate([hb.data for hb in self.heart_beats], axis=1) if lens else np.zeros((12, 0))
        np.savez(path, signal=flat, offsets=np.concatenate([[0], np.cumsum(lens)]).astype(np.int64),
                 rois=np.stack([hb.rois_list for hb in self.heart_beats]) if lens else np.zeros((0, 7, 2), np.int64))


class PTBV2:
    """Map-style dataset (usable with torch.utils.data.DataLoader); `cfg` is the config tree of config/default.py."""

    def __init__(self, cfg, phase, transform=None):
        self.cfg, self.phase, self.transform = cfg, phase, transform
        self.theta = LEAD_THETA.copy()
        train = phase == 'train'
        self.dataset = HeartBeatList(cfg.DATA.train_label_path if train else cfg.DATA.test_label_path,
                                     cfg.DATA.train_data_root,
                                     cfg.DATA.train_pkl_path if train else cfg.DATA.test_pkl_path).heart_beats

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, index):
        hb = self.dataset[index]
        sig, rois = hb.data, np.asarray(hb.rois_list)
        sig = np.concatenate([sig[0:2], sig[6:], sig[2:6]], axis=0)                 # -> I, II, V1..V6, III, aVR, aVL, aVF
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
            'ori_data': _fit(sig),
            'rest_view': _fit(sig[rest]),
            'rest_theta': theta[rest].astype(np.float32),
            'noise': _fit(noise[:, target]).astype(np.float32),
            'unsupervision_lead_name': list(unsup),
        }
