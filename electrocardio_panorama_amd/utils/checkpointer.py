"""Checkpoint save / auto-resume with the on-disk layout of reference codes/utils/checkpointer.py:18-98:
`<dir>/<name>.pkl` = torch.save({'model', 'optimizer', 'scheduler', **extra}) plus a `last_checkpoint`
pointer file; a leading `module.` (nn.DataParallel) is stripped from loaded keys."""
import os

import torch


class CheckPointer:
    _last_checkpoint_name = 'last_checkpoint'

    def __init__(self, model, optimizer=None, scheduler=None, save_dir="", save_to_disk=True):
        self.model, self.optimizer, self.scheduler = model, optimizer, scheduler
        self.save_dir, self.save_to_disk = save_dir, save_to_disk

    def save(self, name, **kwargs):
        if not self.save_dir or not self.save_to_disk:
            return
        os.makedirs(self.save_dir, exist_ok=True)
        data = {'model': self.model.state_dict()}
        if self.optimizer is not None:
            data['optimizer'] = self.optimizer.state_dict()
        if self.scheduler is not None:
            data['scheduler'] = self.scheduler.state_dict()
        core = getattr(self.model, 'module', self.model)
        if hasattr(core, 'h2_state'):
            # beyond the reference's file (checkpointer.py:24-36): the operand magnitudes of the split-fp16 call sites, so that a resumed
            # run is bit-identical to the uninterrupted one (a reference-side loader ignores the extra key)
            data['h2_state'] = core.h2_state()
        data.update(kwargs)
        save_file = os.path.join(self.save_dir, '{}.pkl'.format(name))
        torch.save(data, save_file)
        with open(os.path.join(self.save_dir, self._last_checkpoint_name), 'w') as f:
            f.write(save_file)

    def load(self, f=None, best_valid=False):
        """Which file is loaded follows the reference (checkpointer.py:40-60): an explicit path `f` (cfg.MODEL.resume,
        `Solver.val(epoch=n)`) wins; otherwise `best_valid.pkl` when `best_valid`, else the `last_checkpoint` pointer.
        Returns the extra entries of the checkpoint ({} when there is nothing to load).  One deliberate difference: an
        explicit path is honoured even before any `last_checkpoint` pointer exists in `save_dir` (the reference
        silently starts from scratch then)."""
        if not f:
            if not self.has_checkpoint():
                return {}
            f = os.path.join(self.save_dir, 'best_valid.pkl') if best_valid else self.get_checkpoint_file()
        if not f or not os.path.exists(f):
            if f:
                raise FileNotFoundError(f)
            return {}
        checkpoint = torch.load(f, map_location='cpu')
        model_sd = {(k[7:] if k.startswith('module.') else k): v for k, v in checkpoint.pop('model').items()}
        self.model.load_state_dict(model_sd)
        h2 = checkpoint.pop('h2_state', None)
        core = getattr(self.model, 'module', self.model)
        if h2 is not None and hasattr(core, 'load_h2_state'):
            core.load_h2_state(h2)
        if 'optimizer' in checkpoint and self.optimizer:
            self.optimizer.load_state_dict(checkpoint.pop('optimizer'))
        if 'scheduler' in checkpoint and self.scheduler:
            self.scheduler.load_state_dict(checkpoint.pop('scheduler'))
        return checkpoint

    def has_checkpoint(self):
        return bool(self.save_dir) and os.path.exists(os.path.join(self.save_dir, self._last_checkpoint_name))

    def get_checkpoint_file(self):
        try:
            with open(os.path.join(self.save_dir, self._last_checkpoint_name), 'r') as f:
                return f.read().strip()
        except IOError:
            return ''
