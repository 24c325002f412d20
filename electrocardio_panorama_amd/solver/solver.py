"""Training driver with the surface of reference codes/solver/solver.py:16-245.

Scope: the train-phase body of `run_one_epoch` -- H2D of the meta dict,
model call, losswrapper, backward, optimiser step -- runs entirely on the device with no per-iteration
host synchronisation (the reference issues ~10 D2H copies per step, solver.py:179,189,236-240; here the
four loss scalars stay on the device and are fetched once per epoch).  The test phase reproduces the reference's
metric bookkeeping (solver.py:190-230): `loss_unsperv` on the last four rest views, PSNR / SSIM split into generated
(`gen`, the last `gen_num` rest views) and regressed (`reg`) leads, plus per-lead numbers; TensorBoard is optional."""
import os

import numpy as np
import torch
import torch.distributed as dist

from ..network import build_model, build_loss
from ..utils import CheckPointer
from ..utils.metric import PSNR, SSIM
from .optim_scheduler import get_optimizer, get_lr_scheduler


class Solver:
    def __init__(self, cfg, use_tensorboardx=True, collect_views=False):
        self.cfg = cfg
        self.output_dir = os.path.join(cfg.output_dir, cfg.desc)
        self.desc = cfg.desc
        self.collect_views = collect_views
        self.model = build_model(cfg).float()
        self.loss = build_loss(cfg)
        self._init_model_device()
        self.summary_writer = None

    def _init_model_device(self):
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device: this build has no CPU path (use the oracle for CPU runs)")
        local = int(os.environ.get("LOCAL_RANK", 0))
        self.device = torch.device('cuda', local if dist.is_available() and dist.is_initialized() else 0)
        self.model.to(self.device)

    def train(self, dl_train, dl_test=None):
        optimizer = get_optimizer(self.cfg, self.model.parameters())
        scheduler = get_lr_scheduler(self.cfg, optimizer)
        checkpointer = CheckPointer(self.model, optimizer, scheduler, self.output_dir)
        extra = checkpointer.load(self.cfg.MODEL.resume)
        start_epoch = extra.get('epoch', 0)
        best = extra.get('best_test_psnr_gen', 0.)
        save_arguments = {}
        for epoch in range(start_epoch, self.cfg.SOLVER.epochs):
            train_losses = self.run_one_epoch(dl_train, phase='train', optim=optimizer)[0]
            scheduler.step()
            msg = 'Epoch {}: train_loss: {}'.format(epoch, float(np.mean(train_losses, axis=0)[0]))
            psnr_gen = 0.
            if dl_test is not None:
                test_losses, _, _, _, mertics_all, _, _ = self.run_one_epoch(dl_test, phase='test')
                psnr_gen = float(np.mean(mertics_all, axis=0)[0])
                msg += ', test_loss: {}, psnr_gen: {}'.format(float(np.mean(test_losses, axis=0)[0]), psnr_gen)
            print(msg)
            save_arguments.update(psnr_gen=psnr_gen, epoch=epoch)
            if not dist.is_initialized() or dist.get_rank() == 0:
                checkpointer.save('epoch_{}'.format(epoch), **save_arguments)
                if psnr_gen > best:
                    best = psnr_gen
                    save_arguments['best_test_psnr_gen'] = best
                    checkpointer.save('best_valid', **save_arguments)

    def _to_device(self, meta):
        dev = self.device
        t = lambda v: torch.as_tensor(v).to(dev, non_blocking=True)   # noqa: E731
        return (t(meta['data']), t(meta['rois']), t(meta['input_theta']), t(meta['target_view']).unsqueeze(1),
                t(meta['target_theta']), t(meta['noise']).unsqueeze(1))

    def run_one_epoch(self, dl, phase, optim=None):
        if phase == 'train':
            self.model.train()
        elif phase == 'test':
            self.model.eval()
        else:
            raise ValueError('phase param not found.')
        dev_losses, gt_views, predict_views, input_views, rest_views, mertics_all, rois_all = [], [], [], [], [], [], []
        mertics_gen_singlelead = []
        for meta in dl:
            source_data, rois, input_theta, target_view, target_theta, noise = self._to_device(meta)
            rest_theta = torch.as_tensor(meta['rest_theta']).to(self.device) if 'rest_theta' in meta else None
            if phase == 'train':
                out, shuf_p, shuf_l = self.model(source_data, input_theta, target_theta, rois, rest_theta=rest_theta,
                                                 phase='train')
                if self.cfg.DATA.noise:
                    out = out + noise
                losses = self.loss(out, shuf_p, shuf_l, target_view, self.cfg)
                dev_losses.append(torch.stack([l_.detach() for l_ in losses]))
                losses[0].backward()
                optim.step()
                optim.zero_grad()
                if self.collect_views:
                    predict_views += [x for x in out.squeeze(1).detach().cpu().numpy()]
            else:
                rest_view = torch.as_tensor(meta['rest_view']).to(self.device, torch.float32)
                out, shuf_p, shuf_l, rest_out = self.model(source_data, input_theta, target_theta, rois,
                                                           rest_theta=rest_theta, phase='test')
                losses = self.loss(out, shuf_p, shuf_l, target_view, self.cfg, rest_out[:, -4:, :].contiguous(),
                                   rest_view[:, -4:, :].contiguous())
                dev_losses.append(torch.stack([l_.detach() for l_ in losses]))
                ro, rv, rn = rest_out.cpu().numpy(), rest_view.cpu().numpy(), rois.cpu().numpy()
                # which rest views are "generated" (never supervised): solver.py:197-201
                gen_num = 6 if self.cfg.DATA.lead_num == 336 else 4
                super_mode = str(self.cfg.DATA.get('super_mode', 'normal'))
                if super_mode != 'normal' and super_mode[-1].isdigit():
                    gen_num = int(super_mode[-1])
                if self.cfg.DATA.get('dataset', 'tianchi') == 'mit' or super_mode[-1] == '0' or super_mode == '_mit':
                    psnr_gen = psnr_reg = PSNR(ro, rv)
                    ssim_gen = ssim_reg = SSIM(ro, rv)
                else:
                    psnr_gen, psnr_reg = PSNR(ro[:, -gen_num:], rv[:, -gen_num:], rn), PSNR(ro[:, :-gen_num], rv[:, :-gen_num], rn)
                    ssim_gen, ssim_reg = SSIM(ro[:, -gen_num:], rv[:, -gen_num:], rn), SSIM(ro[:, :-gen_num], rv[:, :-gen_num], rn)
                    single = []
                    for i in range(gen_num):
                        k = ro.shape[1] - gen_num + i
                        single.append([PSNR(ro[:, k:k + 1], rv[:, k:k + 1], rn), SSIM(ro[:, k:k + 1], rv[:, k:k + 1], rn)])
                    mertics_gen_singlelead.append(single)
                mertics_all.append([psnr_gen, psnr_reg, ssim_gen, ssim_reg])
                predict_views += [x for x in ro]
                rest_views += [x for x in rv]
            if self.collect_views:
                gt_views += [x for x in target_view.squeeze(1).cpu().numpy()]
                input_views += [x for x in source_data.cpu().numpy()]
                rois_all += [x for x in rois.cpu().numpy()]
        losses = torch.stack(dev_losses).cpu().numpy().tolist() if dev_losses else []
        if phase == 'train':
            return losses, gt_views, predict_views, input_views, mertics_all, rois_all
        return losses, rest_views, predict_views, input_views, mertics_all, rois_all, mertics_gen_singlelead
