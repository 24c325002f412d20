"""Training driver with the surface of reference codes/solver/solver.py:16-245 (`Solver.train`, `Solver.val`,
`Solver.run_one_epoch`, `write_tensorboardx`).

Scope.  The train-phase body of `run_one_epoch` -- H2D of the meta dict, model
call, losswrapper, backward, optimiser step -- runs on the device with NO per-iteration host synchronisation: the
reference issues ~10 blocking D2H copies per step (solver.py:179,189,236-240); here the loss scalars, the predicted
views and the metric tables are copied to pinned host memory asynchronously and read once at the end of the epoch,
so the returned lists are the reference's while the launch queue never drains.  The test phase reproduces the
reference's bookkeeping (solver.py:190-230): `loss_unsperv` on the last four rest views, PSNR / SSIM split into
generated (`gen`, the last `gen_num` rest views) and regressed (`reg`) leads plus per-lead numbers -- computed on the
device by `nef_view_metrics`, one launch per batch.

Data parallelism replaces `nn.DataParallel` (solver.py:32-34) with one process per GPU (`parallel.py`): the loaders
hand every rank its shard, FusedSGD all-reduces the flat gradient, rank 0 owns the BatchNorm running statistics,
the scalar log and the checkpoints."""
import json
import os

import numpy as np
import torch
from .. import _env
import torch.distributed as dist

from .. import engine, ops, parallel
from ..network import build_model, build_loss
from ..utils import CheckPointer
from .optim_scheduler import get_optimizer, get_lr_scheduler


class JsonlScalarWriter:
    """`add_scalar(tag, value, global_step)` sink used when neither tensorboardX nor torch.utils.tensorboard is
    installed (solver.py:8,24): one JSON object per scalar in `<logdir>/scalars.jsonl`."""

    def __init__(self, logdir):
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, 'scalars.jsonl')

    def add_scalar(self, tag, scalar_value, global_step=None):
        with open(self.path, 'a') as f:
            f.write(json.dumps({'tag': tag, 'value': float(scalar_value), 'step': global_step}) + '\n')

    def close(self):
        pass


def make_summary_writer(logdir):
    try:
        import tensorboardX
        return tensorboardX.SummaryWriter(logdir=logdir)
    except ImportError:
        pass
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(log_dir=logdir)
    except ImportError:
        return JsonlScalarWriter(logdir)


class _HostSink:
    """Device tensors queued for the host without a per-iteration synchronisation: each is copied into a pinned staging
    buffer on the current stream; the staging buffers form a bounded ring (RING entries per sink), and a buffer is drained
    into pageable numpy memory -- after waiting for ITS copy event only -- when the ring comes round to it again.  The
    pinned footprint is therefore RING tensors per sink, not an epoch's worth, and after the first RING adds no
    hipHostMalloc happens (buffers are reused when the shape matches)."""
    RING = 4

    def __init__(self):
        self.done = []           # pageable numpy arrays, in order
        self.ring = []           # [pinned buffer, event, filled?]
        self.pos = 0

    def _drain(self, slot):
        if slot[2]:
            slot[1].synchronize()
            self.done.append(slot[0].numpy().copy())
            slot[2] = False

    def add(self, t):
        t = t.detach()
        if len(self.ring) < self.RING:
            self.ring.append([torch.empty(t.shape, dtype=t.dtype, pin_memory=True), torch.cuda.Event(), False])
            slot = self.ring[-1]
        else:
            slot = self.ring[self.pos]
            self._drain(slot)
            if slot[0].shape != t.shape or slot[0].dtype != t.dtype:       # e.g. the final partial batch
                slot[0] = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        self.pos = (self.pos + 1) % self.RING
        slot[0].copy_(t, non_blocking=True)
        slot[1].record()
        slot[2] = True

    def arrays(self):
        n = len(self.ring)
        start = self.pos if n == self.RING else 0
        for i in range(n):           # oldest first
            self._drain(self.ring[(start + i) % n])
        return self.done

    def rows(self):
        return [x for a in self.arrays() for x in a]


def _is_main():
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


class Solver:
    def __init__(self, cfg, use_tensorboardx=True, collect_views=True):
        self.cfg = cfg
        self.output_dir = os.path.join(cfg.output_dir, cfg.desc)
        self.desc = cfg.desc
        self.collect_views = collect_views       # False: skip the per-view host lists (benchmarks)
        self.model = build_model(cfg).float()
        self.loss = build_loss(cfg)
        self._init_model_device()
        if self.desc != 'debug' and use_tensorboardx and _is_main():       # solver.py:23-26
            self.summary_writer = make_summary_writer(os.path.join(cfg.output_dir, 'tf_logs'))
        else:
            self.summary_writer = None

    def _init_model_device(self):
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device: this build has no CPU path (use the oracle for CPU runs)")
        self.device = torch.device('cuda', parallel.local_rank() if dist.is_available() and dist.is_initialized() else 0)
        self.model.to(self.device)

    def write_tensorboardx(self, scalars, names, epoch):
        for i in range(len(scalars)):
            self.summary_writer.add_scalar(names[i], scalars[i], global_step=epoch)

    def train(self, dl_train, dl_test=None):
        optimizer = get_optimizer(self.cfg, self.model.parameters())
        scheduler = get_lr_scheduler(self.cfg, optimizer)
        checkpointer = CheckPointer(self.model, optimizer, scheduler, self.output_dir)
        extra = checkpointer.load(self.cfg.MODEL.resume)
        max_epochs = self.cfg.SOLVER.epochs
        start_epoch = extra.get('epoch', 0)
        best_test_psnr_gen = extra.get('best_test_psnr_gen', 0.)
        self.model.dropout_epoch = start_epoch          # a resumed run must not replay epoch 0's dropout masks
        print('the latest best_test_psnr_gen is {:06f}'.format(best_test_psnr_gen))
        save_arguments = {}
        for epoch in range(start_epoch, max_epochs):
            print('---------------------------------{}---{}-------------------------------------'.format(self.cfg.desc, epoch))
            if hasattr(getattr(dl_train, 'sampler', None), 'set_epoch'):
                dl_train.sampler.set_epoch(epoch)         # DistributedSampler: a new shuffle per epoch
            self.model.dropout_epoch = epoch
            train_losses = self.run_one_epoch(dl_train, phase='train', optim=optimizer, collect_views=False)[0]
            scheduler.step()
            parallel.broadcast_buffers(self.model)        # rank 0's BatchNorm running statistics are the model's
            tl = np.mean(train_losses, axis=0)
            train_loss_all = float(tl[0])
            scalars, names = [train_loss_all, float(tl[1]), float(tl[2]), float(tl[3])], \
                ['train_loss_all', 'train_loss_1', 'train_loss_2', 'train_3']
            psnr_gen = psnr_reg = 0.
            msg = 'Epoch {}: train_loss: {}'.format(epoch, train_loss_all)
            if dl_test is not None:
                test_losses, _, _, _, mertics_all, _, single = self.run_one_epoch(dl_test, phase='test', collect_views=False)
                te = np.mean(test_losses, axis=0)
                psnr_gen, psnr_reg, ssim_gen, ssim_reg = (float(v) for v in np.mean(mertics_all, axis=0))
                # the reference's scalar set and names (solver.py:82-100)
                scalars = [train_loss_all, float(te[0]), float(tl[1]), float(te[1]), float(tl[2]), float(te[2]),
                           float(tl[3]), float(te[3]), float(te[4]), psnr_gen, psnr_reg, ssim_gen, ssim_reg]
                names = ['train_loss_all', 'test_loss_all', 'train_loss_1', 'test_loss_1', 'train_loss_2', 'test_loss_2',
                         'train_3', 'test_3', 'test_unsuperv', 'psnr_gen', 'psnr_reg', 'ssim_gen', 'ssim_reg']
                if len(single) != 0:
                    single = np.array(single)
                    for i in range(single.shape[1]):
                        names += ['psnr_reg_lead_{}'.format(i), 'ssim_reg_lead_{}'.format(i)]
                        scalars += [float(np.mean(single[:, i, 0])), float(np.mean(single[:, i, 1]))]
                msg += ', test_loss: {}'.format(float(te[0]))
                msg += '\npsnr_gen: {}, psnr_reg: {}, ssim_gen:{}, ssim_reg:{}'.format(psnr_gen, psnr_reg, ssim_gen, ssim_reg)
            if self.summary_writer is not None:
                self.write_tensorboardx(scalars, names, epoch)
            print(msg)
            save_arguments['psnr_gen'] = psnr_gen
            save_arguments['psnr_reg'] = psnr_reg
            save_arguments['epoch'] = epoch
            if _is_main():
                checkpointer.save('epoch_{}'.format(epoch), **save_arguments)
                if psnr_gen > best_test_psnr_gen:
                    best_test_psnr_gen = psnr_gen
                    save_arguments['best_test_psnr_gen'] = best_test_psnr_gen
                    save_arguments['epoch'] = epoch
                    checkpointer.save('best_valid', **save_arguments)

    def val(self, dl_test, epoch=-1):
        """solver.py:118-137: load `best_valid.pkl` (epoch == -1) or `epoch_<n>.pkl`, run the test phase, print and
        return (psnr_gen, psnr_reg, ssim_gen, ssim_reg)."""
        self.model.eval()
        optimizer = get_optimizer(self.cfg, self.model.parameters())
        scheduler = get_lr_scheduler(self.cfg, optimizer)
        checkpointer = CheckPointer(self.model, optimizer, scheduler, self.output_dir)
        if epoch == -1:
            extra = checkpointer.load(best_valid=True)
        else:
            extra = checkpointer.load(os.path.join(self.output_dir, 'epoch_{}.pkl'.format(epoch)))
        print('the latest best_test_psnr_gen is {:06f} of epoch {}'.format(extra.get('best_test_psnr_gen', 0.),
                                                                           extra.get('epoch', 0)))
        with torch.no_grad():
            mertics_all = self.run_one_epoch(dl_test, phase='test', collect_views=False)[4]
        psnr_gen, psnr_reg, ssim_gen, ssim_reg = (float(v) for v in np.mean(mertics_all, axis=0))
        print('psnr_gen:{}, psnr_reg:{}, ssim_gen:{}, ssim_reg:{}'.format(psnr_gen, psnr_reg, ssim_gen, ssim_reg))
        return psnr_gen, psnr_reg, ssim_gen, ssim_reg

    def _to_device(self, meta):
        dev = self.device
        t = lambda v: torch.as_tensor(v).to(dev, non_blocking=True)   # noqa: E731
        return (t(meta['data']), t(meta['rois']), t(meta['input_theta']), t(meta['target_view']).unsqueeze(1),
                t(meta['target_theta']), t(meta['noise']).unsqueeze(1))

    def _gen_num(self):
        """How many trailing rest views are 'generated' (never supervised), and whether the split applies at all
        (solver.py:197-203)."""
        gen_num = 6 if self.cfg.DATA.lead_num == 336 else 4
        super_mode = str(self.cfg.DATA.get('super_mode', 'normal'))
        whole = self.cfg.DATA.get('dataset', 'tianchi') == 'mit' or super_mode[-1] == '0' or super_mode == '_mit'
        # the reference eval()s the last character before it looks at `whole` (solver.py:198) and would crash on '_mit';
        # the split count is only consumed when `whole` is false, so it is only parsed then
        if super_mode != 'normal' and not whole:
            if not super_mode[-1].isdigit():
                raise ValueError(f"DATA.super_mode {super_mode!r}: the last character must be the number of generated views")
            gen_num = int(super_mode[-1])
        return gen_num, whole

    def _graphed_step(self, optim, data, keep):
        """The hipGraph stepper for this batch, or None when the step runs eagerly: `cfg.SOLVER.graph` False, the per-view host lists are wanted (the graph returns losses only), the
        model is not the plain Model_nefnet train path, DATA.noise, or the optimiser is not FusedSGD."""
        mode = self.cfg.SOLVER.get('graph', None)
        mode = 'auto' if mode is None or mode == 'auto' else bool(mode)
        if mode is False or keep or self.cfg.DATA.noise:
            return None
        if not hasattr(optim, '_flat') or len(optim.param_groups) != 1:
            return None
        if type(self.model).__name__ != 'Model_nefnet' or _env.get('NEF_SOLVER_GRAPH', '1') == '0':
            return None
        # 'auto' = wherever the conditions above hold, at every batch size: with the convs on the fp16 matrix cores even the
        # full-size step (configs[1]: ~290 launches, 36 ms) loses 2.5-4.5 ms to launch gaps when a busy host issues it from Python
        # (bench.py --no-graph against the default); rounds 1-3 replayed launch-bound shapes only
        st = getattr(self, '_graph_stepper', None)
        if st is None or st.optimizer is not optim:
            from ..graph import GraphedTrainStep
            st = self._graph_stepper = GraphedTrainStep(self.model, self.cfg, optimizer=optim)
        return st

    def _check_h2_range(self, phase, optim):
        """The split-fp16 convs count a launch whose operand is out of fp16's range even after the in-launch range rescue: non-finite
        data (finite operands beyond ops.H2_HEADROOM x of growth are redone with their own scale inside the launch), or any
        clamp of the opt-in producer / consumer kernel form (fp16 ends at 65504; the reference's fp32 nn.Conv1d,
        model_nefnet.py:18-21, has no such limit).  A train step that contained such a launch was SKIPPED on the device when the
        optimiser is FusedSGD (ops.h2_taint): that is reported; counts nothing protected against -- a test-phase forward,
        another optimiser -- mean wrong results were used: raise, unless NEF_H2_ALLOW_CLAMP=1 (then warn).  NEF_H2=0 runs the
        fp32 kernels instead.  Data parallel: the counters are per rank, so they are SUMMED over the process group before anything
        is decided -- every rank raises (or warns) together; a rank that alone saw the bad operand must not leave the others
        blocking in their next collective."""
        clamped, skipped = parallel.sum_counts([ops.h2_clamped(), ops.h2_skipped()], self.device)
        ops.h2_rebase(self.device)       # clamps of this (test / val) phase are not charged to the next train step's taint word
        tails = ops.h2_tail_sites()
        if tails:      # (per rank: a warning only)
            print('WARNING: {} split-fp16 call site(s) measured an operand with more than {:.0%} of its nonzero elements below 2^-11 of its '
                  'largest: practically the whole tensor is outside the format\'s full-precision window (DESIGN.md 3.0); NEF_H2=0 runs the '
                  'fp32 kernels'.format(tails, ops.H2_TAIL_FRAC))
        if not clamped and not skipped:
            return
        msg = ('{} waves of split-fp16 conv launches met an operand outside fp16\'s range in this {} phase (non-finite data; finite '
               'operands are rescaled inside the launch); {} train step(s) were skipped on the device'.format(clamped, phase, skipped))
        protected = phase == 'train' and hasattr(optim, '_flat') and skipped > 0
        if protected or _env.get('NEF_H2_ALLOW_CLAMP') == '1':
            print('WARNING: ' + msg + ('' if protected else ' -- results of those launches are wrong (NEF_H2_ALLOW_CLAMP=1)'))
            return
        raise RuntimeError(msg + '; their results are wrong.  Set NEF_H2=0 (fp32 kernels), or NEF_H2_ALLOW_CLAMP=1 to continue')

    def run_one_epoch(self, dl, phase, optim=None, collect_views=None):
        """solver.py:139-246.  `collect_views` (default: the Solver's setting) = also return the per-view host lists the
        reference returns (inputs, ground truth, predictions, rois); train() / val() only consume losses and metrics and
        pass False, so an epoch stages nothing but a few scalars per step."""
        if phase == 'train':
            self.model.train()
        elif phase == 'test':
            self.model.eval()
        else:
            raise ValueError('phase param not found.')
        keep = self.collect_views if collect_views is None else bool(collect_views)
        losses_s, pred_s, gt_s, in_s, rest_s, rois_s, psnr_s, ssim_s = (_HostSink() for _ in range(8))
        gen_num, whole = self._gen_num() if phase == 'test' else (0, True)
        for meta in dl:
            source_data, rois, input_theta, target_view, target_theta, noise = self._to_device(meta)
            rest_theta = torch.as_tensor(meta['rest_theta']).to(self.device) if 'rest_theta' in meta else None
            stepper = self._graphed_step(optim, source_data, keep) if phase == 'train' else None
            if stepper is not None:
                # forward + losswrapper + backward + SGD as ONE replayed hipGraph (one graph per batch shape: the final partial
                # batch has its own; a learning-rate change re-captures); data parallel: the flat gradient travels as one
                # all-reduce behind the replay (no early bucket under capture)
                losses_s.add(stepper(source_data, input_theta, target_theta, rois, target_view))
            elif phase == 'train':
                out, shuf_p, shuf_l = self.model(source_data, input_theta, target_theta, rois, rest_theta=rest_theta,
                                                 phase='train')
                if keep:
                    pred_s.add(out.squeeze(1))                     # solver.py:179 (before the optional noise)
                if self.cfg.DATA.noise:
                    out = out + noise
                losses = self.loss(out, shuf_p, shuf_l, target_view, self.cfg)
                losses_s.add(torch.stack([l_.detach() for l_ in losses]))
                losses[0].backward()
                optim.step()
                optim.zero_grad()
            else:
                rest_view = torch.as_tensor(meta['rest_view']).to(self.device, torch.float32).contiguous()
                out, shuf_p, shuf_l, rest_out = self.model(source_data, input_theta, target_theta, rois,
                                                           rest_theta=rest_theta, phase='test')
                losses = self.loss(out, shuf_p, shuf_l, target_view, self.cfg, rest_out[:, -4:, :].contiguous(),
                                   rest_view[:, -4:, :].contiguous())
                losses_s.add(torch.stack([l_.detach() for l_ in losses]))
                ps, ss = ops.view_metrics(rest_out.contiguous(), rest_view, None if whole else rois.contiguous())
                psnr_s.add(ps)
                ssim_s.add(ss)
                if keep:
                    pred_s.add(rest_out)                           # solver.py:182
                    rest_s.add(rest_view)
            if keep:
                gt_s.add(target_view.squeeze(1))
                in_s.add(source_data)
                rois_s.add(rois)
        losses = [a.tolist() for a in losses_s.arrays()]
        self._check_h2_range(phase, optim)       # same cadence as the loss read-back above: the device has been waited for anyway
        if phase == 'train':
            return losses, gt_s.rows(), pred_s.rows(), in_s.rows(), [], rois_s.rows()
        mertics_all, mertics_gen_singlelead = [], []
        for ps, ss in zip(psnr_s.arrays(), ssim_s.arrays()):
            if np.isnan(ss).any():
                raise ValueError("win_size exceeds signal extent (a test row is shorter than the 7-tap SSIM window)")
            if whole:
                pg = pr = float(ps.mean())
                sg = sr = float(ss.mean())
            else:
                pg, pr = float(ps[:, -gen_num:].mean()), float(ps[:, :-gen_num].mean())
                sg, sr = float(ss[:, -gen_num:].mean()), float(ss[:, :-gen_num].mean())
                Q = ps.shape[1]
                mertics_gen_singlelead.append([[float(ps[:, Q - gen_num + i].mean()), float(ss[:, Q - gen_num + i].mean())]
                                               for i in range(gen_num)])
            mertics_all.append([pg, pr, sg, sr])
        return losses, rest_s.rows(), pred_s.rows(), in_s.rows(), mertics_all, rois_s.rows(), mertics_gen_singlelead
