"""Optimiser / scheduler factory of reference codes/solver/optim_scheduler.py:5-18.

'sgd' returns FusedSGD: torch.optim.SGD(lr, momentum=0.9) semantics, but one HIP launch over a flat
parameter buffer, and -- when torch.distributed is initialised -- one RCCL all-reduce of the flat
gradient buffer per step (data-parallel training, one process per GPU)."""
import torch
import torch.distributed as dist
from torch.optim import Adam
from torch.optim.lr_scheduler import StepLR, MultiStepLR

from .. import ops, parallel
from ..parallel import reduce_flat_grads


# The flat gradient buffer starts with a 4-word header (16 bytes: word 0 = the taint word of ops.h2_taint, words 1..3 zero) so that
# the gradients behind it, the encoder bucket's all-reduce (header + bucket) and the SGD kernel's reads stay 16-byte aligned.
GRAD_HDR = 4


class FusedSGD(torch.optim.Optimizer):
    """A tainted step (a split-fp16 launch met non-finite data, on any rank) leaves parameters and momentum untouched; the
    BatchNorm running statistics its forward pass already updated are NOT rolled back (DESIGN.md 3.0 "Range")."""

    def __init__(self, params, lr, momentum=0.9):
        super().__init__(params, dict(lr=lr, momentum=momentum))
        self._flat = {}      # group index -> dict(params, p, g, buf)
        parallel.enable_early_reduce()      # this optimiser consumes engine.backward's early gradient bucket (see _reduce)

    def _build(self, gi, live):
        n = sum(p.numel() for p in live)
        dev = live[0].device
        flat_p = torch.empty(n, device=dev, dtype=torch.float32)
        flat_b = torch.zeros(n, device=dev, dtype=torch.float32)
        off = 0
        for p in live:
            k = p.numel()
            flat_p[off:off + k].copy_(p.data.reshape(-1))
            old_ptr = p.data.data_ptr()
            p.data = flat_p[off:off + k].view_as(p.data)          # parameters become views of the flat buffer
            ops.amax_move(old_ptr, p.data.data_ptr())
            st = self.state[p]
            if "momentum_buffer" in st and st["momentum_buffer"] is not None:
                flat_b[off:off + k].copy_(st["momentum_buffer"].reshape(-1))
            st["momentum_buffer"] = flat_b[off:off + k].view_as(p.data)
            off += k
        # gradients: [header (taint word, 3 zero words) | n gradients] -- the word in front (ops.h2_taint: clamps of this step's
        # split-fp16 launches) is summed by the same all-reduce as the gradients, so a clamp on any rank makes every rank skip the update
        g_all = torch.zeros(n + GRAD_HDR, device=dev, dtype=torch.float32)
        self._flat[gi] = dict(ids=[id(p) for p in live], params=live, p=flat_p, buf=flat_b, g_all=g_all, g=g_all[GRAD_HDR:])

    def load_state_dict(self, state_dict):
        """The loaded momentum buffers replace the flat one: drop the flat views so the next step() re-imports them."""
        super().load_state_dict(state_dict)
        self._flat = {}

    @staticmethod
    def _reduce(live, flat, world, flat_all=None):
        """Sum of the flat gradient buffer across ranks (RCCL over xGMI).  When engine.backward already started the
        all-reduce of the gradients that were final early (parallel.early_reduce: a SUFFIX of the parameter order --
        everything behind the per-lead encoder), only the encoder bucket is reduced here and the early bucket's result
        is copied in behind it; otherwise one all-reduce of the whole buffer."""
        early = parallel.take_early() if world > 1 else None
        names = [getattr(p, "_nef_name", None) for p in live]
        split = None
        if early is not None:
            k = len(names) - len(early["names"])
            # the bucket is the raw gradient of ONE backward pass: it stands in for p.grad only while p.grad is still that
            # gradient -- a fresh tensor nobody has written to since autograd set it (clip_grad_norm_, loss-scale unscaling or
            # accumulation into an existing .grad bump its version counter; parallel.take_early already dropped a bucket that
            # saw two backward passes)
            if (k >= 0 and names[k:] == early["names"] and [p.numel() for p in live[k:]] == early["sizes"]
                    and all(p.grad._version == 0 for p in live[k:])):
                split = sum(p.numel() for p in live[:k])
        if split is None:
            if early is not None:
                early["work"].wait()          # a bucket that does not line up with this optimiser's parameters: drop it
            ev = None
            if parallel.TIMING is not None and world > 1:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            ops.flatten_into([p.grad for p in live], flat)
            if world > 1:
                dist.all_reduce(flat if flat_all is None else flat_all)      # one RCCL sum all-reduce over xGMI (+ the taint word)
            if ev is not None:
                ev[1].record()
                parallel.TIMING.append(ev)
            return
        k = len(names) - len(early["names"])
        if k:
            ops.flatten_into([p.grad for p in live[:k]], flat[:split])
        ev = None
        if parallel.TIMING is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        if flat_all is not None:
            dist.all_reduce(flat_all[:GRAD_HDR + split])       # the encoder bucket, with the header (taint word) in front of it
        elif k:
            dist.all_reduce(flat[:split])
        early["work"].wait()                  # the launching stream now waits for the early bucket's collective
        if ev is not None:
            ev[1].record()
            parallel.TIMING.append(ev)
        flat[split:].copy_(early["flat"])

    @torch.no_grad()
    def step(self, closure=None):
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        taint = None           # ONE taint word per step: h2_taint advances its mark, so later groups reuse the first group's (summed) word
        for gi, group in enumerate(self.param_groups):
            live = [p for p in group["params"] if p.grad is not None]      # torch SGD skips grad=None
            if not live:
                continue
            fl = self._flat.get(gi)
            if fl is None or fl["ids"] != [id(p) for p in live] or any(
                    p.data.data_ptr() < fl["p"].data_ptr() or
                    p.data.data_ptr() >= fl["p"].data_ptr() + fl["p"].numel() * 4 for p in live):
                self._build(gi, live)
                fl = self._flat[gi]
            if taint is None:
                ops.h2_taint(fl["g_all"][:1])      # clamps of this step's split-fp16 launches -> the word in front of the gradients
            else:
                fl["g_all"][:1].zero_()
            self._reduce(live, fl["g"], world, fl["g_all"])
            if taint is None:
                taint = fl["g_all"][:1]
            else:
                fl["g_all"][:1].copy_(taint)       # (already summed over the ranks)
            # buf starts at zero, so mu*buf + g reproduces torch's first-step "buf = g" exactly; a tainted step is skipped
            ops.sgd_momentum(fl["p"], fl["g"], fl["buf"], float(group["lr"]), float(group["momentum"]), 1.0 / world,
                             False, skip=fl["g_all"][:1])
        return None


class DataParallelAdam(Adam):
    """torch Adam (the reference's other optimiser choice, optim_scheduler.py:8) whose step first averages the
    gradients over the data-parallel ranks with the same single flat all-reduce FusedSGD uses -- without it the
    ranks' parameters would silently diverge."""

    @torch.no_grad()
    def step(self, closure=None):
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        if world > 1:
            early = parallel.take_early()          # this optimiser reduces everything itself: retire the early bucket
            if early is not None:
                early["work"].wait()
            live = [p for g in self.param_groups for p in g["params"] if p.grad is not None]
            if live:
                flat = torch.empty(sum(p.numel() for p in live), device=live[0].device, dtype=torch.float32)
                reduce_flat_grads([p.grad for p in live], flat)
                flat.mul_(1.0 / world)
                off = 0
                for p in live:
                    p.grad.copy_(flat[off:off + p.numel()].view_as(p.grad))
                    off += p.numel()
        return super().step(closure)


def get_optimizer(cfg, model_params):
    optim_name = cfg.SOLVER.optim
    if optim_name == 'adam':
        return DataParallelAdam(model_params, lr=cfg.SOLVER.lr)
    elif optim_name == 'sgd':
        return FusedSGD(model_params, lr=cfg.SOLVER.lr, momentum=0.9)


def get_lr_scheduler(cfg, optim=None):
    sche_name = cfg.SOLVER.scheduler
    if sche_name == 'steplr':
        return StepLR(optim, 50, gamma=0.1)
    elif sche_name == 'MultiStep':
        return MultiStepLR(optim, cfg.SOLVER.lr_step, gamma=0.1)
