"""Optimiser / scheduler factory of reference codes/solver/optim_scheduler.py:5-18.

'sgd' returns FusedSGD: torch.optim.SGD(lr, momentum=0.9) semantics, but one HIP launch over a flat
parameter buffer, and -- when torch.distributed is initialised -- one RCCL all-reduce of the flat
gradient buffer per step (data-parallel training, one process per GPU)."""
import torch
import torch.distributed as dist
from torch.optim import Adam
from torch.optim.lr_scheduler import StepLR, MultiStepLR

from .. import ops
from ..parallel import reduce_flat_grads


class FusedSGD(torch.optim.Optimizer):
    def __init__(self, params, lr, momentum=0.9):
        super().__init__(params, dict(lr=lr, momentum=momentum))
        self._flat = {}      # group index -> dict(params, p, g, buf)

    def _build(self, gi, live):
        n = sum(p.numel() for p in live)
        dev = live[0].device
        flat_p = torch.empty(n, device=dev, dtype=torch.float32)
        flat_b = torch.zeros(n, device=dev, dtype=torch.float32)
        off = 0
        for p in live:
            k = p.numel()
            flat_p[off:off + k].copy_(p.data.reshape(-1))
            p.data = flat_p[off:off + k].view_as(p.data)          # parameters become views of the flat buffer
            st = self.state[p]
            if "momentum_buffer" in st and st["momentum_buffer"] is not None:
                flat_b[off:off + k].copy_(st["momentum_buffer"].reshape(-1))
            st["momentum_buffer"] = flat_b[off:off + k].view_as(p.data)
            off += k
        self._flat[gi] = dict(ids=[id(p) for p in live], params=live, p=flat_p, buf=flat_b,
                              g=torch.empty(n, device=dev, dtype=torch.float32))

    def load_state_dict(self, state_dict):
        """The loaded momentum buffers replace the flat one: drop the flat views so the next step() re-imports them."""
        super().load_state_dict(state_dict)
        self._flat = {}

    @torch.no_grad()
    def step(self, closure=None):
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
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
            reduce_flat_grads([p.grad for p in live], fl["g"])             # one RCCL sum all-reduce over xGMI
            # buf starts at zero, so mu*buf + g reproduces torch's first-step "buf = g" exactly
            ops.sgd_momentum(fl["p"], fl["g"], fl["buf"], float(group["lr"]), float(group["momentum"]), 1.0 / world,
                             False)
        return None


class DataParallelAdam(Adam):
    """torch Adam (the reference's other optimiser choice, optim_scheduler.py:8) whose step first averages the
    gradients over the data-parallel ranks with the same single flat all-reduce FusedSGD uses -- without it the
    ranks' parameters would silently diverge."""

    @torch.no_grad()
    def step(self, closure=None):
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        if world > 1:
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
