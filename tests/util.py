import numpy as np
import torch


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu().reshape(-1)
    b = torch.as_tensor(b).detach().double().cpu().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


def maxabs(a, b):
    a = torch.as_tensor(a).detach().double().cpu().reshape(-1)
    b = torch.as_tensor(b).detach().double().cpu().reshape(-1)
    return float((a - b).abs().max())


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def sub(t, n=2048):
    f = torch.as_tensor(t).detach().cpu().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].numpy().copy()


def stats(t):
    t = torch.as_tensor(t).detach().double().cpu()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])


FWD_TOL = 1e-5      # rel-L2, forward activations vs the fp32 CPU oracle (BASELINE.json asks 1e-4)
GRAD_TOL = 1e-4     # rel-L2, gradients
