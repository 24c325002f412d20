"""Diagnostic: per-tensor gradient error of the HIP path vs the fp64 oracle, next to the fp32 oracle's own error."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import rel
from test_model_gpu import _train_once, batch_t
from oracle import hashweights as hw, nefnet_oracle as orc

B, V, L, seed = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (2, 3, 512, 6)
MASKED = int(sys.argv[5]) if len(sys.argv) > 5 else 1        # 0: dropout off (p = 0)
m, outs, losses = _train_once(V, B, L, seed, "l1_loss", bool(MASKED))
b = batch_t(B, V, L, seed, dev="cpu")
masks = hw.hashed_masks(V, B, L // 4) if MASKED else None


def run(dt):
    P = orc.require_grad({k: v.to(dt) for k, v in hw.hashed_params(V).items()})
    Bf = {k: (v.to(dt) if v.dtype.is_floating_point else v) for k, v in hw.hashed_buffers().items()}
    random.seed(seed)
    r = orc.forward(P, Bf, b["data"].to(dt), b["input_theta"].to(dt), b["target_theta"].to(dt), b["rois"],
                    phase="train", training=True, masks=masks, p=0.2 if MASKED else 0.0)
    orc.loss_v1(r[0], r[1], r[2], b["target_view"].unsqueeze(1).to(dt))[0].backward()
    return r, P


r32, P32 = run(torch.float32)
r64, P64 = run(torch.float64)
print("outputs: hip-vs-64 %.2e   32-vs-64 %.2e" % (max(rel(a, c) for a, c in zip(outs, r64)), max(rel(a, c) for a, c in zip(r32, r64))))
named = dict(m.named_parameters())
live = [k for k in P32 if k not in orc.DEAD_PARAMS]
f = lambda Pd: torch.cat([Pd[k].grad.reshape(-1).double() for k in live])
fh = torch.cat([named[k].grad.reshape(-1).double().cpu() for k in live])
print("flat grad: hip-vs-64 %.2e  32-vs-64 %.2e  hip-vs-32 %.2e" % (rel(fh, f(P64)), rel(f(P32), f(P64)), rel(fh, f(P32))))
for k in live:
    print("%-40s hip-vs-64 %.2e   32-vs-64 %.2e   |g| %.2e" % (k, rel(named[k].grad, P64[k].grad), rel(P32[k].grad, P64[k].grad), float(P64[k].grad.norm())))
tgt = b["target_view"].unsqueeze(1)
for name, (i, j) in {"out-vs-p": (0, 1), "out-vs-l": (0, 2)}.items():
    sh = torch.sign(outs[i].detach().cpu() - outs[j].detach().cpu())
    s32 = torch.sign(r32[i].detach() - r32[j].detach())
    s64 = torch.sign(r64[i].detach() - r64[j].detach())
    d32 = (r32[i] - r32[j]).detach().abs()
    print(name, "sign mismatches hip-vs-64:", int((sh != s64).sum()), " 32-vs-64:", int((s32 != s64).sum()),
          " exact ties hip:", int((sh == 0).sum()), " min|d| (32):", float(d32.min()), " #|d|<1e-6:", int((d32 < 1e-6).sum()))
sh = torch.sign(outs[0].detach().cpu() - tgt.cpu()); s64 = torch.sign(r64[0].detach() - tgt.cpu().double())
print("out-vs-target sign mismatches hip-vs-64:", int((sh != s64).sum()))
