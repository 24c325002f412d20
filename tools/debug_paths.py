import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import rel
from test_model_gpu import batch_t, hashed_model, make_cfg
from electrocardio_panorama_amd import engine, ops
from electrocardio_panorama_amd.network import build_loss
from oracle import hashweights as hw

B, V, L, seed = 2, 3, 512, 6
masks = {k: v.cuda() for k, v in hw.hashed_masks(V, B, L // 4).items()}
b = batch_t(B, V, L, seed)
cfg = make_cfg(V)
# path A: module + autograd
m = hashed_model(V).train(); m.dropout_masks = masks
random.seed(seed)
outsA = m(b["data"], b["input_theta"], b["target_theta"], b["rois"], phase="train")
lossesA = build_loss(cfg)(outsA[0], outsA[1], outsA[2], b["target_view"].unsqueeze(1), cfg)
lossesA[0].backward()
gA = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
# path B: engine directly
m2 = hashed_model(V).train()
P = {k: v.detach() for k, v in m2.named_parameters()}
random.seed(seed); choice = (random.randint(0, V - 1), random.randint(0, V - 1))
print("choice", choice)
outsB, sv = engine.forward(P, dict(m2.named_buffers()), b["data"], b["input_theta"], b["target_theta"], b["rois"],
                           phase="train", training=True, drop=engine.DropCfg(True, 0.2, masks), lead_choice=choice, save=True)
print("outs A vs B", [rel(a, c) for a, c in zip(outsA, outsB)])
tgt = b["target_view"].unsqueeze(1).contiguous()
g3 = ops.loss_bwd(outsB[0].contiguous(), outsB[1].contiguous(), outsB[2].contiguous(), tgt, torch.ones(4, device="cuda"),
                  (0.5, 0.5, 1.0), False, 7)
gB = engine.backward(P, sv, g3)
for k in gB:
    r = rel(gA[k], gB[k])
    if r > 1e-7:
        print(f"{k:40s} A-vs-B {r:.2e}")
print("done")
