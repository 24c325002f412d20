import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn.functional as F
from util import rel, maxabs
from test_model_gpu import batch_t, hashed_model
from electrocardio_panorama_amd import engine, ops
from oracle import hashweights as hw

B, V, L, seed = 2, 3, 512, 6
m = hashed_model(V).train()
m.dropout_masks = {k: v.cuda() for k, v in hw.hashed_masks(V, B, L // 4).items()}
b = batch_t(B, V, L, seed)
P = {k: v.detach() for k, v in m.named_parameters()}
Bf = dict(m.named_buffers())
random.seed(seed)
outs, sv = engine.forward(P, Bf, b["data"], b["input_theta"], b["target_theta"], b["rois"], phase="train", training=True,
                          drop=engine.DropCfg(True, 0.2, m.dropout_masks), lead_choice=(0, 1), save=True)
out3 = sv["dec"][2]
tgt = b["target_view"].unsqueeze(1)
g_pred, g_p, g_l = ops.loss_bwd(outs[0].contiguous(), outs[1].contiguous(), outs[2].contiguous(), tgt.contiguous(),
                                torch.ones(4, device="cuda"), (0.5, 0.5, 1.0), False, 7)
g_out = torch.cat([g_pred, g_p, g_l], 0)
saved, a4, out, passes = sv["dec"]
d = lambda t: t.detach().double().cpu()
# (a) outconv bwd data
w4 = P["decoder.4.weight"]
ga4 = ops.outconv_bwd_data(g_out, out, w4, 64)
a4r = d(a4).requires_grad_(True)
o_ref = torch.sigmoid(F.conv1d(a4r, d(w4), d(P["decoder.4.bias"]), 1, 1) / 3)
print("outconv fwd on real a4:", rel(out, o_ref))
o_ref.backward(d(g_out))
print("outconv_bwd_data:", rel(ga4, a4r.grad), " sum err", float((d(ga4).sum() - a4r.grad.sum()).abs()), "sum", float(a4r.grad.sum()))
# (b) bn_relu_bwd layer 3 on real tensors
x, c, mean, invstd, a, bb = saved[3]
pre = "decoder.3.double_conv.4"
gc, gg, gbeta = ops.bn_relu_bwd(ga4, c, P[pre + ".weight"], mean, invstd, a, bb, 3)
cr = d(c).requires_grad_(True)
gam, bet = d(P[pre + ".weight"]).requires_grad_(True), d(P[pre + ".bias"]).requires_grad_(True)
ys = [F.relu(F.batch_norm(cr[p * B:(p + 1) * B], None, None, gam, bet, True, 0.1, 1e-5)) for p in range(3)]
yref = torch.cat(ys, 0)
print("affine_relu fwd:", rel(a4, yref), "mask mismatches:", int(((d(a4) > 0) != (yref > 0)).sum()))
yref.backward(d(ga4))
print("bn bwd: gx", rel(gc, cr.grad), "ggamma", rel(gg, gam.grad), "gbeta", rel(gbeta, bet.grad))
print("gbeta abs err max", maxabs(gbeta, bet.grad), " |gbeta|", float(bet.grad.norm()))
g_masked = d(ga4) * (yref > 0)
print("sum|g| / |sum g| per channel (median):", float((g_masked.abs().sum((0, 2)) / g_masked.sum((0, 2)).abs()).median()))
print("---- engine.backward on the same saved state")
grads = engine.backward(P, sv, (g_pred, g_p, g_l))
k = pre + ".bias"
print("engine gbeta vs piecewise gbeta:", rel(grads[k], gbeta), " vs fp64-on-hip-inputs:", rel(grads[k], bet.grad))
print("engine ggamma vs piecewise:", rel(grads[pre + ".weight"], gg))
grads2 = {}
gD = engine.decoder_bwd(sv["dec"], g_out, P, grads2)
print("decoder_bwd alone gbeta vs piecewise:", rel(grads2[k], gbeta))
