"""Is a gradient deviation a ReLU tie?  For one train-phase case (dropout off) print, per decoder BatchNorm+ReLU, how many
activation signs differ between the HIP path and the fp64 oracle, and how close to zero those pre-activations are.
usage: debug_tie.py B V L seed"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_model_gpu import batch_t, hashed_model
from electrocardio_panorama_amd import engine
from oracle import hashweights as hw, nefnet_oracle as orc

B, V, L, seed = (int(a) for a in sys.argv[1:5])
m = hashed_model(V).train()
b = batch_t(B, V, L, seed)
bc = batch_t(B, V, L, seed, dev="cpu")
random.seed(seed)
choice = (random.randint(0, V - 1), random.randint(0, V - 1))
P = {k: v.detach() for k, v in m.named_parameters()}
with torch.no_grad():
    outs, sv = engine.forward(P, dict(m.named_buffers()), b["data"], b["input_theta"], b["target_theta"], b["rois"],
                              phase="train", training=True, drop=engine.DropCfg(False), lead_choice=choice, save=True)
saved = sv["dec"][0]


def oracle(dt):
    Pc = {k: v.to(dt) for k, v in hw.hashed_params(V).items()}
    Bfc = {k: (v.to(dt) if v.dtype.is_floating_point else v) for k, v in hw.hashed_buffers().items()}
    taps = {}
    orc.forward(Pc, Bfc, bc["data"].to(dt), bc["input_theta"].to(dt), bc["target_theta"].to(dt), bc["rois"], phase="gen",
                training=True, p=0.0, taps=taps)
    z1, z2r = taps["z1"], orc.roi_unpool(taps["z2_seg"], bc["rois"])
    z1m, z2m = orc.lead_mean(z1, V), orc.lead_mean(z2r, V)
    q = torch.nn.functional.linear(orc.angular_encoding(bc["target_theta"].to(dt)), Pc["mlp2.weight"], Pc["mlp2.bias"])
    c1, c2 = choice
    Ds = [torch.cat([z1m, z2m], 1), torch.cat([z1[:, 128 * c1:128 * (c1 + 1)], z2m], 1),
          torch.cat([z1m, z2r[:, 128 * c2:128 * (c2 + 1)]], 1)]
    pre = [[], [], [], []]
    for D in Ds:                      # pre-activations (BatchNorm outputs before ReLU), pass by pass
        x = q[:, :, None] * D
        for li, (blk, cv, bn) in enumerate((("decoder.1", "0", "1"), ("decoder.1", "3", "4"), ("decoder.3", "0", "1"),
                                            ("decoder.3", "3", "4"))):
            if li in (0, 2):
                x = torch.nn.functional.interpolate(x, scale_factor=2, mode="linear", align_corners=False)
            c = torch.nn.functional.conv1d(x, Pc[f"{blk}.double_conv.{cv}.weight"], Pc[f"{blk}.double_conv.{cv}.bias"], padding=1)
            mu, var = c.mean(dim=(0, 2), keepdim=True), c.var(dim=(0, 2), unbiased=False, keepdim=True)
            y = (c - mu) / torch.sqrt(var + 1e-5) * Pc[f"{blk}.double_conv.{bn}.weight"][None, :, None] + \
                Pc[f"{blk}.double_conv.{bn}.bias"][None, :, None]
            pre[li].append(y)
            x = torch.relu(y)
    return [torch.cat(p_, 0) for p_ in pre]


p64 = oracle(torch.float64)
for li, (x_in, c, mean, invstd, a, bb, pro, up_after) in enumerate(saved):
    N, C, T = c.shape
    Bp = N // 3
    y = c.view(3, Bp, C, T) * a.view(3, 1, C, 1) + bb.view(3, 1, C, 1)
    y = y.view(N, C, T).double().cpu()
    mism = (y > 0) != (p64[li] > 0)
    n = int(mism.sum())
    worst = float(p64[li][mism].abs().max()) if n else 0.0
    print(f"BN+ReLU {li + 1}: {n} sign mismatches of {y.numel()}; largest |fp64 pre-activation| among them {worst:.3e}; "
          f"pre-activation rel err {float((y - p64[li]).norm() / p64[li].norm()):.2e}, scale {float(p64[li].abs().mean()):.3f}")
