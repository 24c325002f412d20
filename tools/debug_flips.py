"""Count ReLU-mask disagreements between the HIP forward and the fp64 / fp32 oracle forward, layer by layer."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn.functional as F
from util import rel
from test_model_gpu import batch_t, hashed_model
from electrocardio_panorama_amd import engine
from oracle import hashweights as hw, nefnet_oracle as orc

B, V, L, seed = 2, 3, 512, 6
m = hashed_model(V).train()
masks = hw.hashed_masks(V, B, L // 4)
m.dropout_masks = {k: v.cuda() for k, v in masks.items()}
b = batch_t(B, V, L, seed)
P = {k: v.detach() for k, v in m.named_parameters()}
outs, sv = engine.forward(P, dict(m.named_buffers()), b["data"], b["input_theta"], b["target_theta"], b["rois"],
                          phase="train", training=True, drop=engine.DropCfg(True, 0.2, m.dropout_masks),
                          lead_choice=(2, 0), save=True)
bc = batch_t(B, V, L, seed, dev="cpu")


def oracle_acts(dt):
    Pc = {k: v.to(dt) for k, v in hw.hashed_params(V).items()}
    acts = {}
    x = bc["data"].to(dt)
    w = orc.stem(x, Pc, V)
    acts["stem"] = w
    for i in range(3):
        pre = f"W_encoder.layer1.{i}"
        h = F.relu(F.conv1d(w, Pc[pre + ".conv1.weight"], None, 1, 3, 1, V))
        acts[pre + ".h_prerelu_mask"] = h
        hd = h * masks[pre].to(dt) / 0.8
        y = F.relu(F.conv1d(hd, Pc[pre + ".conv2.weight"], None, 1, 3, 1, V) + w)
        acts[pre + ".y"] = y
        w = y
    return acts


a64, a32 = oracle_acts(torch.float64), oracle_acts(torch.float32)
hip = {"stem": sv["blk_enc"][0][0].t}
for i in range(3):
    pre = f"W_encoder.layer1.{i}"
    hip[pre + ".h_prerelu_mask"] = sv["blk_enc"][i][1]      # h after dropout: >0 iff relu active and kept
    hip[pre + ".y"] = sv["blk_enc"][i][2]
for k in a64:
    h_, r64, r32 = hip[k].detach().cpu(), a64[k], a32[k]
    if "mask" in k:
        keep = masks[k.split(".h_")[0]].bool()
        mm_h = int((((h_ > 0) != (r64 > 0)) & keep).sum()); mm_32 = int((((r32 > 0) != (r64 > 0)) & keep).sum())
        print(f"{k:45s} flips hip-vs-64 {mm_h:4d}  32-vs-64 {mm_32:4d}  of {h_.numel()}")
    else:
        mm_h = int(((h_ > 0) != (r64 > 0)).sum()); mm_32 = int(((r32 > 0) != (r64 > 0)).sum())
        print(f"{k:45s} flips hip-vs-64 {mm_h:4d}  32-vs-64 {mm_32:4d}  rel hip {rel(h_, r64):.2e}  rel 32 {rel(r32, r64):.2e}  "
              f"maxabs hip {float((h_.double()-r64).abs().max()):.2e} 32 {float((r32.double()-r64).abs().max()):.2e}")
print("---- decoder: post-ReLU masks, HIP vs oracle (end to end, same lead choice)")


def oracle_dec(dt):
    Pc = {k: v.to(dt) for k, v in hw.hashed_params(V).items()}
    Bfc = {k: (v.to(dt) if v.dtype.is_floating_point else v) for k, v in hw.hashed_buffers().items()}
    taps = {}
    orc.forward(Pc, Bfc, bc["data"].to(dt), bc["input_theta"].to(dt), bc["target_theta"].to(dt), bc["rois"],
                phase="gen", training=True, masks=masks, taps=taps)
    z1, z2b = taps["z1"], taps["z2_seg"]
    z2r = orc.roi_unpool(z2b, bc["rois"])
    z1m, z2m = orc.lead_mean(z1, V), orc.lead_mean(z2r, V)
    lat = torch.cat([z1m, z2m], 1)
    q = F.linear(orc.angular_encoding(bc["target_theta"].to(dt)), Pc["mlp2.weight"], Pc["mlp2.bias"])
    c1, c2 = 2, 0
    Ds = [lat, torch.cat([z1[:, 128 * c1:128 * (c1 + 1)], z2m], 1), torch.cat([z1m, z2r[:, 128 * c2:128 * (c2 + 1)]], 1)]
    acts = []
    for D in Ds:
        t = []
        orc.decoder(q[:, :, None] * D, Pc, Bfc, True, taps=t)
        acts.append(t)
    return [torch.cat([acts[p][li] for p in range(3)], 0) for li in range(4)], torch.cat([q[:, :, None] * D for D in Ds], 0)


d64, D64 = oracle_dec(torch.float64)
d32, D32 = oracle_dec(torch.float32)
saved, a4, out, passes = sv["dec"]
hip_acts = [saved[1][0], None, saved[3][0], a4]      # a1 = input of conv li=1 ; a3 = input of conv li=3
print("decoder input D: rel hip %.2e  32 %.2e" % (rel(saved[0][0], F.interpolate(D64, scale_factor=2, mode='linear', align_corners=False)),
                                                   rel(D32, D64)))
for li in (0, 2, 3):
    h_ = hip_acts[li].detach().cpu()
    print(f"a{li+1}: flips hip-vs-64 {int(((h_ > 0) != (d64[li] > 0)).sum()):4d}  32-vs-64 {int(((d32[li] > 0) != (d64[li] > 0)).sum()):4d}"
          f"  of {h_.numel()}  rel hip {rel(h_, d64[li]):.2e}  rel 32 {rel(d32[li], d64[li]):.2e}")
print("---- end-to-end gradients with lead_choice=(0,1)")
from electrocardio_panorama_amd import ops
tgt = b["target_view"].unsqueeze(1).contiguous()
g_pred, g_p, g_l = ops.loss_bwd(outs[0].contiguous(), outs[1].contiguous(), outs[2].contiguous(), tgt,
                                torch.ones(4, device="cuda"), (0.5, 0.5, 1.0), False, 7)
grads = engine.backward(P, sv, (g_pred, g_p, g_l))


def oracle_grads(dt):
    Pc = orc.require_grad({k: v.to(dt) for k, v in hw.hashed_params(V).items()})
    Bfc = {k: (v.to(dt) if v.dtype.is_floating_point else v) for k, v in hw.hashed_buffers().items()}
    r = orc.forward(Pc, Bfc, bc["data"].to(dt), bc["input_theta"].to(dt), bc["target_theta"].to(dt), bc["rois"],
                    phase="train", training=True, masks=masks, lead_choice=(2, 0))
    for t in r:
        t.retain_grad()
    orc.loss_v1(r[0], r[1], r[2], bc["target_view"].unsqueeze(1).to(dt))[0].backward()
    return Pc, r


P64, r64 = oracle_grads(torch.float64)
P32, r32 = oracle_grads(torch.float32)
print("loss grads wrt outputs: hip-vs-64", [f"{rel(a, c.grad):.1e}" for a, c in zip((g_pred, g_p, g_l), r64)])
for k in ["decoder.4.bias", "decoder.4.weight", "decoder.3.double_conv.4.bias", "decoder.3.double_conv.4.weight",
          "decoder.3.double_conv.3.weight", "decoder.3.double_conv.1.bias", "decoder.1.double_conv.4.bias", "mlp2.bias",
          "z1_conv.0.conv2.weight", "W_encoder.conv1.weight"]:
    print(f"{k:36s} hip-vs-64 {rel(grads[k], P64[k].grad):.2e}   32-vs-64 {rel(P32[k].grad, P64[k].grad):.2e}")
