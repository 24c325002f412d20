"""CPU oracle for the Nef-Net train-step hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain torch-CPU restatement of the reference's algorithm.  It is
the checker the HIP path is compared against; it is never part of the product
path.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
leg may import it.

Parity pin: the reference (WhatAShot/Electrocardio-Panorama) ships no golden
vectors and no value-asserting tests.  The oracle is
pinned instead against outputs of the reference itself, imported in the build
container by `oracle/make_golden.py`; those outputs are committed as fixtures
under `tests/golden/` and `tests/test_oracle_golden.py` re-checks the oracle
against them on every run.

The arithmetic of the reference lives in PyTorch (pinned torch==1.3.1 in the
reference's requirements.txt:2; the fixtures were produced with torch 2.10 CPU,
every resampling call in the reference passes align_corners explicitly).  Each
function cites the reference file:line it restates (paths relative to the
reference's `codes/` directory).

State is held in two flat dicts keyed by the reference's state_dict names:
`params` (53 tensors) and `buffers` (BatchNorm running statistics).
"""
import math
import random

import torch
import torch.nn.functional as F

N_SEG = 7          # heartbeat segments per sample (dataset/tianchi.py:103-106)
ROI_BINS = 16      # roi_algin size (network/model_nefnet.py:136)
SCALE = 0.25       # spatial_scale = 128 / 512 (network/model_nefnet.py:136,143)
DROP_P = 0.2       # every nn.Dropout on the path (model_nefnet.py:46, resnet_1d.py:37)
BN_EPS = 1e-5
BN_MOM = 0.1

DROPOUT_SITES = (
    "W_encoder.layer1.0", "W_encoder.layer1.1", "W_encoder.layer1.2",
    "w_conv.0", "z1_conv.0", "z2_conv1.0", "z2_conv2.0", "z2_conv2.2",
)


# --------------------------------------------------------------------------
# parameter inventory (network/model_nefnet.py:67-107, encoder/resnet_1d.py:97-120)
# --------------------------------------------------------------------------
def param_shapes(V):
    """Ordered {name: shape} of every parameter tensor in Model_nefnet(lead_num=V)."""
    C = 128 * V
    s = {}
    s["W_encoder.conv1.weight"] = (C, 1, 15)
    for i in range(3):
        s[f"W_encoder.layer1.{i}.conv1.weight"] = (C, 128, 7)
        s[f"W_encoder.layer1.{i}.conv2.weight"] = (C, 128, 7)
    s["mlp1.weight"] = (128, 12)
    s["mlp1.bias"] = (128,)
    s["mlp2.weight"] = (256, 12)
    s["mlp2.bias"] = (256,)
    s["w_feature_extractor.0.weight"] = (128, 128, 3)
    s["w_feature_extractor.0.bias"] = (128,)

    def block(prefix, cin, cout, groups):
        s[prefix + ".conv1.weight"] = (cout, cin // groups, 3)
        s[prefix + ".conv2.weight"] = (cout, cout // groups, 3)
        s[prefix + ".residual_conv.weight"] = (cout, cin // groups, 1)
        s[prefix + ".residual_conv.bias"] = (cout,)

    block("w_conv.0", C, C, V)
    block("z1_conv.0", 64 * V, C, V)
    block("z2_conv1.0", 64 * V, C, V)
    block("z2_conv2.0", 7 * C, 7 * C, 7 * V)
    s["z2_conv2.1.weight"] = (7 * C, 64, 2)      # ConvTranspose1d: [in, out/groups, k]
    s["z2_conv2.1.bias"] = (7 * C // 2,)
    block("z2_conv2.2", 7 * C // 2, 7 * C, 7 * V)
    for blk, (cin, cout) in (("decoder.1", (256, 128)), ("decoder.3", (128, 64))):
        s[blk + ".double_conv.0.weight"] = (cout, cin, 3)
        s[blk + ".double_conv.0.bias"] = (cout,)
        s[blk + ".double_conv.1.weight"] = (cout,)
        s[blk + ".double_conv.1.bias"] = (cout,)
        s[blk + ".double_conv.3.weight"] = (cout, cout, 3)
        s[blk + ".double_conv.3.bias"] = (cout,)
        s[blk + ".double_conv.4.weight"] = (cout,)
        s[blk + ".double_conv.4.bias"] = (cout,)
    s["decoder.4.weight"] = (1, 64, 3)
    s["decoder.4.bias"] = (1,)
    return s


def param_shapes2():
    """Model_nefnet2 (network/model_nefnet2.py:68-116): the single-lead inventory plus the two single convs."""
    s = param_shapes(1)
    for name in ("single_conv_z1.0", "single_conv_z2.0"):
        s[name + ".weight"] = (128, 128, 3)
        s[name + ".bias"] = (128,)
    return s


def buffer_shapes():
    s = {}
    for blk, c in (("decoder.1", 128), ("decoder.3", 64)):
        for bn in ("1", "4"):
            s[f"{blk}.double_conv.{bn}.running_mean"] = (c,)
            s[f"{blk}.double_conv.{bn}.running_var"] = (c,)
            s[f"{blk}.double_conv.{bn}.num_batches_tracked"] = ()
    return s


DEAD_PARAMS = (  # never receive a gradient
    "w_feature_extractor.0.weight", "w_feature_extractor.0.bias",
    "w_conv.0.residual_conv.weight", "w_conv.0.residual_conv.bias",
    "z2_conv2.0.residual_conv.weight", "z2_conv2.0.residual_conv.bias",
)


def fresh_buffers():
    b = {}
    for k, shp in buffer_shapes().items():
        if k.endswith("running_var"):
            b[k] = torch.ones(shp)
        elif k.endswith("num_batches_tracked"):
            b[k] = torch.zeros((), dtype=torch.int64)
        else:
            b[k] = torch.zeros(shp)
    return b


# --------------------------------------------------------------------------
# decision replay (test instrument for gradient parity)
# --------------------------------------------------------------------------
class Decisions:
    """The backward pass is discontinuous wherever the forward takes a DISCRETE decision: ReLU on/off
    (model_nefnet.py:39,59; resnet_1d.py:34,52; DoubleConv :20,23) and the sign of an L1 residual (losses.py:8,27).
    When a pre-activation lies within fp32 round-off of the switching point, two correct fp32 implementations may
    decide differently, and one such tie can move the whole flat gradient by 1e-3 at small shapes.  This object lets
    the oracle REPLAY the decisions another implementation took (`masks[site]`: bool tensor, True = ReLU passes; for
    sign sites a float tensor of -1/0/+1; or a callable(own_decision) -> decision for partial overrides) and records,
    per site, how many decisions the oracle itself would have taken differently and how far from the switching point
    those pre-activations were.  A parity test then asserts (a) every differing decision is a tie -- |pre-activation|
    at round-off level -- and (b) the gradients agree at the tie-free tolerance.  With `masks` empty it only records
    nothing and the oracle is the plain reference restatement."""

    def __init__(self, masks=None):
        self.masks = masks or {}
        self.report = {}          # site -> dict(flips, worst, rms, numel)

    def _note(self, site, own, taken, pre):
        flips = own != taken
        n = int(flips.sum())
        # `degenerate`: the oracle's argument is EXACTLY zero there (e.g. a conv over the all-zero tail of a beat); which
        # side of the switch an implementation lands on is then decided by rounding residue of its algorithm alone
        self.report[site] = dict(flips=n, degenerate=int((flips & (pre == 0)).sum()) if n else 0,
                                 worst=float(pre[flips].abs().max()) if n else 0.0,
                                 rms=float(pre.detach().double().pow(2).mean().sqrt()), numel=pre.numel())

    def relu(self, site, pre):
        m = self.masks.get(site)
        if m is None:
            return F.relu(pre)
        own = pre.detach() > 0
        if callable(m):
            m = m(own)
        m = m.to(pre.device)
        self._note(site, own, m, pre.detach())
        return pre * m.to(pre.dtype)

    def l1(self, site, a, b):
        """nn.L1Loss()(a, b) = mean(|a - b|); with a replayed sign tensor its derivative uses that sign."""
        sg = self.masks.get(site)
        if sg is None:
            return F.l1_loss(a, b)
        d = a - b
        own = torch.sign(d.detach())
        sg = sg.to(d.dtype)
        self._note(site, own, sg, d.detach())
        return (d * sg).mean()

    def total_flips(self):
        return sum(r["flips"] for r in self.report.values())


_PLAIN = Decisions()


# --------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------
def angular_encoding(theta):
    """network/utils/theta_encoder.py:13-29.  [..., 2] -> [..., 12]."""
    t, p = theta[..., 0:1], theta[..., 1:2]
    a = torch.cat([theta, t + p, t - p], dim=-1)                 # [..., 4]
    enc = torch.stack([a, torch.sin(a), torch.cos(a)], dim=-1)   # [..., 4, 3]
    return enc.reshape(*theta.shape[:-1], 12)


def _drop(x, site, training, masks, p):
    """nn.Dropout(0.2); `masks[site]` (0/1 keep mask) replaces the random draw."""
    if not training or p == 0.0:
        return x
    if masks is not None:
        return x * masks[site].to(x.dtype) / (1.0 - p)
    return F.dropout(x, p, True)


def res_block(x, P, prefix, groups, k, training, masks, p, dec=_PLAIN):
    """BasicBlock: encoder/resnet_1d.py:42-53 (k=7) and model_nefnet.py:48-60 (k=3)."""
    h = F.conv1d(x, P[prefix + ".conv1.weight"], None, 1, k // 2, 1, groups)
    h = dec.relu(prefix + ".relu1", h)
    h = _drop(h, prefix, training, masks, p)
    h = F.conv1d(h, P[prefix + ".conv2.weight"], None, 1, k // 2, 1, groups)
    res = x
    if k == 3 and h.shape[1] != x.shape[1]:       # model_nefnet.py:54
        res = F.conv1d(x, P[prefix + ".residual_conv.weight"],
                       P[prefix + ".residual_conv.bias"], 1, 0, 1, groups)
    return dec.relu(prefix + ".relu2", h + res)


def stem(x, P, V):
    """encoder/encoder.py:35-38 with resnet_1d.py:102-105."""
    y = F.conv1d(x, P["W_encoder.conv1.weight"], None, 2, 7, 1, V)
    return F.max_pool1d(F.relu(y), 3, 2, 1)


def roi_align_mid(z, rois, size=ROI_BINS, scale=SCALE):
    """network/utils/roi_pooling_1d.py:38-69 (`roi_algin`), restated literally:
    the grid's x component indexes the size-1 axis."""
    B, n_roi, T = rois.shape[0], rois.shape[1], z.shape[2]
    r = rois.detach().to(torch.float32).clone()
    r.mul_(scale)
    r.mul_(2 / T).add_(-1)
    rows = []
    for i in range(B):
        rows.append(torch.stack([torch.linspace(r[i, j, 0], r[i, j, 1], steps=size)
                                 for j in range(n_roi)], dim=0))
    gx = torch.stack(rows, dim=0)
    grid = torch.stack([gx, torch.zeros_like(gx)], dim=3).to(z.device)
    grid = grid.to(z.dtype)     # no-op in fp32 (the reference is fp32-only); lets tests run an fp64 yardstick
    return F.grid_sample(z.unsqueeze(-1), grid, align_corners=False)


def roi_segment_table(rois, scale=SCALE):
    """Integer bookkeeping of roi_pooling_1d.py:82-92: latent start and length per segment."""
    r = (rois.detach().to(torch.float32) * scale).long()
    return r[..., 0], r[..., 1] - r[..., 0]


def roi_unpool(z, rois, scale=SCALE):
    """network/utils/roi_pooling_1d.py:72-99 (`roi_pooling_reverse`)."""
    _, seg_len = roi_segment_table(rois, scale)
    out = []
    for i in range(z.shape[0]):
        parts = []
        for j in range(rois.shape[1]):
            n = int(seg_len[i, j])
            if n != 0:
                parts.append(F.interpolate(z[i:i + 1, :, j, :], n, mode="linear", align_corners=False))
            else:
                parts.append(torch.empty(0))
        out.append(torch.cat(parts, dim=-1))
    return torch.cat(out, dim=0)


def batch_norm(x, P, Bf, prefix, training):
    """nn.BatchNorm1d inside DoubleConv (model_nefnet.py:19,22); updates Bf in place when training."""
    if training:
        Bf[prefix + ".num_batches_tracked"] += 1
    return F.batch_norm(x, Bf[prefix + ".running_mean"], Bf[prefix + ".running_var"],
                        P[prefix + ".weight"], P[prefix + ".bias"], training, BN_MOM, BN_EPS)


def decoder(x, P, Bf, training, taps=None, dec=_PLAIN, pass_id=0):
    """model_nefnet.py:101-107 followed by sigmoid(x / 3) (:168).  `taps` (a list) collects the four post-ReLU
    activations for diagnostics.  Decision sites: "pass<pass_id>.<blk>.double_conv.<bn>"."""
    for blk in ("decoder.1", "decoder.3"):
        x = F.interpolate(x, scale_factor=2, mode="linear", align_corners=False)
        for conv, bn in (("0", "1"), ("3", "4")):
            x = F.conv1d(x, P[f"{blk}.double_conv.{conv}.weight"], P[f"{blk}.double_conv.{conv}.bias"], 1, 1)
            x = dec.relu(f"pass{pass_id}.{blk}.double_conv.{bn}", batch_norm(x, P, Bf, f"{blk}.double_conv.{bn}", training))
            if taps is not None:
                taps.append(x)
    x = F.conv1d(x, P["decoder.4.weight"], P["decoder.4.bias"], 1, 1)
    return torch.sigmoid(x / 3)


def lead_mean(z, V):
    return torch.mean(torch.stack(torch.chunk(z, V, dim=1), dim=0), dim=0)


# --------------------------------------------------------------------------
# Model_nefnet.forward / gen_ecg
# --------------------------------------------------------------------------
def forward(P, Bf, x, input_thetas, query_theta, rois, rest_theta=None, phase="train",
            training=True, masks=None, p=DROP_P, lead_choice=None, taps=None, dec=_PLAIN):
    """network/model_nefnet.py:109-194.

    `lead_choice=(c1, c2)` overrides the two `random.randint` draws (:154,:156); when None
    they are drawn from Python's `random` in the reference's order.  `taps`, if a dict, is
    filled with named intermediates."""
    V = x.shape[1]
    w = stem(x, P, V)                                                       # :117
    for i in range(3):
        w = res_block(w, P, f"W_encoder.layer1.{i}", V, 7, training, masks, p, dec)
    enc_in = F.linear(angular_encoding(input_thetas), P["mlp1.weight"], P["mlp1.bias"])   # :118,:121
    B, C, T = w.shape
    ew = w * enc_in.reshape(B, C, 1)                                        # :122-123
    ew = res_block(ew, P, "w_conv.0", V, 3, training, masks, p, dec)             # :124
    ew = ew.reshape(B, V, 2, 64, T)                                         # :127-131
    z1 = ew[:, :, 0].reshape(B, 64 * V, T)
    z2 = ew[:, :, 1].reshape(B, 64 * V, T)
    z1 = res_block(z1, P, "z1_conv.0", V, 3, training, masks, p, dec)            # :133
    z2 = res_block(z2, P, "z2_conv1.0", V, 3, training, masks, p, dec)           # :134
    z2a = roi_align_mid(z2, rois)                                           # :136
    h = z2a.contiguous().view(B, C * N_SEG, ROI_BINS)                       # :137
    h = res_block(h, P, "z2_conv2.0", N_SEG * V, 3, training, masks, p, dec)
    h = F.conv_transpose1d(h, P["z2_conv2.1.weight"], P["z2_conv2.1.bias"], 2, 0, 0, N_SEG * V)
    h = res_block(h, P, "z2_conv2.2", N_SEG * V, 3, training, masks, p, dec)
    z2b = h.view(B, C, N_SEG, 2 * ROI_BINS)                                 # :138
    if taps is not None:
        taps.update(w=w, z1=z1, z2_conv1=z2, z2_roi=z2a, z2_seg=z2b)
    if phase == "gen":
        return z1, z2b
    z2r = roi_unpool(z2b, rois)                                             # :143
    z1m, z2m = lead_mean(z1, V), lead_mean(z2r, V)                          # :146-149
    latent = torch.cat([z1m, z2m], dim=1)
    if lead_choice is None:
        c1 = random.randint(0, V - 1)                                       # :154
        c2 = random.randint(0, V - 1)                                       # :156
    else:
        c1, c2 = lead_choice
    shuf_p = torch.cat([z1[:, 128 * c1:128 * (c1 + 1)], z2m], dim=1)        # :159
    shuf_l = torch.cat([z1m, z2r[:, 128 * c2:128 * (c2 + 1)]], dim=1)       # :160
    q = F.linear(angular_encoding(query_theta).reshape(B, -1), P["mlp2.weight"], P["mlp2.bias"])
    out = decoder(q[:, :, None] * latent, P, Bf, training, dec=dec, pass_id=0)                  # :166-168
    out_p = decoder(q[:, :, None] * shuf_p, P, Bf, training, dec=dec, pass_id=1)                # :170-172
    out_l = decoder(q[:, :, None] * shuf_l, P, Bf, training, dec=dec, pass_id=2)                # :174-176
    if taps is not None:
        taps.update(z2_rev=z2r, latent_all=latent, q=q)
    if phase == "train":
        return out, out_p, out_l
    if phase in ("val", "test"):
        rq = F.linear(angular_encoding(rest_theta), P["mlp2.weight"], P["mlp2.bias"])
        rest = [decoder(rq[:, i, :, None] * latent, P, Bf, training) for i in range(rq.shape[1])]
        return out, out_p, out_l, torch.cat(rest, dim=1)
    raise KeyError("please type correct phase")


class _LeadSites:
    """Decision sites of the shared single-lead encoder, one set per lead: "<site>@<lead>"."""

    def __init__(self, dec, lead):
        self.dec, self.lead = dec, lead

    def relu(self, site, pre):
        return self.dec.relu(f"{site}@{self.lead}", pre)


def forward2(P, Bf, x, input_thetas, query_theta, rois, rest_theta=None, phase="train", training=True, masks=None,
             p=DROP_P, lead_choice=None, dec=_PLAIN):
    """network/model_nefnet2.py:118-194: the lead loop with ONE shared single-lead encoder, restated as written."""
    B, V, _ = x.shape
    z1_list, z2_list = [], []
    for i in range(V):                                                       # :126
        ld = _LeadSites(dec, i)
        xs, ths = x[:, i:i + 1], input_thetas[:, i:i + 1]
        w = stem(xs, P, 1)                                                   # :130
        for k in range(3):
            w = res_block(w, P, f"W_encoder.layer1.{k}", 1, 7, training, masks, p, ld)
        e = F.linear(angular_encoding(ths), P["mlp1.weight"], P["mlp1.bias"])   # :131-133
        w = e[:, 0][:, :, None] * w                                          # :134
        w = res_block(w, P, "w_conv.0", 1, 3, training, masks, p, ld)            # :135
        z1, z2 = torch.chunk(w, 2, dim=1)                                    # :137
        z1 = res_block(z1, P, "z1_conv.0", 1, 3, training, masks, p, ld)         # :139
        z1 = F.conv1d(z1, P["single_conv_z1.0.weight"], P["single_conv_z1.0.bias"], padding=1)   # :140
        z2 = res_block(z2, P, "z2_conv1.0", 1, 3, training, masks, p, ld)        # :141
        z2 = roi_align_mid(z2, rois)                                         # :143
        h = z2.contiguous().view(B, 128 * N_SEG, ROI_BINS)                   # :144
        h = res_block(h, P, "z2_conv2.0", N_SEG, 3, training, masks, p, ld)      # :145
        h = F.conv_transpose1d(h, P["z2_conv2.1.weight"], P["z2_conv2.1.bias"], 2, 0, 0, N_SEG)
        h = res_block(h, P, "z2_conv2.2", N_SEG, 3, training, masks, p, ld)
        z2 = roi_unpool(h.view(B, 128, N_SEG, 2 * ROI_BINS), rois)           # :147
        z2 = F.conv1d(z2, P["single_conv_z2.0.weight"], P["single_conv_z2.0.bias"], padding=1)   # :148
        z1_list.append(z1)
        z2_list.append(z2)
    z1m = torch.mean(torch.stack(z1_list, dim=0), dim=0)                     # :154-155
    z2m = torch.mean(torch.stack(z2_list, dim=0), dim=0)
    latent = torch.cat([z1m, z2m], dim=1)
    if phase == "gen":                                                       # :158-159
        return z1m, z2m
    if lead_choice is None:
        c1 = random.randint(0, V - 1)                                        # :162
        c2 = random.randint(0, V - 1)                                        # :164
    else:
        c1, c2 = lead_choice
    shuf_p = torch.cat([z1_list[c1], z2m], dim=1)                            # :167
    shuf_l = torch.cat([z1m, z2_list[c2]], dim=1)                            # :168
    q = F.linear(angular_encoding(query_theta).reshape(B, -1), P["mlp2.weight"], P["mlp2.bias"])
    out = decoder(q[:, :, None] * latent, P, Bf, training, dec=dec, pass_id=0)                   # :174-176
    out_p = decoder(q[:, :, None] * shuf_p, P, Bf, training, dec=dec, pass_id=1)
    out_l = decoder(q[:, :, None] * shuf_l, P, Bf, training, dec=dec, pass_id=2)
    if phase == "train":
        return out, out_p, out_l
    if phase in ("val", "test"):
        rq = F.linear(angular_encoding(rest_theta), P["mlp2.weight"], P["mlp2.bias"])
        rest = [decoder(rq[:, i, :, None] * latent, P, Bf, training) for i in range(rq.shape[1])]
        return out, out_p, out_l, torch.cat(rest, dim=1)
    raise KeyError("please type correct phase")


def gen_ecg(P, Bf, z1, z2, query_theta, rois):
    """network/model_nefnet.py:196-218 (always eval mode)."""
    V = z1.shape[1] // 128
    z2r = roi_unpool(z2, rois)
    latent = torch.cat([lead_mean(z1, V), lead_mean(z2r, V)], dim=1)
    q = F.linear(angular_encoding(query_theta), P["mlp2.weight"], P["mlp2.bias"])
    outs = [decoder(q[:, i, :, None] * latent, P, Bf, False) for i in range(q.shape[1])]
    return torch.cat(outs, dim=1)


# --------------------------------------------------------------------------
# loss (network/loss/losses.py:5-50) and the solver step (solver/solver.py:171-189,232-235)
# --------------------------------------------------------------------------
def loss_v1(pred, pred_p, pred_l, target, loss_factor=(0.5, 0.5, 1.0), loss_using=(1, 2, 3),
            reg_loss="l1_loss", rest_out=None, rest_view=None, dec=_PLAIN):
    reg = F.mse_loss if reg_loss == "l2_loss" else F.l1_loss
    if reg_loss not in ("l1_loss", "l2_loss"):
        raise NotImplementedError
    # nn.L1Loss == mean(|a - b|); `dec` may replay the residual signs (see Decisions)
    l1 = dec.l1("loss1", pred.detach(), pred_p) if 1 in loss_using else 0.0
    l2 = dec.l1("loss2", pred.detach(), pred_l) if 2 in loss_using else 0.0
    if 3 in loss_using:
        l3 = dec.l1("loss3", pred, target) if reg_loss == "l1_loss" else reg(pred, target)
    else:
        l3 = 0.0
    f = loss_factor
    total = l1 * f[0] + l2 * f[1] + l3 * f[2]
    if rest_out is not None and rest_view is not None:
        return total, l1 * f[0], l2 * f[1], l3 * f[2], reg(rest_out, rest_view)
    return total, l1 * f[0], l2 * f[1], l3 * f[2]


class SGDState:
    """torch.optim.SGD(lr, momentum=0.9) over a params dict (solver/optim_scheduler.py:10)."""

    def __init__(self, lr, momentum=0.9):
        self.lr, self.momentum, self.buf = lr, momentum, {}

    def step(self, P):
        with torch.no_grad():
            for k, v in P.items():
                if v.grad is None:
                    continue
                if k not in self.buf:
                    self.buf[k] = v.grad.clone()
                else:
                    self.buf[k].mul_(self.momentum).add_(v.grad)
                v.add_(self.buf[k], alpha=-self.lr)
                v.grad = None


def train_step(P, Bf, opt, batch, loss_factor=(0.5, 0.5, 1.0), loss_using=(1, 2, 3), reg_loss="l1_loss",
               masks=None, p=DROP_P, lead_choice=None, add_noise=False):
    """One iteration of Solver.run_one_epoch(phase='train') (solver/solver.py:157-189,232-235).
    Leaves gradients applied; returns the four loss floats."""
    out, out_p, out_l = forward(P, Bf, batch["data"], batch["input_theta"], batch["target_theta"],
                                batch["rois"], phase="train", training=True, masks=masks, p=p,
                                lead_choice=lead_choice)
    if add_noise:
        out = out + batch["noise"].unsqueeze(1)
    losses = loss_v1(out, out_p, out_l, batch["target_view"].unsqueeze(1), loss_factor, loss_using, reg_loss)
    losses[0].backward()
    vals = [float(v.detach()) for v in losses]
    opt.step(P)
    return vals


def dp_train_step(P, Bf_ranks, opt, shards, masks_ranks=None, p=DROP_P, lead_choice=None, **loss_kw):
    """One data-parallel iteration with the semantics of the reference's `nn.DataParallel` wrapper
    (solver/solver.py:32-34): every replica runs the forward on its equal-sized shard with
    BatchNorm statistics of THAT shard (replica r updates `Bf_ranks[r]`; replica 0's buffers are the module's own),
    the loss is the mean over the gathered batch, i.e. the parameter gradient is the average of the per-shard
    mean-loss gradients; one SGD step on the shared parameters.  Both Standin lead draws are shared by the replicas
    (the build's defined behaviour: identical seeds on every rank).  Returns (per-shard loss 4-tuples, averaged grads)."""
    if lead_choice is None:
        V = shards[0]["data"].shape[1]
        lead_choice = (random.randint(0, V - 1), random.randint(0, V - 1))
    world = len(shards)
    acc, vals = {}, []
    for r, batch in enumerate(shards):
        for v in P.values():
            v.grad = None
        out, out_p, out_l = forward(P, Bf_ranks[r], batch["data"], batch["input_theta"], batch["target_theta"],
                                    batch["rois"], phase="train", training=True,
                                    masks=None if masks_ranks is None else masks_ranks[r], p=p, lead_choice=lead_choice)
        losses = loss_v1(out, out_p, out_l, batch["target_view"].unsqueeze(1), **loss_kw)
        losses[0].backward()
        vals.append([float(v.detach()) for v in losses])
        for k, v in P.items():
            if v.grad is not None:
                acc[k] = v.grad.clone() if k not in acc else acc[k] + v.grad
    for k, v in P.items():
        v.grad = acc[k] / world if k in acc else None
    avg = {k: v.grad.clone() for k, v in P.items() if v.grad is not None}
    opt.step(P)
    return vals, avg


def require_grad(P):
    for v in P.values():
        v.requires_grad_(True)
    return P


def reference_style_init(V, seed=123):
    """Random init with the reference's distributions: encoder convs normal(0, sqrt(2/(k*k*C_out)))
    (resnet_1d.py:114-120), everything else torch's default kaiming-uniform(a=sqrt(5)); BN gamma=1 beta=0."""
    g = torch.Generator().manual_seed(seed)
    P = {}
    for name, shp in param_shapes(V).items():
        if name.startswith("W_encoder."):
            std = math.sqrt(2.0 / (shp[2] * shp[2] * shp[0]))
            P[name] = torch.randn(shp, generator=g) * std
        elif ".double_conv.1." in name or ".double_conv.4." in name:
            P[name] = torch.ones(shp) if name.endswith("weight") else torch.zeros(shp)
        else:
            fan_in = shp[1] * (shp[2] if len(shp) == 3 else 1) if len(shp) > 1 else None
            if fan_in is None:   # bias: fan_in of the matching weight
                wshp = param_shapes(V)[name[:-4] + "weight"]
                fan_in = wshp[1] * (wshp[2] if len(wshp) == 3 else 1)
            bound = 1.0 / math.sqrt(fan_in)
            P[name] = (torch.rand(shp, generator=g) * 2 - 1) * bound
    return P
