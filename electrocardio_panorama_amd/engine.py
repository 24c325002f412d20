"""Forward / backward schedule of the Nef-Net step on top of the HIP ops (ops.py).

This is the hand-written equivalent of what torch autograd derives for the reference's
`Model_nefnet.forward` (reference codes/network/model_nefnet.py:109-194): one explicit forward that
records the tensors the backward needs, and one explicit backward that walks the same graph in
reverse.  Parameters and BatchNorm buffers are addressed by the reference's state_dict names.
All arithmetic happens in libnefnet_hip.so; this file only sequences launches.
"""
import os

import torch
from . import _env

from . import ops
from .ops import GV

N_SEG, ROI_BINS = 7, 16
DROP_P = 0.2           # every nn.Dropout on the path (model_nefnet.py:46, encoder/resnet_1d.py:37)
BN_EPS, BN_MOM = 1e-5, 0.1

# NEF_FUSE_L2=0: third decoder conv on a materialised u2 = up2(relu(bn(c2))) instead of the (affine + ReLU, x2) prologue on c2.
# Round 2 measured the prologue form slower (60.2 against 58.8 ms/step: the register-staged weight gradient paid for the
# prologue per staged element); since the LDS-DMA weight gradient interpolates while it forms its fragments it is the
# faster one: 47.85 -> 47.36 ms/step, and the 1.97 GB tensor is never written.  On the split-fp16 kernels of round 4 the two
# forms are equal within the run-to-run noise (three alternating runs on one box: 31.2 - 31.7 against 31.5 - 31.9 ms/step,
# bit-identical loss): the prologue stays.
_FUSE_L2 = _env.get("NEF_FUSE_L2", "1") == "1"

# NEF_FUSE_STATS=0: BatchNorm statistics by a pass over the conv output (nef_bn_train_stats) instead of the conv epilogue
_FUSE_STATS = _env.get("NEF_FUSE_STATS", "1") == "1"
# NEF_BNB_UP=0: the BatchNorm-backward sums behind a x2 upsampling by the pass (bn_relu_bwd_up reduces them itself) instead of the
# backward-data conv's epilogue
_BNB_UP = _env.get("NEF_BNB_UP", "1") == "1"
# NEF_FOLD_CHSCALE=0: the theta scaling in front of w_conv as a pass of its own (chscale_fwd) instead of in_scale / res_scale on the block's convs
_FOLD_CHSCALE = _env.get("NEF_FOLD_CHSCALE", "1") == "1"
_FOLD_CHSCALE_BWD = _env.get("NEF_FOLD_CHSCALE_BWD", "1") == "1"      # ... and its backward in the epilogue of the block's last backward-data launch

_BWD_F4 = _env.get("NEF_BWD_F4", "1")

DROPOUT_SITES = ("W_encoder.layer1.0", "W_encoder.layer1.1", "W_encoder.layer1.2", "w_conv.0", "z1_conv.0",
                 "z2_conv1.0", "z2_conv2.0", "z2_conv2.2")


class DropCfg:
    """Dropout behaviour of one forward: off, replayed keep-masks (parity tests), or the in-kernel counter RNG."""

    def __init__(self, training, p=DROP_P, masks=None, seed=0, seed_dev=None):
        self.seed_dev = seed_dev           # device uint64 added to the seed at run time (hipGraph replay)
        self.on = bool(training) and p > 0.0
        self.p = p if self.on else 0.0
        self.scale = 1.0 / (1.0 - p) if self.on else 1.0
        self.masks = masks
        self.seed = seed

    def windowed(self, site, t0, W):
        """Same behaviour on a time window of `site`'s activation (replayed masks are cropped to the window)."""
        if not self.on or self.masks is None:
            return self
        m = dict(self.masks)
        m[site] = self.masks[site][:, :, t0:t0 + W].contiguous()
        return DropCfg(True, self.p, m, self.seed, self.seed_dev)

    def args(self, site):
        if not self.on:
            return dict(mask=None, drop_p=0.0, drop_scale=1.0, seed=0)
        if self.masks is not None:
            return dict(mask=self.masks[site], drop_p=0.0, drop_scale=self.scale, seed=0)
        return dict(mask=None, drop_p=self.p, drop_scale=self.scale, seed_dev=self.seed_dev,
                    seed=(self.seed * 0x9E3779B1 + DROPOUT_SITES.index(site) * 0x85EBCA6B + 1) & 0xFFFFFFFFFFFFFFFF)


# ----------------------------------------------------------------------------------------------
# BasicBlock (resnet_1d.py:42-53 with k=7; model_nefnet.py:48-60 with k=3)
# ----------------------------------------------------------------------------------------------
def block_fwd(xv, P, prefix, K, Cog, drop, in_scale=None):
    """`in_scale` (tensor [B, G, Cin], batch stride, group stride): the block's input is xv * in_scale per (sample, channel) -- applied
    by the first conv while it stages and by the last one to its residual, so the scaled tensor is never written (split-fp16
    launches, identity residual only: the caller checks block_scale_ok)."""
    G = xv.G
    h = ops.conv(xv, ops.pack_weight(P[prefix + ".conv1.weight"], G, T=xv.T), Cog, K, relu=True, in_scale=in_scale, **drop.args(prefix))
    res_conv = (K == 3 and Cog != xv.Cg)                     # model_nefnet.py:54
    assert in_scale is None or not res_conv
    if res_conv:
        r = ops.conv(xv, ops.pack_weight(P[prefix + ".residual_conv.weight"], G, T=xv.T), Cog, 1,
                     bias=P[prefix + ".residual_conv.bias"])
        resv = GV.dense(r, G)
    else:
        resv = xv
    y = ops.conv(GV.dense(h, G), ops.pack_weight(P[prefix + ".conv2.weight"], G, T=xv.T), Cog, K, res=resv, relu=True,
                 res_scale=in_scale)
    return y, (xv, h, y, prefix, K, Cog, res_conv, drop.scale, in_scale)


def block_scale_ok(P, prefix, G, T):
    """Both convs of the block on the split-fp16 kernel at full-size tiles (what in_scale / res_scale need)?"""
    return all(ops._pack_shape(P[prefix + n], G, False, T, plain=False)[3] == 3 for n in (".conv1.weight", ".conv2.weight")) and T >= 128


def _bwd_f4(K):
    """Backward-data launches of the encoder-side blocks take F(4,3) (K=3) resp. F(4,4) + F(4,3) (K=7): no ReLU decision is ever taken
    on a gradient, so the exact-zero / decision-flip argument that keeps their FORWARD convs on F(2,3) does not apply.
    Measured on the reference's 3-step SGD trajectory: worst parameter 1.4e-5 (bar 2e-4), stem weight 1.8e-6.
    NEF_BWD_F4=0: F(2,3) there too; =3: K=3 only."""
    return _BWD_F4 != "0" and (K == 3 or (K == 7 and _BWD_F4 != "3"))


def _block_pack_requests(P, prefix, G, T, flip):
    """The operands block_fwd / block_bwd will ask ops.pack_weight for, for ops.pack_many (one launch per pass)."""
    w1 = P[prefix + ".conv1.weight"]
    f4 = bool(flip) and _bwd_f4(w1.shape[2])
    reqs = [(P[prefix + ".conv1.weight"], G, flip, T, f4), (P[prefix + ".conv2.weight"], G, flip, T, f4)]
    if w1.shape[2] == 3 and w1.shape[0] // G != w1.shape[1]:          # block_fwd's res_conv condition
        reqs.append((P[prefix + ".residual_conv.weight"], G, flip, T))
    return reqs


def _latent_pack_requests(P, V, T, win, flip):
    reqs = []
    for i in range(3):
        reqs += _block_pack_requests(P, f"W_encoder.layer1.{i}", V, T, flip)
    reqs += _block_pack_requests(P, "w_conv.0", V, T, flip)
    reqs += _block_pack_requests(P, "z1_conv.0", V, T, flip)
    reqs += _block_pack_requests(P, "z2_conv1.0", V, win[1] if win is not None else T, flip)
    reqs += _block_pack_requests(P, "z2_conv2.0", N_SEG * V, ROI_BINS, flip)
    reqs += _block_pack_requests(P, "z2_conv2.2", N_SEG * V, 2 * ROI_BINS, flip)
    return reqs


def _decoder_pack_requests(P, T_lat, flip):
    """Decoder convs whose weight is used as stored (the first one is regrouped per call): output lengths 2T, 4T, 4T.  The conv
    behind the second upsampling packs its own (phase) weights when it runs in polyphase form."""
    w3 = P["decoder.3.double_conv.0.weight"]
    poly = (ops.poly_bwd_ok if flip else ops.poly_fwd_ok)(1, w3.shape[0], w3.shape[1], 4 * T_lat)
    return ([(P["decoder.1.double_conv.3.weight"], 1, flip, 2 * T_lat, True)] +
            ([] if poly else [(w3, 1, flip, 4 * T_lat, True)]) + [(P["decoder.3.double_conv.3.weight"], 1, flip, 4 * T_lat, True)])


def _step_pack_requests(P, V, T):
    """Every conv operand of a whole TRAIN step -- forward and backward-data, the polyphase ones included -- for ONE ops.pack_many
    launch at the top of engine.forward(save=True) (round 5: 8 pack launches + 4 poly_weights launches per step).  The pass-level
    pack_many calls further down find their requests in the table and do nothing; anything this list misses packs on demand."""
    r0 = (T - 1) // 2
    win = (r0 - 2, 6) if (T % 2 == 0 and r0 - 2 >= 0 and r0 + 4 <= T) else None
    reqs = _latent_pack_requests(P, V, T, win, False) + _latent_pack_requests(P, V, T, win, True)
    if 2 * T < 128:            # (_fusable: shorter sequences take the unfused decoder, which packs on demand)
        return reqs
    reqs += _decoder_pack_requests(P, T, False) + _decoder_pack_requests(P, T, True)
    w0, w3 = P["decoder.1.double_conv.0.weight"], P["decoder.3.double_conv.0.weight"]
    wg, site0 = _regroup_halves(w0, P), w0.data_ptr()
    co0, co3 = w0.shape[0], w3.shape[0]
    # first decoder conv (shared halves: 2 groups) and the conv behind the second upsampling: polyphase operands where the shape allows
    if ops.poly_fwd_ok(2, co0, w0.shape[1] // 2, 2 * T):
        reqs.append((wg, 2, False, T, False, ("poly", co0), False, site0))
    else:
        reqs.append((wg, 2, False, 2 * T, True, None, True, site0))
    if ops.poly_bwd_ok(2, co0, w0.shape[1] // 2, 2 * T):
        reqs.append((wg, 2, True, T, False, ("poly", 0), True, site0))
    else:
        reqs.append((wg, 2, True, 2 * T, True, None, True, site0))
    if ops.poly_fwd_ok(1, co3, w3.shape[1], 4 * T):
        reqs.append((w3, 1, False, 2 * T, False, ("poly", co3), False, None))
    if ops.poly_bwd_ok(1, co3, w3.shape[1], 4 * T):
        reqs.append((w3, 1, True, 2 * T, False, ("poly", 0), True, None))
    return reqs


# Below this many latent elements per step (B * 128V * T) the step is bound by the host issuing launches, and the second
# stream's events and waits cost more than the overlap returns (batch 32 / L=512: 4.30 ms with it, 3.45 without)
_SIDE_MIN_WORK = 1 << 24


def _side(device, work=None):
    mode = _env.get("NEF_SIDE_STREAM", "auto")
    if mode == "0" or (mode == "auto" and work is not None and work < _SIDE_MIN_WORK
                       and not torch.cuda.is_current_stream_capturing()):
        return ops._Inline()
    if torch.cuda.is_current_stream_capturing() and _env.get("NEF_GRAPH_SIDE", "1") == "0":
        return ops._Inline()          # NEF_GRAPH_SIDE=0: a captured step stays on the capturing stream
    return ops.SideStream.get(device)


def block_bwd(saved, gy, P, grads, out=None, side=None, pre_gated=False, gate_input=False, scale_bwd=False):
    """Accumulates the block's parameter gradients into `grads`; returns the gradient wrt the block input
    (written into the GV `out` when given, e.g. one half of the z1/z2 split).  Weight / bias gradients are issued on
    the side stream: they are off the dependency chain and overlap with the chain's HBM-bound kernels.
    `gate_input`: the block input is itself a ReLU output whose producer would mask this gradient first thing in its own
    backward -- apply that mask in the epilogue of the last conv here; the producer is then called with `pre_gated`."""
    xv, h, y, prefix, K, Cog, res_conv, dscale = saved[:8]
    in_scale = saved[8] if len(saved) > 8 else None      # the block input is xv * in_scale: the returned gradient is wrt THAT product
    side = side or ops._Inline()
    G, Cig = xv.G, xv.Cg
    g2 = gy if pre_gated else ops.gate(gy, y)                # through the final ReLU
    g2v, hv = GV.dense(g2, G), GV.dense(h, G)
    grads[prefix + ".conv2.weight"] = side.run(lambda: ops.conv_bwd_weight(hv, g2v, K, site=P[prefix + ".conv2.weight"].data_ptr()), h, g2)
    # through conv2, then dropout and the inner ReLU: h > 0 <=> ReLU active and kept
    gc1 = ops.conv(g2v, ops.pack_weight(P[prefix + ".conv2.weight"], G, flip=True, T=xv.T, f4=_bwd_f4(K)), Cog, K, gate=hv,
                   gate_scale=dscale, role="conv_bwd_data")
    gc1v = GV.dense(gc1, G)
    grads[prefix + ".conv1.weight"] = side.run(lambda: ops.conv_bwd_weight(xv, gc1v, K, in_scale=in_scale, site=P[prefix + ".conv1.weight"].data_ptr()), xv.t, gc1)
    if res_conv:
        grads[prefix + ".residual_conv.weight"] = side.run(lambda: ops.conv_bwd_weight(xv, g2v, 1, site=P[prefix + ".residual_conv.weight"].data_ptr()), xv.t, g2)
        grads[prefix + ".residual_conv.bias"] = side.run(lambda: ops.chan_sum(g2), g2)
        gres = ops.conv(g2v, ops.pack_weight(P[prefix + ".residual_conv.weight"], G, flip=True, T=xv.T), Cig, 1,
                        role="conv_bwd_data")
        resv = GV.dense(gres, G)
    else:
        resv = g2v
    wp1f = ops.pack_weight(P[prefix + ".conv1.weight"], G, flip=True, T=xv.T, f4=_bwd_f4(K))
    if in_scale is not None and scale_bwd and int(getattr(wp1f, "nef_wino", 0)) == 3 and out is None:
        # the channel scaling's own backward in this launch's epilogue (ops.chscale_bwd with relu_x): the gradient wrt the UNSCALED
        # input, masked where that input (a ReLU output) is zero, and the per-(sample, channel) sums of (gradient x input)
        slots = ops.conv_stats_buffer(wp1f, xv.B, G, Cig, xv.T, xv.t.device)
        if slots is not None:
            gx = ops.conv(gc1v, wp1f, Cig, K, res=resv, gate=xv, gate_scale=1.0, gate_rowscale=in_scale, stats=slots, stats_mode=1,
                          role="conv_bwd_data")
            return gx, ops.slots_to_rows(slots, xv.B)
    return ops.conv(gc1v, wp1f, Cig, K, res=resv,
                    out=out, gate=xv if gate_input else None, gate_scale=1.0, role="conv_bwd_data")


# ----------------------------------------------------------------------------------------------
# decoder (model_nefnet.py:101-107) + sigmoid(x/3) (:168), P passes stacked along batch
# ----------------------------------------------------------------------------------------------
_DEC = (("decoder.1", "0", "1", 128), ("decoder.1", "3", "4", 128), ("decoder.3", "0", "1", 64),
        ("decoder.3", "3", "4", 64))


_HALVES = "decoder.1.double_conv.0.weight#halves"      # key of the regrouped first decoder weight inside a pass's parameter dict


def _regroup_halves(w, P=None):
    """decoder.1.double_conv.0.weight [128, 256, 3] -> grouped-conv weight [2*128, 128, 3] (group = input-channel half).  With the
    pass's parameter dict `P` the regrouped tensor is made once per step (forward) and found again by the backward pass -- the same
    OBJECT, which is also what lets the step-level ops.pack_many hand its packed operands to both."""
    if P is not None:
        hit = P.get(_HALVES)
        if hit is not None and hit[1] is w and hit[2] == w._version:
            return hit[0]
    wg = ops.regroup_halves(w)
    if P is not None:
        P[_HALVES] = (wg, w, w._version)
    return wg


def _ungroup_halves(w2):
    return ops.regroup_halves(w2, inverse=True)


def _fusable(D, passes):
    """The prologue variants index the per-pass BN affine by tile, so a column tile must not span two passes: always
    true once a sample fills a tile (2T >= 128, i.e. L >= 256); shorter sequences take the unfused path."""
    return 2 * D.shape[2] >= 128


def _decoder_fwd_unfused(D, P, Bf, passes, training, save):
    x = D
    saved = []
    for li, (blk, cv, bn, cout) in enumerate(_DEC):
        if li in (0, 2):
            x = ops.upsample2_fwd(x)
        wname, bname, pre = f"{blk}.double_conv.{cv}.weight", f"{blk}.double_conv.{cv}.bias", f"{blk}.double_conv.{bn}"
        c = ops.conv(GV.dense(x, 1), ops.pack_weight(P[wname], 1, T=x.shape[2], f4=True), cout, 3, bias=P[bname])
        if training:
            mean, invstd, a, b = ops.bn_train_stats(c, P[pre + ".weight"], P[pre + ".bias"], Bf[pre + ".running_mean"],
                                                    Bf[pre + ".running_var"], passes, BN_EPS, BN_MOM, nbt=Bf[pre + ".num_batches_tracked"])
            np_ = passes
        else:
            a, b = ops.bn_eval_affine(P[pre + ".weight"], P[pre + ".bias"], Bf[pre + ".running_mean"],
                                      Bf[pre + ".running_var"], BN_EPS)
            mean = invstd = None
            np_ = 1
        act = ops.affine_relu_fwd(c, a, b, np_)
        if save:
            saved.append((x, c, mean, invstd, a, b, (0, None, None, 1), li in (0, 2)))
        x = act
    out = ops.outconv_fwd(x, P["decoder.4.weight"], P["decoder.4.bias"])
    return out, (saved, x, out, passes, None)


def decoder_fwd(D, P, Bf, passes, training, save, shared_B=None):
    """Upsample -> (conv, BN, ReLU) x2 -> Upsample -> (conv, BN, ReLU) x2 -> conv -> sigmoid(x/3).  Only the conv
    outputs c1..c4 are materialised: the x2 upsampling and each BatchNorm-affine + ReLU are applied by the CONSUMING
    kernel while it stages its input (conv prologue modes, outconv prologue)."""
    if not _fusable(D, passes):
        assert shared_B is None
        return _decoder_fwd_unfused(D, P, Bf, passes, training, save)
    x, pro_in = D, None                # pro_in: (a, b) of the BN whose output feeds the next conv
    saved = []
    ops.pack_many(_decoder_pack_requests(P, D.shape[2], False))
    # shared_B: D holds only the two distinct inputs of the three Standin passes (ops.mix_fwd_shared); the first conv
    # runs once per distinct channel half (a 2-group conv) and pass_combine_fwd assembles the three pass outputs
    N = D.shape[0] if shared_B is None else 3 * shared_B
    for li, (blk, cv, bn, cout) in enumerate(_DEC):
        wname, bname, pre = f"{blk}.double_conv.{cv}.weight", f"{blk}.double_conv.{cv}.bias", f"{blk}.double_conv.{bn}"
        if li == 2 and not _FUSE_L2:
            # NEF_FUSE_L2=0 (the round-2 form): this layer materialises u2 = up(relu(bn(c2))) and runs the plain conv
            x_in = ops.upsample2_aff_fwd(x, pro_in[0], pro_in[1], pro_in[2])
            pro = (0, None, None, 1)
            up_after = True
        else:
            x_in = x
            mode = (2 if li in (0, 2) else 0) | (1 if pro_in is not None else 0)
            pro = (mode, pro_in[0], pro_in[1], pro_in[2]) if pro_in is not None else (mode, None, None, 1)
            up_after = bool(mode & 2)
        T_out = x_in.shape[2] * (2 if up_after and pro[0] & 2 else 1)
        stats = xedge = None
        # the two convs behind a x2 upsampling run in polyphase form where the shape allows: a conv over the half-resolution input
        # whose rows are the two output phases (ops.conv_poly_fwd) -- half the staged elements, no interpolation arithmetic
        poly = bool(pro[0] & 2) and ops.poly_fwd_ok(2 if (li == 0 and shared_B is not None) else 1, cout,
                                                     x_in.shape[1] // (2 if (li == 0 and shared_B is not None) else 1), T_out)
        if li == 0 and shared_B is not None:
            if poly:
                p2 = ops.conv_poly_fwd(GV.dense(x_in, 2), _regroup_halves(P[wname], P), cout, pro=pro, site=P[wname].data_ptr(), save_edge=save)
                xedge = p2.nef_xedge
            else:
                p2 = ops.conv(GV.dense(x_in, 2), ops.pack_weight(_regroup_halves(P[wname], P), 2, T=T_out, f4=True, site=P[wname].data_ptr()), cout, 3, pro=pro)
            if training and passes == 3:      # the BatchNorm statistics of c1 come out of the same pass
                c, *stats = ops.pass_combine_fwd_stats(p2, P[bname], shared_B, P[pre + ".weight"], P[pre + ".bias"],
                                                       Bf[pre + ".running_mean"], Bf[pre + ".running_var"], BN_EPS, BN_MOM,
                                                       nbt=Bf[pre + ".num_batches_tracked"])
            else:
                c = ops.pass_combine_fwd(p2, P[bname], shared_B)
        elif poly:
            slots = None
            if training and _FUSE_STATS:
                c, slots = ops.conv_poly_fwd(GV.dense(x_in, 1), P[wname], cout, bias=P[bname], pro=pro, stats=True, save_edge=save)
            else:
                c = ops.conv_poly_fwd(GV.dense(x_in, 1), P[wname], cout, bias=P[bname], pro=pro, save_edge=save)
            xedge = c.nef_xedge
            if slots is not None:
                stats = ops.bn_stats_from_slots(slots, P[pre + ".weight"], P[pre + ".bias"], Bf[pre + ".running_mean"],
                                                Bf[pre + ".running_var"], passes, N, T_out, BN_EPS, BN_MOM,
                                                nbt=Bf[pre + ".num_batches_tracked"])
        else:
            wp = ops.pack_weight(P[wname], 1, T=T_out, f4=True)
            # train mode: the F(4,3) epilogue leaves the BatchNorm slot sums of c -- no statistics pass over c
            slots = ops.conv_stats_buffer(wp, N, 1, cout, T_out, D.device) if (training and _FUSE_STATS) else None
            c = ops.conv(GV.dense(x_in, 1), wp, cout, 3, bias=P[bname], pro=pro, stats=slots)
            if slots is not None:
                stats = ops.bn_stats_from_slots(slots, P[pre + ".weight"], P[pre + ".bias"], Bf[pre + ".running_mean"],
                                                Bf[pre + ".running_var"], passes, N, T_out, BN_EPS, BN_MOM,
                                                nbt=Bf[pre + ".num_batches_tracked"])
        if stats is not None:      # (num_batches_tracked += passes: by the launch that updated the running statistics)
            mean, invstd, a, b = stats
            Bp = N // passes
        elif training:
            mean, invstd, a, b = ops.bn_train_stats(c, P[pre + ".weight"], P[pre + ".bias"], Bf[pre + ".running_mean"],
                                                    Bf[pre + ".running_var"], passes, BN_EPS, BN_MOM, nbt=Bf[pre + ".num_batches_tracked"])
            Bp = N // passes
        else:
            a, b = ops.bn_eval_affine(P[pre + ".weight"], P[pre + ".bias"], Bf[pre + ".running_mean"],
                                      Bf[pre + ".running_var"], BN_EPS)
            mean = invstd = None
            Bp = N
        if save:
            saved.append((x_in, c, mean, invstd, a, b, pro, up_after, xedge))
        x, pro_in = c, (a, b, Bp)
    out = ops.outconv_fwd(x, P["decoder.4.weight"], P["decoder.4.bias"], pro=pro_in)
    return out, (saved, x, out, passes, pro_in, shared_B)


def decoder_bwd(dsaved, g_out, P, grads, side=None):
    saved, c4, out, passes, pro4 = dsaved[:5]
    shared_B = dsaved[5] if len(dsaved) > 5 else None
    side = side or ops._Inline()
    if len(dsaved) > 5:                # the fused decoder: its backward-data operands in one launch
        ops.pack_many(_decoder_pack_requests(P, c4.shape[2] // 4, True))
    grads["decoder.4.weight"], grads["decoder.4.bias"] = side.run(
        lambda: ops.outconv_bwd_weight(g_out, out, c4, pro=pro4), g_out, out, c4)
    # the last BatchNorm's backward rebuilds the last conv's input gradient from go on the fly (never materialised)
    fuse_last = c4.shape[2] % 4 == 0
    g_is_up = False
    g = None if fuse_last else ops.outconv_bwd_data(g_out, out, P["decoder.4.weight"], c4.shape[1])
    g_slots = None            # BatchNorm-backward sums the conv that produced g left in its epilogue
    poly_first = False        # the first layer's backward-data pass already went through its upsampling
    for li in (3, 2, 1, 0):
        blk, cv, bn, cout = _DEC[li]
        x, c, mean, invstd, a, b, pro, up_after = saved[li][:8]
        xedge = saved[li][8] if len(saved[li]) > 8 else None
        # polyphase backward of a conv behind the x2 upsampling (ops.conv_bwd_data_poly / conv_bwd_weight_poly): with the forward's
        # row-end values at hand the BatchNorm-backward pass writes this conv's output gradient PHASE-MAJOR ([.., 2C, T/2]) and both
        # backward convs run on that at half resolution; without them (forward not in polyphase form) only the backward-data pass does
        sh0 = li == 0 and shared_B is not None
        Cog_, Cig_ = c.shape[1], x.shape[1] // (2 if sh0 else 1)
        poly_b = bool(up_after and pro[0] & 2 and (not sh0 or pro[0] == 2) and ops.poly_bwd_ok(2 if sh0 else 1, Cog_, Cig_, c.shape[2]))
        pm = bool(poly_b and xedge is not None and ops.poly_w_ok((2 if sh0 else 1) * (shared_B if sh0 else c.shape[0]), 2 if sh0 else 1, Cog_, Cig_, c.shape[2]))
        wname, bname, pre = f"{blk}.double_conv.{cv}.weight", f"{blk}.double_conv.{cv}.bias", f"{blk}.double_conv.{bn}"
        pm0_done = False
        if li == 3 and fuse_last:
            gc, gg, gbeta, gbias = ops.bn_relu_bwd_outconv(g_out, out, P["decoder.4.weight"], c, mean, invstd, a, b, passes)
        elif li == 0 and shared_B is not None and not g_is_up:
            # gc is the per-half gradient [2B, 2*128, 2T] straight away (pass_combine_bwd fused into the apply pass)
            gc, gg, gbeta, gbias = ops.bn_relu_bwd_combine3(g, c, mean, invstd, a, b, slots=g_slots, phase_major=pm)
            pm0_done = pm
        elif g_is_up:
            gc, gg, gbeta, gbias = ops.bn_relu_bwd_up(g, c, mean, invstd, a, b, passes, slots=g_slots)
        else:
            gc, gg, gbeta, gbias = ops.bn_relu_bwd(g, c, P[pre + ".weight"], mean, invstd, a, b, passes,
                                                   with_chan_sum=True, slots=g_slots, phase_major=pm and not sh0)
            pm0_done = pm and not sh0
        grads[pre + ".weight"], grads[pre + ".bias"], grads[bname] = gg, gbeta, gbias
        pm = pm and pm0_done      # (the branches that did not write phase-major keep the interleaved forms)
        if li == 0 and shared_B is not None:
            gp2 = gc if gc.shape[0] == 2 * shared_B else ops.pass_combine_bwd(gc)   # [2B, 2*128, 2T] (pm: [2B, 4*128, T])
            gpv, xv = GV.dense(gp2, 2), GV.dense(x, 2)
            if pm:
                grads[wname] = side.run(lambda: _ungroup_halves(ops.conv_bwd_weight_poly(xv, gp2, Cog_, pro, xedge, site=P[wname].data_ptr())), x, gp2, xedge)
            else:
                grads[wname] = side.run(lambda: _ungroup_halves(ops.conv_bwd_weight(xv, gpv, 3, pro=pro, site=P[wname].data_ptr())), x, gp2)
            if poly_b:      # straight to the gradient wrt the half-resolution D (polyphase pass): nothing left for the consumer to fold
                g = ops.conv_bwd_data_poly(gpv, _regroup_halves(P[wname], P), x.shape[1] // 2, site=P[wname].data_ptr(), phase_major=pm)
                poly_first = True
            else:
                g = ops.conv(gpv, ops.pack_weight(_regroup_halves(P[wname], P), 2, flip=True, T=gp2.shape[2], f4=True, site=P[wname].data_ptr()), x.shape[1] // 2, 3,
                             role="conv_bwd_data")
        else:
            gcv, xv = GV.dense(gc, 1), GV.dense(x, 1)
            if pm:
                grads[wname] = side.run(lambda: ops.conv_bwd_weight_poly(xv, gc, Cog_, pro, xedge, site=P[wname].data_ptr()), x, gc, xedge)
            else:
                grads[wname] = side.run(lambda: ops.conv_bwd_weight(xv, gcv, 3, pro=pro, site=P[wname].data_ptr()), x, gc)
            if poly_b:
                # x2 upsampling in front of this conv: the polyphase pass leaves the gradient wrt the half-resolution input (= what the
                # BatchNorm below produced) directly, with that BatchNorm's backward sums in its epilogue
                bnb = None
                if li > 0 and _FUSE_STATS and saved[li - 1][2] is not None:
                    cb, mb, ib, ab, bb = saved[li - 1][1:6]
                    bnb = (cb, mb, ib, ab, bb, cb.shape[0] // passes)
                g = ops.conv_bwd_data_poly(gcv, P[wname], x.shape[1], bnb=bnb, phase_major=pm)
                g_slots = g.nef_slots
                g_is_up = False
                if li == 0:
                    poly_first = True
                continue
            wpf = ops.pack_weight(P[wname], 1, flip=True, T=gc.shape[2], f4=True)
            # g is the gradient wrt relu(bn(c_below)) when no upsampling sits in between: the epilogue then leaves the
            # reduction sums of that BatchNorm's backward (it reads c_below's tile for the ReLU decision and xhat)
            # (with the x2 upsampling in between, the sums are those of its adjoint -- what bn_relu_bwd_up reduces)
            g_slots = bnb = None
            if li > 0 and _FUSE_STATS:
                cb, mb, ib, ab, bb = saved[li - 1][1:6]
                Tg = gc.shape[2]
                plain = not up_after and cb.shape[2] == Tg
                upv = _BNB_UP and bool(up_after) and 2 * cb.shape[2] == Tg and Tg % 8 == 0 and Tg >= 16     # the g_is_up case below
                if (plain or upv) and mb is not None:
                    g_slots = ops.conv_stats_buffer(wpf, cb.shape[0], 1, x.shape[1], Tg, gc.device)
                    if g_slots is not None:
                        bnb = (cb, mb, ib, ab, bb, cb.shape[0] // passes, g_slots, upv)
            g = ops.conv(gcv, wpf, x.shape[1], 3, role="conv_bwd_data", bnb=bnb)
        # back through the x2 upsampling in front of this layer: the next BatchNorm backward takes the adjoint while it
        # reads (rows of 4k >= 8 samples), otherwise it is a pass of its own
        g_is_up = bool(up_after and li > 0 and c.shape[2] % 8 == 0 and c.shape[2] >= 16)
        if up_after and li > 0 and not g_is_up:
            g = ops.upsample2_bwd(g)
    # the adjoint of the FIRST upsampling is left to the consumer (mix_bwd takes it while reading)
    return g, bool(saved[0][7]) and not poly_first, shared_B is not None


# ----------------------------------------------------------------------------------------------
# Model_nefnet.forward / backward
# ----------------------------------------------------------------------------------------------
def _latents(P, x, in_theta, rois, drop, save, pack_side=None):
    """model_nefnet.py:117-138: everything up to (z1, z2 segments).  `pack_side`: the side stream the step-level weight pack was
    issued on (engine.forward): it runs under the stem -- which reads the raw stem weight, not a packed operand -- and is joined
    in front of the first conv that needs one."""
    B, V, L = x.shape
    T = L // 4
    sv = {}
    r0 = (T - 1) // 2
    win = (r0 - 2, 6) if (T % 2 == 0 and r0 - 2 >= 0 and r0 + 4 <= T) else None
    ops.pack_many(_latent_pack_requests(P, V, T, win, False))       # every conv operand of this function in one launch
    a = ops.stem_fwd(x, P["W_encoder.conv1.weight"])
    if pack_side is not None:
        pack_side.join()
    sv["blk_enc"] = []
    for i in range(3):
        a, s = block_fwd(GV.dense(a, V), P, f"W_encoder.layer1.{i}", 7, 128, drop)
        sv["blk_enc"].append(s)
    e = ops.theta_mlp_fwd(in_theta, P["mlp1.weight"], P["mlp1.bias"])           # [B, V, 128]
    if _FOLD_CHSCALE and block_scale_ok(P, "w_conv.0", V, T):
        # a * e is never written: w_conv's first conv scales while it stages, its last conv scales the residual (round 5)
        enc, sv["blk_w_conv"] = block_fwd(GV.dense(a, V), P, "w_conv.0", 3, 128, drop, in_scale=(e, V * 128, 128))
    else:
        ew = ops.chscale_fwd(a, e)
        enc, sv["blk_w_conv"] = block_fwd(GV.dense(ew, V), P, "w_conv.0", 3, 128, drop)
    z1, sv["blk_z1"] = block_fwd(GV.half(enc, V, 0), P, "z1_conv.0", 3, 128, drop)
    # z2_conv1 feeds only roi_algin, which reads exactly two time rows: run the block on the window of
    # six samples whose centre two are exact (k=3 twice -> 2 samples of context per side) instead of all T.
    if win is not None:
        xw = ops.window_crop(GV.half(enc, V, 1), win[0], win[1])
        wdrop = drop.windowed("z2_conv1.0", win[0], win[1])
        z2c, sv["blk_z2c"] = block_fwd(GV.dense(xw, V), P, "z2_conv1.0", 3, 128, wdrop)
        z2a = ops.roi_align_fwd(z2c, rois, T, win[0])                           # [B, 128V, 7, 16]
    else:
        z2c, sv["blk_z2c"] = block_fwd(GV.half(enc, V, 1), P, "z2_conv1.0", 3, 128, drop)
        z2a = ops.roi_align_fwd(z2c, rois)
    sv["z2_win"] = win
    h0 = z2a.view(B, 128 * V * N_SEG, ROI_BINS)                                 # raw memory order
    h1, sv["blk_c20"] = block_fwd(GV.dense(h0, N_SEG * V), P, "z2_conv2.0", 3, 128, drop)
    h2 = ops.convt2_fwd(h1, P["z2_conv2.1.weight"], P["z2_conv2.1.bias"], N_SEG * V)
    h3, sv["blk_c22"] = block_fwd(GV.dense(h2, N_SEG * V), P, "z2_conv2.2", 3, 128, drop)
    z2b = h3.view(B, 128 * V, N_SEG, 2 * ROI_BINS)
    if save:
        sv.update(x=x, w=a, e=e, in_theta=in_theta, h1=h1, rois=rois, T=T, V=V, B=B)
    return z1, z2b, (sv if save else None)


def _head_fwd(P, Bf, z1, z2r, q_theta, V, rest_theta, phase, training, lead_choice, sv, rest_chunk, half_sweep):
    """model_nefnet.py:146-190: lead means, Standin mixes, query scaling, the three decoder passes (+ the sweep)."""
    B = z1.shape[0]
    save = sv is not None
    q = ops.theta_mlp_fwd(q_theta, P["mlp2.weight"], P["mlp2.bias"])             # [B, 256]
    if 2 * z1.shape[2] >= 128:         # (_fusable) the first decoder conv sees each distinct channel half once
        latent, D2 = ops.lead_mean_mix_shared(z1, z2r, q, V, lead_choice)         # [B, 256, T], [2B, 256, T]
        out3, dsv = decoder_fwd(D2, P, Bf, 3, training, save, shared_B=B)
    else:
        latent = ops.lead_mean(z1, z2r, V)
        D = ops.mix_fwd(latent, z1, z2r, q, V, lead_choice)                       # [3B, 256, T]
        out3, dsv = decoder_fwd(D, P, Bf, 3, training, save)
    outs = (out3[0:B], out3[B:2 * B], out3[2 * B:3 * B])
    if save:
        sv.update(z1=z1, z2r=z2r, latent=latent, q=q, q_theta=q_theta, choice=lead_choice, dec=dsv, hB=B, hV=V)
    if phase == "train":
        return outs, sv
    if phase in ("val", "test"):
        rest = sweep(P, Bf, latent, rest_theta, training, rest_chunk, half_sweep)
        return outs + (rest,), sv
    raise KeyError("please type correct phase")


def forward(P, Bf, x, in_theta, q_theta, rois, rest_theta=None, phase="train", training=True, drop=None,
            lead_choice=(0, 0), save=False, rest_chunk=8, status=None, half_sweep=False):
    """Returns (outputs tuple, saved-state or None).  `lead_choice` are the two Standin lead indices
    (model_nefnet.py:154,156), drawn by the caller: a tuple of ints, or a device int32[2] tensor (graph replay)."""
    drop = drop or DropCfg(False)
    B, V, L = x.shape
    T = L // 4
    ops.BATCH_HINT = B     # small batches stay on the fp32 kernels (ops._h2_fills)
    ops.amax_roll()        # split-fp16 convs: last pass's operand magnitudes become this pass's input scales
    pack_side = None
    if save and phase == "train":
        # ONE pack launch for the whole step (~80 us at configs[1]), on the side stream: it overlaps the stem (round 6)
        reqs = _step_pack_requests(P, V, T)
        pack_side = _side(x.device, B * 128 * V * T)
        pack_side.run(lambda: ops.pack_many(reqs))
    z1, z2b, sv = _latents(P, x, in_theta, rois, drop, save, pack_side)
    if phase == "gen":
        return (z1, z2b), None
    z2r = ops.roi_unpool_fwd(z2b, rois, T, status)
    return _head_fwd(P, Bf, z1, z2r, q_theta, V, rest_theta, phase, training, lead_choice, sv, rest_chunk, half_sweep)


def _lead_view(t, i, V):
    """Lead i of a lead-blocked [B, 128V, T] tensor as a one-group view."""
    B, Ct, T = t.shape
    return GV(t, B, 1, 128, T, Ct * T, 128 * T, i * 128 * T)


def forward2(P, Bf, x, in_theta, q_theta, rois, rest_theta=None, phase="train", training=True, drop=None,
             lead_choice=(0, 0), save=False, rest_chunk=8, status=None, half_sweep=False):
    """Model_nefnet2.forward (reference codes/network/model_nefnet2.py:118-194): ONE single-lead encoder shared by all
    leads.  The reference loops over the leads; here the leads are folded into the batch (lead-major, n = v*B + b), so
    the shared-weight encoder runs once on V*B single-lead samples, and the two extra convs (`single_conv_z1/_z2`,
    :140,:149) write their outputs lead by lead into lead-blocked [B, 128V, T] tensors, from where the lead mean, the
    Standin mixes and the decoder are exactly Model_nefnet's."""
    drop = drop or DropCfg(False)
    B, V, L = x.shape
    T, N = L // 4, V * B
    ops.BATCH_HINT = N
    ops.amax_roll()
    xf = x.transpose(0, 1).reshape(N, 1, L).contiguous()
    thf = in_theta.transpose(0, 1).reshape(N, 1, 2).contiguous()
    roisf = rois.repeat(V, 1, 1)
    z1f, z2bf, sv = _latents(P, xf, thf, roisf, drop, save)
    z2rf = ops.roi_unpool_fwd(z2bf, roisf, T, status)                              # [N, 128, T]
    Z1 = torch.empty(B, 128 * V, T, device=x.device, dtype=torch.float32)
    Z2 = torch.empty_like(Z1)
    wp1 = ops.pack_weight(P["single_conv_z1.0.weight"], 1, T=T)
    wp2 = ops.pack_weight(P["single_conv_z2.0.weight"], 1, T=T)
    for i in range(V):
        ops.conv(GV.dense(z1f[i * B:(i + 1) * B], 1), wp1, 128, 3, bias=P["single_conv_z1.0.bias"],
                 out=_lead_view(Z1, i, V))
        ops.conv(GV.dense(z2rf[i * B:(i + 1) * B], 1), wp2, 128, 3, bias=P["single_conv_z2.0.bias"],
                 out=_lead_view(Z2, i, V))
    if phase == "gen":                                                             # :158-159: the two lead means
        latent = ops.lead_mean(Z1, Z2, V)
        return (latent[:, :128].contiguous(), latent[:, 128:].contiguous()), None
    if save:
        sv.update(z1f=z1f, z2rf=z2rf, fold=(B, V))
    return _head_fwd(P, Bf, Z1, Z2, q_theta, V, rest_theta, phase, training, lead_choice, sv, rest_chunk, half_sweep)


def sweep(P, Bf, latent, query_thetas, training=False, chunk=8, half=False):
    """Decode `latent` [B,256,T] at Q query angles [B,Q,2] -> [B,Q,L] (model_nefnet.py:181-190, :207-216).
    In training mode each angle is its own BatchNorm pass, in the reference's order.  `half` selects the fp16
    matrix-core decoder for the eval-mode sweep (opt-in; fp32 is the reference behaviour)."""
    if not training:
        if half:
            return sweep_eval_h(P, Bf, latent, query_thetas)
        return sweep_eval(P, Bf, latent, query_thetas, chunk)
    B, Q = query_thetas.shape[0], query_thetas.shape[1]
    rq = ops.theta_mlp_fwd(query_thetas, P["mlp2.weight"], P["mlp2.bias"])       # [B, Q, 256]
    L = latent.shape[2] * 4
    rest = torch.empty(B, Q, L, device=latent.device, dtype=torch.float32)
    for q0 in range(0, Q, chunk):
        n = min(chunk, Q - q0)
        Dq = torch.empty(n * B, 256, latent.shape[2], device=latent.device, dtype=torch.float32)
        for i in range(n):
            Dq[i * B:(i + 1) * B] = ops.chscale_fwd(latent, rq[:, q0 + i].contiguous())
        o, _ = decoder_fwd(Dq, P, Bf, n, training, False)                         # [n*B, 1, L]
        rest[:, q0:q0 + n] = o.view(n, B, L).transpose(0, 1)
    return rest


def sweep_eval(P, Bf, latent, query_thetas, chunk=8):
    """Eval-mode sweep (the panorama): BatchNorm is a fixed affine, so it is folded into the conv weights once per
    call; the per-angle scaling commutes with the x2 upsampling, so the latent is upsampled once and the first conv
    applies q[b, angle, :] while staging its input -- no per-angle elementwise passes at all."""
    B, Q, T = query_thetas.shape[0], query_thetas.shape[1], latent.shape[2]
    dev = latent.device
    rq = ops.theta_mlp_fwd(query_thetas, P["mlp2.weight"], P["mlp2.bias"])       # [B, Q, 256]
    u = ops.upsample2_fwd(latent)                                                 # [B, 256, 2T], shared by all angles
    wp, bias = [], []
    for li, (blk, cv, bn, cout) in enumerate(_DEC):
        pre = f"{blk}.double_conv.{bn}"
        a, b = ops.bn_eval_affine(P[pre + ".weight"], P[pre + ".bias"], Bf[pre + ".running_mean"],
                                  Bf[pre + ".running_var"], BN_EPS)
        wf, bf_ = ops.fold_bn(P[f"{blk}.double_conv.{cv}.weight"], P[f"{blk}.double_conv.{cv}.bias"], a, b)
        # the folded weights are temporaries and every chunk of angles runs through them: one named call site per layer
        # (layer 0 runs with the per-angle channel scale: not a plain launch -- at 2T <= 64 it must not take the packed short-row form)
        wp.append(ops.pack_weight(wf, 1, T=(2 * T if li < 2 else 4 * T), f4=True, site=("sweep", li), shared=True, plain=li > 0))
        bias.append(bf_)
    uv = GV.dense(u, 1)
    rest = torch.empty(B, Q, 4 * T, device=dev, dtype=torch.float32)
    for q0 in range(0, Q, chunk):
        n = min(chunk, Q - q0)
        c1 = torch.empty(n * B, 128, 2 * T, device=dev, dtype=torch.float32)
        for i in range(n):
            ops.conv(uv, wp[0], 128, 3, bias=bias[0], relu=True, in_scale=(rq[:, q0 + i], Q * 256, 0),
                     out=GV.dense(c1[i * B:(i + 1) * B], 1))
        c2 = ops.conv(GV.dense(c1, 1), wp[1], 128, 3, bias=bias[1], relu=True)
        u2 = ops.upsample2_fwd(c2)
        c3 = ops.conv(GV.dense(u2, 1), wp[2], 64, 3, bias=bias[2], relu=True)
        c4 = ops.conv(GV.dense(c3, 1), wp[3], 64, 3, bias=bias[3], relu=True)
        o = ops.outconv_fwd(c4, P["decoder.4.weight"], P["decoder.4.bias"])       # [n*B, 1, L]
        rest[:, q0:q0 + n] = o.view(n, B, 4 * T).transpose(0, 1)
    return rest


PANO_FUSE_PAIR = _env.get("NEF_PANO_FUSE_PAIR", "1") != "0"   # measurement switch: 0 = two launches
PANO_FUSE_TAIL = _env.get("NEF_PANO_FUSE_TAIL", "1") != "0"   # round 6: layers 3 + 4 + last conv in one launch (L <= 512); 0 = two launches


def sweep_eval_h(P, Bf, latent, query_thetas, pair_budget=16384):
    """The eval-mode sweep on the fp16 matrix cores (pano_h.hip; SURVEY 8-f2, BASELINE configs 4/5).  Same folding as
    sweep_eval; activations are fp16 [pair][time][channel] with pair = (sample, angle) sample-major, so the result
    lands in rest[b, q] without a transposing copy.  Both x2 upsamplings and the per-angle query scaling happen while
    the consuming conv stages its input.  Opt-in: there is no reduced-precision behaviour in the reference."""
    B, Q, T = query_thetas.shape[0], query_thetas.shape[1], latent.shape[2]
    dev = latent.device
    rq = ops.theta_mlp_fwd(query_thetas, P["mlp2.weight"], P["mlp2.bias"])       # [B, Q, 256] fp32
    lat_h = ops.pano_h_from_f32(latent)                                           # [B, T, 256] fp16
    wp, bias = [], []
    for blk, cv, bn, cout in _DEC:
        pre = f"{blk}.double_conv.{bn}"
        a, b = ops.bn_eval_affine(P[pre + ".weight"], P[pre + ".bias"], Bf[pre + ".running_mean"],
                                  Bf[pre + ".running_var"], BN_EPS)
        wf, bf_ = ops.fold_bn(P[f"{blk}.double_conv.{cv}.weight"], P[f"{blk}.double_conv.{cv}.bias"], a, b)
        wp.append(ops.pano_h_pack_weight(wf))
        bias.append(bf_)
    rest = torch.empty(B, Q, 4 * T, device=dev, dtype=torch.float32)
    nq = max(1, min(Q, pair_budget // max(B, 1)))      # angles per chunk: bounds the fp16 intermediates
    bufs = None
    for q0 in range(0, Q, nq):
        n = min(nq, Q - q0)
        N = B * n
        if bufs is None or bufs[0].shape[0] != N:
            bufs = (torch.empty(N, 2 * T, 128, device=dev, dtype=torch.float16),
                    torch.empty(N, 2 * T, 128, device=dev, dtype=torch.float16),
                    torch.empty(N, 4 * T, 64, device=dev, dtype=torch.float16))
        if PANO_FUSE_PAIR:      # layers 1 + 2 in one pass (one tile per pair up to 256 rows, tiles of 252 rows beyond)
            c2 = ops.pano_h_conv_pair(lat_h, wp[0], bias[0], (rq[:, q0:], Q * 256, 256), wp[1], bias[1], N, n, n,
                                      out=bufs[1])
        else:
            c1 = ops.pano_h_conv(lat_h, wp[0], bias[0], 128, N=N, upsample=True, scale=(rq[:, q0:], Q * 256, 256),
                                 x_div=n, nq=n, out=bufs[0])
            c2 = ops.pano_h_conv(c1, wp[1], bias[1], 128, out=bufs[1])
        if PANO_FUSE_TAIL:      # layers 3 + 4 + last conv + sigmoid in one pass (one tile per pair up to 512 rows, tiles of 508 beyond)
            ops.pano_h_conv_tail(c2, wp[2], bias[2], wp[3], bias[3], P["decoder.4.weight"], P["decoder.4.bias"], rest[:, q0:], n,
                                 Q * 4 * T, 4 * T)
            continue
        c3 = ops.pano_h_conv(c2, wp[2], bias[2], 64, upsample=True, out=bufs[2])
        ops.pano_h_conv_outconv(c3, wp[3], bias[3], P["decoder.4.weight"], P["decoder.4.bias"], rest[:, q0:], n,
                                Q * 4 * T, 4 * T)
    return rest


def gen_ecg(P, Bf, z1, z2b, query_thetas, rois, chunk=8, half=False):
    """model_nefnet.py:196-218 (always eval-mode BatchNorm)."""
    V = z1.shape[1] // 128
    ops.BATCH_HINT = z1.shape[0]      # as forward(): the kernel choice follows THIS call's batch, not whichever pass ran last
    ops.amax_roll()                   # ... and the split-fp16 convs of the sweep start from the scales their last sweep left
    z2r = ops.roi_unpool_fwd(z2b.contiguous(), rois, z1.shape[2])
    latent = ops.lead_mean(z1, z2r, V)
    if half:
        return sweep_eval_h(P, Bf, latent, query_thetas)
    return sweep(P, Bf, latent, query_thetas, False, chunk)


def _head_bwd(P, sv, g_outs, grads, side, relu_z1=False):
    """Back through decoder passes, Standin mixes and mlp2: returns the gradients wrt the lead-blocked z1 and z2r."""
    B, V = sv["hB"], sv["hV"]
    like = sv["dec"][2]        # stacked decoder output [3B, 1, L]
    g_out = ops.stacked3(g_outs) if all(g is not None for g in g_outs) else None      # ops.loss_bwd's three views of one buffer: no copy
    if g_out is None:
        parts = [g if g is not None else torch.zeros_like(like[0:B]) for g in g_outs]
        g_out = torch.cat([p_.contiguous() for p_ in parts], dim=0)
    gD, up, shared = decoder_bwd(sv["dec"], g_out, P, grads, side)
    # relu_z1: z1 is the ReLU output of z1_conv's block, whose backward would start by masking gz1 -- done here
    if shared:
        gz1, gz2r, gq = ops.mix_bwd_shared_up(gD, sv["latent"], sv["z1"], sv["z2r"], sv["q"], V, sv["choice"],
                                              relu_z1=relu_z1)
    else:
        gz1, gz2r, gq = ops.mix_bwd(gD, sv["latent"], sv["z1"], sv["z2r"], sv["q"], V, sv["choice"], upsampled=up,
                                    relu_z1=relu_z1)
    gW2, gb2 = side.run(lambda: ops.theta_mlp_bwd(sv["q_theta"], gq, 256), gq)
    grads["mlp2.weight"], grads["mlp2.bias"] = gW2, gb2
    return gz1, gz2r


EARLY_HOOK = None      # callable(P, grads, side) invoked instead of parallel.early_reduce at the early-bucket point of backward()


def _latents_bwd(P, sv, gz1, gz2r, grads, side, z1_pre_gated=False, early=False):
    """Back through `_latents` (+ the segment un-pooling that follows it): encoder-side parameter gradients."""
    B, V, T = sv["B"], sv["V"], sv["T"]
    ops.pack_many(_latent_pack_requests(P, V, T, sv["z2_win"], True))
    gz2b = ops.roi_unpool_bwd(gz2r, sv["rois"])                                  # [B, 128V, 7, 32]
    gh3 = gz2b.view(B, 128 * V * N_SEG, 2 * ROI_BINS)
    gh2 = block_bwd(sv["blk_c22"], gh3, P, grads, side=side)
    gh2q = ops.convt2_deinterleave(gh2)                  # shared by the data and the weight gradient
    gwt, gbt = side.run(lambda: ops.convt2_bwd_weight(sv["h1"], gh2, N_SEG * V, gyq=gh2q), sv["h1"], gh2, gh2q)
    grads["z2_conv2.1.weight"], grads["z2_conv2.1.bias"] = gwt, gbt
    gh1 = ops.convt2_bwd_data(gh2, P["z2_conv2.1.weight"], N_SEG * V, gyq=gh2q)
    gh0 = block_bwd(sv["blk_c20"], gh1, P, grads, side=side)
    genc = torch.empty(B, 128 * V, T, device=gz1.device, dtype=torch.float32)
    # z1_conv / z2_conv1 read the ReLU output of w_conv: they mask their input gradient with it, w_conv skips its gate
    block_bwd(sv["blk_z1"], gz1, P, grads, out=GV.half(genc, V, 0), side=side, pre_gated=z1_pre_gated, gate_input=True)
    win = sv["z2_win"]
    if win is not None:
        gz2c = ops.roi_align_bwd(gh0.view(B, 128 * V, N_SEG, ROI_BINS), sv["rois"], T, win[1], win[0])
        gxw = block_bwd(sv["blk_z2c"], gz2c, P, grads, side=side, gate_input=True)   # [B, 64V, 6]
        ops.window_scatter(gxw, GV.half(genc, V, 1), win[0])
    else:
        gz2c = ops.roi_align_bwd(gh0.view(B, 128 * V, N_SEG, ROI_BINS), sv["rois"], T)
        block_bwd(sv["blk_z2c"], gz2c, P, grads, out=GV.half(genc, V, 1), side=side, gate_input=True)
    gew = block_bwd(sv["blk_w_conv"], genc, P, grads, side=side, pre_gated=True, scale_bwd=_FOLD_CHSCALE_BWD)
    if isinstance(gew, tuple):      # the scaling's backward came out of the block's last launch (block_fwd(..., in_scale=...) + scale_bwd)
        g, ge = gew
    else:
        g, ge = ops.chscale_bwd(gew, sv["w"], sv["e"], relu_x=True)      # sv["w"]: ReLU output of the last encoder block
    gW1, gb1 = side.run(lambda: ops.theta_mlp_bwd(sv["in_theta"], ge, 128), ge)
    grads["mlp1.weight"], grads["mlp1.bias"] = gW1, gb1
    if early:      # data parallel: everything but the encoder blocks' gradients is final -- start summing it across ranks now
        if EARLY_HOOK is not None:      # graph.GraphedTrainStep: the capture is cut in two here, the collective runs between the replays
            EARLY_HOOK(P, grads, side)
        else:
            from . import parallel
            parallel.early_reduce(P, grads, getattr(side, "stream", None))
    for i in (2, 1, 0):
        g = block_bwd(sv["blk_enc"][i], g, P, grads, side=side, pre_gated=True, gate_input=(i > 0))
    grads["W_encoder.conv1.weight"] = ops.stem_bwd_weight(sv["x"], P["W_encoder.conv1.weight"], g)


def backward(P, sv, g_outs):
    """g_outs: gradients wrt (out, shuffle_p, shuffle_l), each [B,1,L] or None.  Returns {param name: grad}."""
    grads = {}
    side = _side(sv["z1"].device, sv["z1"].numel())
    gz1, gz2r = _head_bwd(P, sv, g_outs, grads, side, relu_z1=True)
    _latents_bwd(P, sv, gz1, gz2r, grads, side, z1_pre_gated=True, early=True)
    side.join()
    return grads


def backward2(P, sv, g_outs):
    """Backward of forward2: the head as in Model_nefnet, then the two shared single convs lead by lead (their weight
    gradients summed over the leads in lead order), then the folded-batch encoder."""
    grads = {}
    side = _side(sv["z1"].device, sv["z1"].numel())
    gZ1, gZ2 = _head_bwd(P, sv, g_outs, grads, side)                              # [B, 128V, T]
    B, V = sv["fold"]
    T = gZ1.shape[2]
    gz1f = torch.empty(V * B, 128, T, device=gZ1.device, dtype=torch.float32)
    gz2rf = torch.empty_like(gz1f)
    for name, gZ, xin, gout in (("single_conv_z1.0", gZ1, sv["z1f"], gz1f), ("single_conv_z2.0", gZ2, sv["z2rf"], gz2rf)):
        wf = ops.pack_weight(P[name + ".weight"], 1, flip=True, T=T)
        gw = None
        for i in range(V):
            gv = _lead_view(gZ, i, V)
            ops.conv(gv, wf, 128, 3, out=GV.dense(gout[i * B:(i + 1) * B], 1), role="conv_bwd_data")
            w = ops.conv_bwd_weight(GV.dense(xin[i * B:(i + 1) * B], 1), gv, 3)
            gw = w if gw is None else ops.add(gw, w)
        cs = ops.chan_sum(gZ)                                                      # [128V]
        gb = cs[0:128].contiguous()
        for i in range(1, V):
            gb = ops.add(gb, cs[i * 128:(i + 1) * 128].contiguous())
        grads[name + ".weight"], grads[name + ".bias"] = gw, gb
    _latents_bwd(P, sv, gz1f, gz2rf, grads, side)
    side.join()
    return grads
