"""Per-kernel parity: every HIP op, called through the C ABI (ops.py -> ctypes), against the CPU oracle's
torch-fp32 restatement of the same reference op on the same seeded inputs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from util import FWD_TOL, GRAD_TOL, maxabs, rel, rnd

pytestmark = pytest.mark.gpu

DEV = "cuda"


def ops():
    from electrocardio_panorama_amd import ops as o
    return o


def g(t):
    return t.to(DEV).contiguous()


@pytest.fixture(autouse=True, params=["h2", "wino"])
def conv_algo(request):
    """Every test of this file runs twice: with the split-fp16 direct convs (the default, ops.H2) wherever their shape rules
    apply, and with them off -- the fp32 Winograd / direct kernels that remain the path for every other shape and for NEF_H2=0."""
    o = ops()
    saved = o.H2
    o.H2 = request.param == "h2"
    o.BATCH_HINT = None          # no engine pass is announced here: the shape rules alone pick the kernel
    yield request.param
    o.H2 = saved


def fwd_form(o, K, Cig, Cog, T, f4):
    """The conv args `wino` value pack_weight picks for a forward / backward-data launch of this shape."""
    if o.h2_ok(K, Cig, Cog, T):
        return 3
    return (o.WINO_FWD if f4 else 1) if o.wino_ok(K, Cig, Cog, T) else 0


def packed_floats(o, form, K, G, Cog, Cig, flip=False):
    return o._packed_floats(form, K, G, Cog, Cig, flip)


@pytest.mark.parametrize("B,V,L", [(2, 3, 512), (2, 1, 1000), (3, 2, 520), (1, 8, 256)])
def test_stem(B, V, L):
    o = ops()
    x, w = rnd(B, V, L, seed=1).abs(), rnd(128 * V, 1, 15, seed=2, scale=0.3)
    wr = w.clone().requires_grad_(True)
    ref = F.max_pool1d(F.relu(F.conv1d(x, wr, None, 2, 7, 1, V)), 3, 2, 1)
    y = o.stem_fwd(g(x), g(w))
    assert rel(y, ref) < FWD_TOL
    gy = rnd(*ref.shape, seed=3)
    ref.backward(gy)
    gw = o.stem_bwd_weight(g(x), g(w), g(gy))
    assert rel(gw, wr.grad) < GRAD_TOL


def test_pack_weight():
    o = ops()
    G, Cog, Cig, K = 3, 128, 64, 3
    w = rnd(G * Cog, Cig, K, seed=4)
    wp = o.pack_weight(g(w), G).cpu().view(G, K, Cig, Cog)
    exp = w.view(G, Cog, Cig, K).permute(0, 3, 2, 1)
    assert torch.equal(wp, exp.contiguous())
    wf = o.pack_weight(g(w), G, flip=True).cpu().view(G, K, Cog, Cig)
    expf = w.view(G, Cog, Cig, K).flip(3).permute(0, 3, 1, 2)
    assert torch.equal(wf, expf.contiguous())


def test_pack_many_equals_single_packs():
    """One launch for a whole pass's operands (nef_pack_weights): every pre-packed operand equals the single-call pack,
    is handed out exactly once, and anything that was not requested still packs on demand."""
    o = ops()
    ws = [(g(rnd(3 * 128, 128, 7, seed=40)), 3, False, 1250), (g(rnd(3 * 128, 64, 3, seed=41)), 3, True, 1250),
          (g(rnd(3 * 128, 64, 1, seed=42)), 3, False, None), (g(rnd(21 * 128, 128, 3, seed=43)), 21, True, 16),
          (g(rnd(64, 128, 3, seed=44)), 1, False, 5000), (g(rnd(128, 128, 7, seed=45)), 1, True, 300)]
    want = [o.pack_weight(w, G, flip=f, T=T) for w, G, f, T in ws]
    o.pack_many(ws + ws[:2])                         # duplicates are packed once
    assert len(o._PREPACKED) == len(ws)
    for (w, G, f, T), ref in zip(ws, want):
        got = o.pack_weight(w, G, flip=f, T=T)
        assert torch.equal(got, ref) and getattr(got, "nef_wino", False) == getattr(ref, "nef_wino", False)
    assert not o._PREPACKED                          # consumed
    other = g(rnd(128, 128, 3, seed=46))
    assert torch.equal(o.pack_weight(other, 1, T=256), o.pack_weight(other, 1, T=256))
    o.pack_many([])


CONV_CASES = [  # K, G, Cig, Cog, T, B
    (7, 3, 128, 128, 128, 2), (7, 1, 128, 128, 300, 2), (7, 2, 128, 128, 125, 2),
    (3, 3, 128, 128, 125, 3), (3, 3, 64, 128, 130, 2), (1, 3, 64, 128, 125, 2), (1, 2, 128, 64, 70, 2),
    (3, 21, 128, 128, 16, 5), (3, 21, 64, 128, 32, 5), (1, 21, 64, 128, 32, 3),
    (3, 1, 256, 128, 250, 3), (3, 1, 128, 128, 250, 2), (3, 1, 128, 64, 500, 2), (3, 1, 64, 64, 500, 2),
    (3, 1, 128, 256, 250, 2), (3, 7, 128, 128, 20, 9),
]


@pytest.mark.parametrize("K,G,Cig,Cog,T,B", CONV_CASES)
def test_conv_fwd_bwd(K, G, Cig, Cog, T, B):
    o = ops()
    from electrocardio_panorama_amd.ops import GV
    x = rnd(B, G * Cig, T, seed=5)
    w = rnd(G * Cog, Cig, K, seed=6, scale=(Cig * K) ** -0.5)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = F.conv1d(xr, wr, None, 1, K // 2, 1, G)
    xd, wd = g(x), g(w)
    y = o.conv(GV.dense(xd, G), o.pack_weight(wd, G), Cog, K)
    assert rel(y, ref) < FWD_TOL, "forward"
    gy = rnd(*ref.shape, seed=7)
    ref.backward(gy)
    gyd = g(gy)
    gx = o.conv(GV.dense(gyd, G), o.pack_weight(wd, G, flip=True), Cig, K)
    assert rel(gx, xr.grad) < GRAD_TOL, "bwd-data"
    gw = o.conv_bwd_weight(GV.dense(xd, G), GV.dense(gyd, G), K)
    assert rel(gw, wr.grad) < GRAD_TOL, "bwd-weight"


def test_conv_epilogue_and_views():
    o = ops()
    from electrocardio_panorama_amd.ops import GV
    B, V, T, K = 2, 3, 141, 3
    enc = rnd(B, 128 * V, T, seed=8)
    w = rnd(128 * V, 64, K, seed=9, scale=0.1)
    bias, res = rnd(128 * V, seed=10), rnd(B, 128 * V, T, seed=11)
    gate = rnd(B, 128 * V, T, seed=12)
    mask = (rnd(B, 128 * V, T, seed=13) > -0.6).to(torch.uint8)
    scale = rnd(B, 128 * V, seed=14)
    for which in (0, 1):
        xin = enc.view(B, V, 2, 64, T)[:, :, which].reshape(B, 64 * V, T)
        sc = scale.view(B, V, 2, 64)[:, :, which].reshape(B, 64 * V)
        ref = F.conv1d(xin * sc[:, :, None], w, bias, 1, 1, 1, V) + res
        ref = F.relu(ref) * mask / 0.8
        ref = torch.where(gate > 0, ref * 1.25, torch.zeros_like(ref))
        encd, scd = g(enc), g(scale)
        y = o.conv(GV.half(encd, V, which), o.pack_weight(g(w), V), 128, K, bias=g(bias),
                   in_scale=(scd.view(-1)[which * 64:], 128 * V, 128), res=GV.dense(g(res), V), gate=GV.dense(g(gate), V),
                   gate_scale=1.25, relu=True, mask=g(mask), drop_scale=1.25)
        assert rel(y, ref) < FWD_TOL
        # bwd-weight through the strided view + in_scale
        gy = rnd(B, 128 * V, T, seed=15)
        wr = w.clone().requires_grad_(True)
        F.conv1d(xin * sc[:, :, None], wr, None, 1, 1, 1, V).backward(gy)
        gw = o.conv_bwd_weight(GV.half(encd, V, which), GV.dense(g(gy), V), K,
                               in_scale=(scd.view(-1)[which * 64:], 128 * V, 128))
        assert rel(gw, wr.grad) < GRAD_TOL
        # bwd-data written into one half of a full gradient tensor
        full = torch.full((B, 128 * V, T), 7.0, device=DEV)
        xr = xin.clone().requires_grad_(True)
        F.conv1d(xr, w, None, 1, 1, 1, V).backward(gy)
        o.conv(GV.dense(g(gy), V), o.pack_weight(g(w), V, flip=True), 64, K, out=GV.half(full, V, which))
        got = full.cpu().view(B, V, 2, 64, T)
        assert rel(got[:, :, which].reshape(B, 64 * V, T), xr.grad) < GRAD_TOL
        assert torch.all(got[:, :, 1 - which] == 7.0)


@pytest.mark.parametrize("mode", [1, 2, 3])
@pytest.mark.parametrize("Cig,Cog,T_out", [(256, 128, 250), (128, 64, 500), (64, 64, 260), (128, 128, 128)])
def test_conv_prologue_modes(mode, Cig, Cog, T_out):
    """Decoder fusion: BatchNorm affine + ReLU (bit0) and x2 linear upsampling (bit1) applied while the conv stages
    its input, forward and backward-weight, three passes with their own affine."""
    o = ops()
    from electrocardio_panorama_amd.ops import GV
    P, Bp = 3, 2
    N = P * Bp
    Tin = T_out // 2 if mode & 2 else T_out
    x = rnd(N, Cig, Tin, seed=80)
    a, b = rnd(P, Cig, seed=81) + 1.2, rnd(P, Cig, seed=82) * 0.5
    w, bias = rnd(Cog, Cig, 3, seed=83, scale=(3 * Cig) ** -0.5), rnd(Cog, seed=84)
    xin = x
    if mode & 1:
        xin = F.relu(x * a.repeat_interleave(Bp, 0)[:, :, None] + b.repeat_interleave(Bp, 0)[:, :, None])
    if mode & 2:
        xin = F.interpolate(xin, scale_factor=2, mode="linear", align_corners=False)
    wr = w.clone().requires_grad_(True)
    ref = F.conv1d(xin, wr, bias, 1, 1)
    pro = (mode, g(a) if mode & 1 else None, g(b) if mode & 1 else None, Bp)
    xd = g(x)
    y = o.conv(GV.dense(xd, 1), o.pack_weight(g(w), 1), Cog, 3, bias=g(bias), pro=pro)
    assert y.shape == ref.shape and rel(y, ref) < FWD_TOL
    gy = rnd(*ref.shape, seed=85)
    ref.backward(gy)
    gw = o.conv_bwd_weight(GV.dense(xd, 1), GV.dense(g(gy), 1), 3, pro=pro)
    assert rel(gw, wr.grad) < GRAD_TOL


def test_outconv_prologue():
    o = ops()
    P, Bp, C, L = 3, 2, 64, 500
    x, w, b = rnd(P * Bp, C, L, seed=86), rnd(1, C, 3, seed=87, scale=0.2), rnd(1, seed=88)
    a, bb = rnd(P, C, seed=89) + 1.1, rnd(P, C, seed=90) * 0.4
    act = F.relu(x * a.repeat_interleave(Bp, 0)[:, :, None] + bb.repeat_interleave(Bp, 0)[:, :, None])
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = torch.sigmoid(F.conv1d(act, wr, br, 1, 1) / 3)
    out = o.outconv_fwd(g(x), g(w), g(b), pro=(g(a), g(bb), Bp))
    assert rel(out, ref) < 1e-6
    gy = rnd(*ref.shape, seed=91)
    ref.backward(gy)
    gw, gb = o.outconv_bwd_weight(g(gy), out, g(x), pro=(g(a), g(bb), Bp))
    assert rel(gw, wr.grad) < 1e-5 and rel(gb, br.grad) < 1e-5


def test_conv_dropout_rng():
    o = ops()
    from electrocardio_panorama_amd.ops import GV
    B, G, C, T = 4, 1, 128, 512
    x = torch.ones(B, C, T, device=DEV)
    w = torch.zeros(C, C, 1)
    w[torch.arange(C), torch.arange(C), 0] = 1.0
    wp = o.pack_weight(g(w), G)
    y1 = o.conv(GV.dense(x, G), wp, C, 1, relu=True, drop_p=0.2, drop_scale=1.25, seed=99)
    y2 = o.conv(GV.dense(x, G), wp, C, 1, relu=True, drop_p=0.2, drop_scale=1.25, seed=99)
    y3 = o.conv(GV.dense(x, G), wp, C, 1, relu=True, drop_p=0.2, drop_scale=1.25, seed=100)
    assert torch.equal(y1, y2) and not torch.equal(y1, y3)
    keep = (y1 > 0).float().mean().item()
    assert abs(keep - 0.8) < 0.01
    assert set(torch.unique(y1).cpu().tolist()) == {0.0, 1.25}


@pytest.mark.parametrize("shape", [(5, 37, 333), (40, 21, 32), (3, 8, 6), (33, 5, 1), (2, 3, 1250)])
def test_chan_sum(shape):
    o = ops()
    x = rnd(*shape, seed=16)
    assert rel(o.chan_sum(g(x)), x.double().sum(dim=(0, 2))) < 1e-6


@pytest.mark.parametrize("fma", [False, True])      # matrix-core path (1x1 conv + interleave) and the plain-FMA kernel
@pytest.mark.parametrize("B,G,T", [(3, 7, 16), (5, 21, 16), (2, 56, 16)])
def test_convt2(B, G, T, fma):
    o = ops()
    x, w, b = rnd(B, G * 128, T, seed=17), rnd(G * 128, 64, 2, seed=18, scale=0.1), rnd(G * 64, seed=19)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    ref = F.conv_transpose1d(xr, wr, br, 2, 0, 0, G)
    y = o.convt2_fwd(g(x), g(w), g(b), G, fma=fma)
    assert rel(y, ref) < FWD_TOL
    gy = rnd(*ref.shape, seed=20)
    ref.backward(gy)
    assert rel(o.convt2_bwd_data(g(gy), g(w), G, fma=fma), xr.grad) < GRAD_TOL
    gw, gb = o.convt2_bwd_weight(g(x), g(gy), G, fma=fma)
    assert gw.shape == wr.shape and rel(gw, wr.grad) < GRAD_TOL and rel(gb, br.grad) < GRAD_TOL


def test_theta(golden_dir):
    o = ops()
    from oracle import nefnet_oracle as orc
    z = np.load(f"{golden_dir}/theta_table.npz")
    enc = o.theta_encode(g(torch.from_numpy(z["theta"])))
    assert maxabs(enc, z["enc"]) < 2e-6
    th = rnd(6, 3, 2, seed=21, scale=3.0)
    W, b = rnd(128, 12, seed=22), rnd(128, seed=23)
    Wr, br = W.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.linear(orc.angular_encoding(th), Wr, br)
    y = o.theta_mlp_fwd(g(th), g(W), g(b))
    assert y.shape == ref.shape and rel(y, ref) < FWD_TOL
    gy = rnd(*ref.shape, seed=24)
    ref.backward(gy)
    gW, gb = o.theta_mlp_bwd(g(th), g(gy), 128)
    assert rel(gW, Wr.grad) < GRAD_TOL and rel(gb, br.grad) < GRAD_TOL


def test_chscale_gate_add():
    o = ops()
    x, s, gy = rnd(3, 50, 77, seed=25), rnd(3, 50, seed=26), rnd(3, 50, 77, seed=27)
    xr, sr = x.clone().requires_grad_(True), s.clone().requires_grad_(True)
    ref = xr * sr[:, :, None]
    assert rel(o.chscale_fwd(g(x), g(s)), ref) < 1e-6
    ref.backward(gy)
    gx, gs = o.chscale_bwd(g(gy), g(x), g(s))
    assert rel(gx, xr.grad) < 1e-6 and rel(gs, sr.grad) < 1e-5
    assert rel(o.gate(g(gy), g(x), 1.25), torch.where(x > 0, gy * 1.25, torch.zeros_like(gy))) < 1e-7
    assert rel(o.add(g(x), g(gy)), x + gy) < 1e-7


def _roi_cases(golden_dir):
    z = np.load(f"{golden_dir}/roi_cases.npz")
    names = sorted({k.split(":")[0] for k in z.files})
    return z, names


def test_roi_golden(golden_dir):
    """ROI bookkeeping bit-exact, resampled values to fp32 round-off, against the reference's own outputs."""
    o = ops()
    from oracle import hashweights as hw
    z, names = _roi_cases(golden_dir)
    for name in names:
        L = int(z[f"{name}:L"])
        rois = torch.from_numpy(z[f"{name}:rois"])
        Bn, T, C = rois.shape[0], L // 4, 6
        zz = torch.from_numpy(hw.unit_noise("roi-z:" + name, Bn * C * T).reshape(Bn, C, T).astype(np.float32))
        zs = torch.from_numpy(hw.unit_noise("roi-s:" + name, Bn * C * 7 * 32).reshape(Bn, C, 7, 32).astype(np.float32))
        start, length = o.roi_segment_table(g(rois))
        assert np.array_equal(start.cpu().numpy(), z[f"{name}:seg_start"]), name
        assert np.array_equal(length.cpu().numpy(), z[f"{name}:seg_len"]), name
        assert rel(o.roi_align_fwd(g(zz), g(rois)), z[f"{name}:align"]) < 1e-6, name
        status = torch.zeros(1, dtype=torch.int32, device=DEV)
        assert rel(o.roi_unpool_fwd(g(zs), g(rois), T, status), z[f"{name}:unpool"]) < 1e-6, name
        assert int(status.item()) == 0


def test_roi_backward_and_status():
    o = ops()
    from oracle import nefnet_oracle as orc
    from electrocardio_panorama_amd import synth
    rng = np.random.default_rng(3)
    for L in (512, 1000, 5000, 20000):        # 20000: rows longer than the transpose's LDS strip (in-place variant)
        B, C, T = 3, 5, L // 4
        rois = torch.from_numpy(synth.make_rois(rng, B, L))
        z = rnd(B, C, T, seed=28).requires_grad_(True)
        ref = orc.roi_align_mid(z, rois)
        gy = rnd(*ref.shape, seed=29)
        ref.backward(gy)
        assert rel(o.roi_align_fwd(g(z.detach()), g(rois)), ref) < 1e-6
        assert rel(o.roi_align_bwd(g(gy), g(rois), T), z.grad) < 1e-5
        zs = rnd(B, C, 7, 32, seed=30).requires_grad_(True)
        ref = orc.roi_unpool(zs, rois)
        gy = rnd(*ref.shape, seed=31)
        ref.backward(gy)
        assert rel(o.roi_unpool_fwd(g(zs.detach()), g(rois), T), ref) < 1e-6
        assert rel(o.roi_unpool_bwd(g(gy), g(rois)), zs.grad) < 1e-5
    # windowed storage of z / gz and the crop / scatter pair
    from electrocardio_panorama_amd.ops import GV
    L, B, C = 1000, 3, 6
    T = L // 4
    rois = torch.from_numpy(synth.make_rois(rng, B, L))
    z = rnd(B, 2 * C, T, seed=70)
    t0, W = (T - 1) // 2 - 2, 6
    half = z.view(B, 2, C, T)[:, 1]
    zw = o.window_crop(GV(g(z), B, 2, C, T, 2 * C * T, C * T, 0), t0, W)
    assert torch.equal(zw.cpu(), z[:, :, t0:t0 + W])
    assert rel(o.roi_align_fwd(zw, g(rois), T, t0), orc.roi_align_mid(z, rois)) < 1e-6
    gy = rnd(B, 2 * C, 7, 16, seed=71)
    full = o.roi_align_bwd(g(gy), g(rois), T)
    part = o.roi_align_bwd(g(gy), g(rois), T, W, t0)
    assert torch.equal(full[:, :, t0:t0 + W], part)
    outside = full.clone()
    outside[:, :, t0:t0 + W] = 0
    assert float(outside.abs().max()) == 0.0
    dst = torch.full((B, 2 * (2 * C), T), 7.0, device=DEV)                      # two halves per group, like the z split
    o.window_scatter(part, GV(dst, B, 2, C, T, 4 * C * T, 2 * C * T, C * T), t0)
    got = dst.cpu().view(B, 2, 2, C, T)
    assert torch.all(got[:, :, 0] == 7.0)
    assert torch.equal(got[:, :, 1, :, t0:t0 + W].reshape(B, 2 * C, W), part.cpu())
    rest = got[:, :, 1].clone()
    rest[..., t0:t0 + W] = 0
    assert float(rest.abs().max()) == 0.0
    bad = torch.tensor([[[0, 100], [100, 90], [90, 200], [200, 300], [300, 400], [400, 450], [450, 500]]])
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    o.roi_unpool_fwd(g(rnd(1, 2, 7, 32)), g(bad), 128, status)
    assert int(status.item()) == 1


@pytest.mark.parametrize("V", [1, 3, 8])
def test_mix(V):
    o = ops()
    from oracle import nefnet_oracle as orc
    B, T, c1, c2 = 3, 70, V - 1, 0
    z1, z2r, q = (rnd(B, 128 * V, T, seed=32).requires_grad_(True), rnd(B, 128 * V, T, seed=33).requires_grad_(True),
                  rnd(B, 256, seed=34).requires_grad_(True))
    z1m, z2m = orc.lead_mean(z1, V), orc.lead_mean(z2r, V)
    latent = torch.cat([z1m, z2m], 1)
    D = torch.cat([q[:, :, None] * latent,
                   q[:, :, None] * torch.cat([z1[:, 128 * c1:128 * (c1 + 1)], z2m], 1),
                   q[:, :, None] * torch.cat([z1m, z2r[:, 128 * c2:128 * (c2 + 1)]], 1)], 0)
    lat_d = o.lead_mean(g(z1.detach()), g(z2r.detach()), V)
    assert rel(lat_d, latent) < 1e-6
    Dd = o.mix_fwd(lat_d, g(z1.detach()), g(z2r.detach()), g(q.detach()), V, c1, c2)
    assert rel(Dd, D) < 1e-6
    gD = rnd(*D.shape, seed=35)
    D.backward(gD)
    gz1, gz2r, gq = o.mix_bwd(g(gD), lat_d, g(z1.detach()), g(z2r.detach()), g(q.detach()), V, c1, c2)
    assert rel(gz1, z1.grad) < 1e-5 and rel(gz2r, z2r.grad) < 1e-5 and rel(gq, q.grad) < 1e-5


def test_upsample2():
    o = ops()
    for T in (1, 2, 7, 125, 300):
        x = rnd(2, 5, T, seed=36).requires_grad_(True)
        ref = F.interpolate(x, scale_factor=2, mode="linear", align_corners=False)
        assert rel(o.upsample2_fwd(g(x.detach())), ref) < 1e-6
        gy = rnd(*ref.shape, seed=37)
        ref.backward(gy)
        assert rel(o.upsample2_bwd(g(gy)), x.grad) < 1e-6


def test_batchnorm_train_three_passes():
    o = ops()
    P, Bp, C, L = 3, 4, 16, 250
    x = rnd(P * Bp, C, L, seed=38) * 2 + 0.3
    gamma, beta = rnd(C, seed=39) + 1.5, rnd(C, seed=40)
    rm, rv = rnd(C, seed=41) * 0.1, rnd(C, seed=42).abs() + 0.5
    xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    ys = [F.relu(F.batch_norm(xr[p * Bp:(p + 1) * Bp], rm_ref, rv_ref, gr, br, True, 0.1, 1e-5)) for p in range(P)]
    ref = torch.cat(ys, 0)
    rmd, rvd = g(rm), g(rv)
    mean, invstd, a, b = o.bn_train_stats(g(x), g(gamma), g(beta), rmd, rvd, P)
    y = o.affine_relu_fwd(g(x), a, b, P)
    assert rel(y, ref) < FWD_TOL
    assert rel(rmd, rm_ref) < 1e-6 and rel(rvd, rv_ref) < 1e-6
    gy = rnd(*ref.shape, seed=43)
    ref.backward(gy)
    gx, gg, gb, gs = o.bn_relu_bwd(g(gy), g(x), g(gamma), mean, invstd, a, b, P, with_chan_sum=True)
    assert rel(gx, xr.grad) < 1e-5 and rel(gg, gr.grad) < 1e-5 and rel(gb, br.grad) < 1e-5
    # analytically zero per pass and channel: both sides are fp32 round-off of a sum of O(1) terms
    assert maxabs(gs, xr.grad.double().sum(dim=(0, 2))) < 2e-7 * float(xr.grad.abs().sum(dim=(0, 2)).max())
    # eval affine
    a1, b1 = o.bn_eval_affine(g(gamma), g(beta), rmd, rvd)
    ye = o.affine_relu_fwd(g(x), a1, b1, 1)
    assert rel(ye, F.relu(F.batch_norm(x, rm_ref, rv_ref, gamma, beta, False, 0.1, 1e-5))) < FWD_TOL


def test_outconv():
    o = ops()
    N, C, L = 3, 64, 500
    x, w, b = rnd(N, C, L, seed=44), rnd(1, C, 3, seed=45, scale=0.2), rnd(1, seed=46)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    ref = torch.sigmoid(F.conv1d(xr, wr, br, 1, 1) / 3)
    out = o.outconv_fwd(g(x), g(w), g(b))
    assert out.shape == ref.shape and rel(out, ref) < 1e-6
    gy = rnd(*ref.shape, seed=47)
    ref.backward(gy)
    assert rel(o.outconv_bwd_data(g(gy), out, g(w), C), xr.grad) < GRAD_TOL
    gw, gb = o.outconv_bwd_weight(g(gy), out, g(x))
    assert rel(gw, wr.grad) < GRAD_TOL and rel(gb, br.grad) < GRAD_TOL


@pytest.mark.parametrize("L", [500, 2048, 12])
def test_bn_relu_bwd_through_last_conv_equals_two_calls(L):
    """nef_bn_relu_bwd_outconv rebuilds the last conv's input gradient from go on the fly: it must give exactly what
    nef_outconv_bwd_data followed by nef_bn_relu_bwd gives (same expressions, same summation order)."""
    o = ops()
    P, Bp, C = 3, 2, 64
    N = P * Bp
    x = rnd(N, C, L, seed=120).to(DEV)
    gamma, beta = (rnd(C, seed=121) + 1.2).to(DEV), rnd(C, seed=122, scale=0.3).to(DEV)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    mean, invstd, a, b = o.bn_train_stats(x, gamma, beta, rm, rv, P)
    w, bias = rnd(1, C, 3, seed=123, scale=0.2).to(DEV), rnd(1, seed=124).to(DEV)
    out = o.outconv_fwd(x, w, bias, pro=(a, b, Bp))
    gout = rnd(N, 1, L, seed=125).to(DEV)
    g = o.outconv_bwd_data(gout, out, w, C)
    ref = o.bn_relu_bwd(g, x, gamma, mean, invstd, a, b, P, with_chan_sum=True)
    got = o.bn_relu_bwd_outconv(gout, out, w, x, mean, invstd, a, b, P)
    for r, q in zip(ref, got):
        assert torch.equal(r, q)


def test_mix_bwd_takes_the_upsampling_adjoint_on_the_fly():
    """mix_bwd(upsampled=True) on the gradient wrt the x2-upsampled decoder input == upsample2_bwd followed by mix_bwd,
    bit for bit (same expressions)."""
    o = ops()
    B, V, T = 3, 3, 130
    latent, z1, z2r = g(rnd(B, 256, T, seed=130)), g(rnd(B, 128 * V, T, seed=131)), g(rnd(B, 128 * V, T, seed=132))
    q = g(rnd(B, 256, seed=133))
    gU = g(rnd(3 * B, 256, 2 * T, seed=134))
    ref = o.mix_bwd(o.upsample2_bwd(gU), latent, z1, z2r, q, V, (2, 0))
    got = o.mix_bwd(gU, latent, z1, z2r, q, V, (2, 0), upsampled=True)
    for r, c in zip(ref, got):
        assert torch.equal(r, c)


@pytest.mark.parametrize("L", [500, 16, 2048])
def test_bn_relu_bwd_takes_the_upsampling_adjoint_on_the_fly(L):
    """nef_bn_relu_bwd_up on the gradient wrt the x2-upsampled activation == upsample2_bwd + bn_relu_bwd, bit for bit."""
    o = ops()
    P, Bp, C = 3, 2, 128
    N = P * Bp
    x = rnd(N, C, L, seed=140).to(DEV)
    gamma, beta = (rnd(C, seed=141) + 1.2).to(DEV), rnd(C, seed=142, scale=0.3).to(DEV)
    mean, invstd, a, b = o.bn_train_stats(x, gamma, beta, torch.zeros(C, device=DEV), torch.ones(C, device=DEV), P)
    gu = rnd(N, C, 2 * L, seed=143).to(DEV)
    ref = o.bn_relu_bwd(o.upsample2_bwd(gu), x, gamma, mean, invstd, a, b, P, with_chan_sum=True)
    got = o.bn_relu_bwd_up(gu, x, mean, invstd, a, b, P)
    for r, q in zip(ref, got):
        assert torch.equal(r, q)


def test_relu_masks_folded_into_producers():
    """chscale_bwd(relu_x) and mix_bwd(relu_z1) == the unmasked call followed by the gate pass, bit for bit."""
    o = ops()
    B, V, T = 3, 3, 130
    x, gy, sc = g(rnd(B, 128 * V, T, seed=150)), g(rnd(B, 128 * V, T, seed=151)), g(rnd(B, 128 * V, seed=152))
    gx, gs = o.chscale_bwd(gy, x, sc)
    gxm, gsm = o.chscale_bwd(gy, x, sc, relu_x=True)
    assert torch.equal(gxm, o.gate(gx, x)) and torch.equal(gsm, gs)
    latent, z1, z2r = g(rnd(B, 256, T, seed=153)), g(rnd(B, 128 * V, T, seed=154)), g(rnd(B, 128 * V, T, seed=155))
    q, gD = g(rnd(B, 256, seed=156)), g(rnd(3 * B, 256, T, seed=157))
    ref = o.mix_bwd(gD, latent, z1, z2r, q, V, (1, 2))
    got = o.mix_bwd(gD, latent, z1, z2r, q, V, (1, 2), relu_z1=True)
    assert torch.equal(got[0], o.gate(ref[0], z1)) and torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2])


def test_shared_first_conv_pieces():
    """mix_fwd_shared / pass_combine_fwd / pass_combine_bwd / mix_bwd_shared_up against their definitions (torch CPU)."""
    o = ops()
    B, V, T, c1, c2 = 3, 3, 66, 2, 0
    lat, z1, z2r, q = rnd(B, 256, T, seed=160), rnd(B, 128 * V, T, seed=161), rnd(B, 128 * V, T, seed=162), rnd(B, 256, seed=163)
    z1r, z2rr, qr, latr = (t.clone().double().requires_grad_(True) for t in (z1, z2r, q, lat))
    pick = torch.cat([z1r[:, 128 * c1:128 * (c1 + 1)], z2rr[:, 128 * c2:128 * (c2 + 1)]], 1)
    D2ref = torch.cat([qr[:, :, None] * latr, qr[:, :, None] * pick], 0)
    D2 = o.mix_fwd_shared(g(lat), g(z1), g(z2r), g(q), V, (c1, c2))
    assert D2.shape == (2 * B, 256, T) and rel(D2, D2ref) < 1e-6
    # the one-pass form of lead_mean + mix_fwd_shared: bit-identical, even and odd T, device-side lead choice
    for Tq in (T, 67, 68, 1250):      # (66 and 1250: T % 4 == 2, the row-pair form on 16-byte accesses)
        z1q, z2q = g(rnd(B, 128 * V, Tq, seed=175)), g(rnd(B, 128 * V, Tq, seed=176))
        lat_ref = o.lead_mean(z1q, z2q, V)
        D2_ref = o.mix_fwd_shared(lat_ref, z1q, z2q, g(q), V, (c1, c2))
        lat_f, D2_f = o.lead_mean_mix_shared(z1q, z2q, g(q), V, (c1, c2))
        assert torch.equal(lat_f, lat_ref) and torch.equal(D2_f, D2_ref)
        cdev = torch.tensor([c1, c2], dtype=torch.int32, device=DEV)
        assert torch.equal(o.lead_mean_mix_shared(z1q, z2q, g(q), V, cdev)[1], D2_ref)
    # backward through the x2 upsampling: latent is (z1 mean | z2r mean) in the real graph; here an independent tensor,
    # so compare the three gradients the kernel produces (gz1, gz2r with the 1/V mean spread, gq)
    gU2 = rnd(2 * B, 256, 2 * T, seed=164)
    up = F.interpolate(D2ref, scale_factor=2, mode="linear", align_corners=False)
    (up * gU2.double()).sum().backward()
    gz1, gz2r, gq = o.mix_bwd_shared_up(g(gU2), g(lat), g(z1), g(z2r), g(q), V, (c1, c2))
    glat = latr.grad                                         # d/d(mean input): spread over the V leads by 1/V
    exp_z1 = z1r.grad + glat[:, :128].repeat(1, V, 1) / V
    exp_z2 = z2rr.grad + glat[:, 128:].repeat(1, V, 1) / V
    assert rel(gz1, exp_z1) < 1e-5 and rel(gz2r, exp_z2) < 1e-5 and rel(gq, qr.grad) < 1e-5
    # the same backward at the latent's own resolution (what the polyphase backward-data pass leaves): T % 4 == 2 runs on row PAIRS
    # with 16-byte accesses when every tensor is 16-byte aligned -- against the 8-byte form (the same operands, 8 bytes off)
    def off8(t):
        buf = torch.empty(t.numel() + 2, device=DEV)
        v = buf[2:].view(t.shape)
        v.copy_(t)
        return v
    for Tq in (66, 1250):
        latq, z1q, z2q = g(rnd(B, 256, Tq, seed=180)), g(rnd(B, 128 * V, Tq, seed=181)), g(rnd(B, 128 * V, Tq, seed=182))
        gD2 = g(rnd(2 * B, 256, Tq, seed=183))
        for relu in (False, True):
            a = o.mix_bwd_shared_up(gD2, latq, z1q, z2q, g(q), V, (c1, c2), relu_z1=relu)
            b = o.mix_bwd_shared_up(off8(gD2), off8(latq), off8(z1q), off8(z2q), g(q), V, (c1, c2), relu_z1=relu)
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and rel(a[2], b[2]) < 1e-6
    # combine: c1[p] = A[ia] + B[ib] + bias and its adjoint
    C, L = 8, 37
    P2, bias, gc1 = rnd(2 * B, 2 * C, L, seed=165), rnd(C, seed=166), rnd(3 * B, C, L, seed=167)
    A, Bh = P2[:, :C], P2[:, C:]
    ref = torch.cat([A[:B] + Bh[:B], A[B:] + Bh[:B], A[:B] + Bh[B:]], 0) + bias[None, :, None]
    c1t = o.pass_combine_fwd(g(P2), g(bias), B)
    assert torch.equal(c1t.cpu(), ref)
    gP2 = o.pass_combine_bwd(g(gc1)).cpu()
    g0, g1, g2 = gc1[:B], gc1[B:2 * B], gc1[2 * B:]
    assert torch.equal(gP2[:B, :C], g0 + g2) and torch.equal(gP2[B:, :C], g1)
    assert torch.equal(gP2[:B, C:], g0 + g1) and torch.equal(gP2[B:, C:], g2)
    # the fused form: same c1 and the BatchNorm statistics nef_bn_train_stats would compute from it
    for Bq, Cq, Lq in ((3, 8, 37), (5, 16, 300)):
        P2 = rnd(2 * Bq, 2 * Cq, Lq, seed=168)
        bias, gamma, beta = rnd(Cq, seed=166), rnd(Cq, seed=169) + 1.2, rnd(Cq, seed=170) * 0.3
        rm0, rv0 = rnd(Cq, seed=171) * 0.1, rnd(Cq, seed=172).abs() + 0.5
        c_ref = o.pass_combine_fwd(g(P2), g(bias), Bq)
        rm1, rv1 = g(rm0), g(rv0)
        want = o.bn_train_stats(c_ref, g(gamma), g(beta), rm1, rv1, 3)
        rm2, rv2 = g(rm0), g(rv0)
        c_f, *got = o.pass_combine_fwd_stats(g(P2), g(bias), Bq, g(gamma), g(beta), rm2, rv2)
        assert torch.equal(c_f, c_ref)
        for a_, b_ in zip(got, want):
            assert rel(a_, b_) < 1e-6
        assert rel(rm2, rm1) < 1e-6 and rel(rv2, rv1) < 1e-6


@pytest.mark.parametrize("B,Cin,Cout,T,pro", [(6, 128, 128, 300, 1), (6, 128, 64, 600, 0), (3, 64, 64, 514, 1),
                                              (6, 256, 128, 128, 2)])
def test_conv_epilogue_bn_slot_sums(B, Cin, Cout, T, pro):
    """The conv epilogue's slot sums (nef_conv_args.stats; the split-fp16 kernel and the F(4,3) kernel) + nef_bn_stats_from_slots against nef_bn_train_stats on
    the conv output: same mean / invstd / affine / running statistics (three passes), ragged last tile included."""
    o = ops()
    from electrocardio_panorama_amd.ops import GV
    if o.WINO_FWD != 2:
        pytest.skip("F(4,3) switched off (NEF_WINOGRAD)")
    Tin = T // 2 if pro & 2 else T
    x, w, bias = g(rnd(B, Cin, Tin, seed=180)), g(rnd(Cout, Cin, 3, seed=181, scale=0.05)), g(rnd(Cout, seed=182))
    gamma, beta = g(rnd(Cout, seed=183) + 1.2), g(rnd(Cout, seed=184, scale=0.3))
    rm0, rv0 = rnd(Cout, seed=185) * 0.1, rnd(Cout, seed=186).abs() + 0.5
    pa, pb = g(rnd(3, Cin, seed=187) * 0.5 + 1.0), g(rnd(3, Cin, seed=188) * 0.2)
    prol = (pro, pa, pb, B // 3) if pro & 1 else (pro, None, None, 1)
    wp = o.pack_weight(w, 1, T=T, f4=True)
    slots = o.conv_stats_buffer(wp, B, 1, Cout, T, x.device)
    assert slots is not None and slots[0].shape == (Cout, B * slots[1], 2)
    slots[0].fill_(float("nan"))                     # every slot must be written
    c = o.conv(GV.dense(x, 1), wp, Cout, 3, bias=bias, pro=prol, stats=slots)
    c_plain = o.conv(GV.dense(x, 1), wp, Cout, 3, bias=bias, pro=prol)
    assert torch.equal(c, c_plain)
    rm1, rv1, rm2, rv2 = g(rm0), g(rv0), g(rm0), g(rv0)
    want = o.bn_train_stats(c, gamma, beta, rm1, rv1, 3)
    got = o.bn_stats_from_slots(slots, gamma, beta, rm2, rv2, 3, B, T)
    for a_, b_ in zip(got, want):
        assert rel(a_, b_) < 1e-6
    assert rel(rm2, rm1) < 1e-6 and rel(rv2, rv1) < 1e-6
    # the other conv kernels (direct fp32, F(2,3)) do not leave slot sums: the boundary says so instead of ignoring the request
    saved, o.H2 = o.H2, False
    assert o.conv_stats_buffer(o.pack_weight(w, 1, T=T), B, 1, Cout, T, x.device) is None
    o.H2 = saved


@pytest.mark.parametrize("B,Cin,Cout,T", [(6, 128, 128, 300), (6, 64, 64, 514), (3, 128, 64, 256)])
def test_conv_epilogue_bn_backward_slot_sums(B, Cin, Cout, T):
    """A backward-data launch on the F(4,3) kernel that also leaves the BatchNorm-backward sums of the layer below
    (nef_conv_args.bnb_*): bn_relu_bwd / bn_relu_bwd_combine3 fed with those slots against their own reduction pass."""
    o = ops()
    from electrocardio_panorama_amd.ops import GV
    if o.WINO_FWD != 2:
        pytest.skip("F(4,3) switched off (NEF_WINOGRAD)")
    gc = g(rnd(B, Cin, T, seed=190))                       # gradient at the upper layer's conv output
    w = g(rnd(Cin, Cout, 3, seed=191, scale=0.05))         # upper conv weight [Cout_upper = Cin here][Cin_upper = Cout]
    c_below = g(rnd(B, Cout, T, seed=192))                 # the lower layer's conv output (BatchNorm input)
    gamma, beta = g(rnd(Cout, seed=193) + 1.2), g(rnd(Cout, seed=194, scale=0.3))
    mean, invstd, a_, b_ = o.bn_train_stats(c_below, gamma, beta, torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV), 3)
    wpf = o.pack_weight(w, 1, flip=True, T=T, f4=True)
    slots = o.conv_stats_buffer(wpf, B, 1, Cout, T, gc.device)
    assert slots is not None
    slots[0].fill_(float("nan"))
    gref = o.conv(GV.dense(gc, 1), wpf, Cout, 3, role="conv_bwd_data")
    gv = o.conv(GV.dense(gc, 1), wpf, Cout, 3, role="conv_bwd_data", bnb=(c_below, mean, invstd, a_, b_, B // 3, slots))
    assert torch.equal(gv, gref)
    want = o.bn_relu_bwd(gv, c_below, gamma, mean, invstd, a_, b_, 3, with_chan_sum=True)
    got = o.bn_relu_bwd(gv, c_below, gamma, mean, invstd, a_, b_, 3, with_chan_sum=True, slots=slots)
    for x_, y_ in zip(got[:3], want[:3]):
        assert rel(x_, y_) < 2e-6
    assert maxabs(got[3], want[3]) < 1e-4 * float(want[0].abs().sum() / Cout) + 1e-6     # a cancelled sum: absolute bar
    want3 = o.bn_relu_bwd_combine3(gv, c_below, mean, invstd, a_, b_)
    got3 = o.bn_relu_bwd_combine3(gv, c_below, mean, invstd, a_, b_, slots=slots)
    for x_, y_ in zip(got3[:3], want3[:3]):
        assert rel(x_, y_) < 2e-6


@pytest.mark.parametrize("B,Cin,Cout,T", [(6, 64, 128, 304), (3, 64, 64, 520), (3, 128, 128, 136)])
def test_conv_epilogue_bn_backward_slot_sums_through_upsampling(B, Cin, Cout, T):
    """The same with a x2 upsampling between the BatchNorm below and this launch's output (bnb_up): the slots carry the
    sums of the upsampling's adjoint, i.e. what bn_relu_bwd_up reduces from (g, x) itself."""
    o = ops()
    from electrocardio_panorama_amd.ops import GV
    if o.WINO_FWD != 2:
        pytest.skip("F(4,3) switched off (NEF_WINOGRAD)")
    gc = g(rnd(B, Cin, T, seed=195))
    w = g(rnd(Cin, Cout, 3, seed=196, scale=0.05))
    c_below = g(rnd(B, Cout, T // 2, seed=197))
    gamma, beta = g(rnd(Cout, seed=198) + 1.2), g(rnd(Cout, seed=199, scale=0.3))
    mean, invstd, a_, b_ = o.bn_train_stats(c_below, gamma, beta, torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV), 3)
    wpf = o.pack_weight(w, 1, flip=True, T=T, f4=True)
    slots = o.conv_stats_buffer(wpf, B, 1, Cout, T, gc.device)
    slots[0].fill_(float("nan"))
    gref = o.conv(GV.dense(gc, 1), wpf, Cout, 3, role="conv_bwd_data")
    gv = o.conv(GV.dense(gc, 1), wpf, Cout, 3, role="conv_bwd_data",
                bnb=(c_below, mean, invstd, a_, b_, B // 3, slots, True))
    assert torch.equal(gv, gref)
    want = o.bn_relu_bwd_up(gv, c_below, mean, invstd, a_, b_, 3)
    got = o.bn_relu_bwd_up(gv, c_below, mean, invstd, a_, b_, 3, slots=slots)
    for x_, y_ in zip(got[:3], want[:3]):
        assert rel(x_, y_) < 2e-6
    assert maxabs(got[3], want[3]) < 1e-4 * float(want[0].abs().sum() / Cout) + 1e-6


@pytest.mark.parametrize("B,G,C,T", [(6, 3, 128, 1250), (3, 1, 64, 300), (5, 2, 128, 130)])
def test_conv_h2_channel_scaled_block_input(B, G, C, T):
    """A residual block on a per-(sample, channel) scaled input without writing the scaled tensor (the theta scaling in front of
    w_conv, codes/network/model_nefnet.py:122-124): in_scale on the first conv, res_scale on the residual of the last one -- against
    the same two launches on the materialised product, and against fp64."""
    o = ops()
    from electrocardio_panorama_amd.ops import GV
    if not o.h2_ok(3, C, C, T):
        pytest.skip("split-fp16 convs switched off")
    x = g(rnd(B, G * C, T, seed=1230))
    e = g(rnd(B, G, C, seed=1231) + 0.3)
    w1, w2 = g(rnd(G * C, C, 3, seed=1232, scale=0.05)), g(rnd(G * C, C, 3, seed=1233, scale=0.05))
    xs = x * e.reshape(B, G * C, 1)
    wp1, wp2 = o.pack_weight(w1, G, T=T, plain=False), o.pack_weight(w2, G, T=T, plain=False)
    if getattr(wp1, "nef_wino", 0) != 3:
        pytest.skip("shape not on the split-fp16 kernel")
    sc = (e, G * C, C)
    h_a = o.conv(GV.dense(xs, G), wp1, C, 3, relu=True, x_scale=2.0 ** 6)
    h_b = o.conv(GV.dense(x, G), wp1, C, 3, relu=True, in_scale=sc, x_scale=2.0 ** 6)
    assert torch.equal(h_a, h_b)                      # x * e is formed identically while staging
    y_a = o.conv(GV.dense(h_a, G), wp2, C, 3, res=GV.dense(xs, G), relu=True, x_scale=2.0 ** 6)
    y_b = o.conv(GV.dense(h_a, G), wp2, C, 3, res=GV.dense(x, G), relu=True, res_scale=sc, x_scale=2.0 ** 6)
    want = torch.relu(F.conv1d(h_a.double(), w2.double(), padding=1, groups=G) + xs.double())
    e_a, e_b = rel(y_a.double(), want), rel(y_b.double(), want)
    assert e_b < max(1.5 * e_a, 5e-7), (e_a, e_b)     # one fused multiply-add instead of a rounded product + an add
    assert maxabs(y_a, y_b) < 1e-5 * float(want.abs().max())
    gw_a = o.conv_bwd_weight(GV.dense(xs, G), GV.dense(h_a, G), 3)
    gw_b = o.conv_bwd_weight(GV.dense(x, G), GV.dense(h_a, G), 3, in_scale=sc)
    assert rel(gw_b, gw_a) < 1e-6
    # ... and the scaling's backward in the epilogue of the block's last backward-data launch: gate by the (ReLU-output) input,
    # row scale e, per-(sample, channel) sums of (gradient x input) through the slots -- against ops.chscale_bwd on the materialised
    # gradient
    xr = torch.relu(x)
    gc, g2 = g(rnd(B, G * C, T, seed=1234, scale=1e-3)), g(rnd(B, G * C, T, seed=1235, scale=1e-3))
    wpf = o.pack_weight(w1, G, flip=True, T=T, plain=False)
    gew = o.conv(GV.dense(gc, G), wpf, C, 3, res=GV.dense(g2, G), role="conv_bwd_data", x_scale=2.0 ** 16)
    want_g, want_s = o.chscale_bwd(gew, xr, e.reshape(B, G * C), relu_x=True)
    slots = o.conv_stats_buffer(wpf, B, G, C, T, x.device)
    slots[0].fill_(float("nan"))
    got_g = o.conv(GV.dense(gc, G), wpf, C, 3, res=GV.dense(g2, G), role="conv_bwd_data", x_scale=2.0 ** 16, gate=GV.dense(xr, G),
                   gate_rowscale=sc, stats=slots, stats_mode=1)
    got_s = o.slots_to_rows(slots, B)
    assert torch.equal(got_g, want_g)
    assert rel(got_s, want_s) < 2e-6


@pytest.mark.parametrize("B,G,Cog,Cig,T,aff", [(6, 1, 64, 128, 1000, True), (3, 1, 64, 128, 520, False), (6, 2, 128, 128, 512, False),
                                                (3, 1, 128, 64, 776, True), (3, 1, 64, 64, 256, True)])
def test_conv_fwd_polyphase_behind_upsampling(B, G, Cog, Cig, T, aff):
    """conv1d(upsample2(relu(bn(x))), w) + bias in polyphase form (ops.conv_poly_fwd: conv args pro_mode 8 / 9 + the row-end
    columns) against fp64 through the reference's own ops (codes/network/model_nefnet.py:102-105), against the upsampling-prologue
    form, and its BatchNorm slot sums against the statistics pass.  fp32-class: the bar is the other form's own distance from fp64."""
    o = ops()
    from electrocardio_panorama_amd.ops import GV
    if not o.poly_fwd_ok(G, Cog, Cig, T):
        pytest.skip("polyphase form switched off / shape outside the split-fp16 kernel")
    x = g(rnd(B, G * Cig, T // 2, seed=1210))
    w = g(rnd(G * Cog, Cig, 3, seed=1211, scale=0.05))
    bias = g(rnd(G * Cog, seed=1212, scale=0.2))
    pa, pb = g(rnd(3, G * Cig, seed=1213) + 1.0), g(rnd(3, G * Cig, seed=1214, scale=0.3))
    pro = (3, pa, pb, B // 3) if aff else (2, None, None, 1)
    x64 = x.double()
    if aff:
        x64 = torch.relu(x64 * pa.double().repeat_interleave(B // 3, 0)[:, :, None] + pb.double().repeat_interleave(B // 3, 0)[:, :, None])
    u = torch.nn.functional.interpolate(x64, scale_factor=2, mode="linear", align_corners=False)
    want = torch.nn.functional.conv1d(u, w.double(), bias.double(), padding=1, groups=G)
    other = o.conv(GV.dense(x, G), o.pack_weight(w, G, T=T, f4=True), Cog, 3, bias=bias, pro=pro)
    got, slots = o.conv_poly_fwd(GV.dense(x, G), w, Cog, bias=bias, pro=pro, stats=True)
    e_o, e_g = rel(other.double(), want), rel(got.double(), want)
    print(f"polyphase forward B={B} G={G} {Cig}->{Cog} T={T} aff={aff}: rel-L2 vs fp64 {e_g:.2e} (upsampling-prologue form {e_o:.2e})")
    assert e_g < max(2.0 * e_o, 1e-6)
    for col in (0, 1, 2, T - 3, T - 2, T - 1):
        assert rel(got[:, :, col].double(), want[:, :, col]) < 2e-6
    assert torch.equal(o.conv_poly_fwd(GV.dense(x, G), w, Cog, bias=bias, pro=pro), got)
    if slots is not None:
        sl, nslot = slots
        ssum = sl.view(G * Cog, B * nslot, 2).double().sum(1)
        assert rel(ssum[:, 0], got.double().sum((0, 2))) < 1e-6
        assert rel(ssum[:, 1], (got.double() ** 2).sum((0, 2))) < 1e-6


@pytest.mark.parametrize("B,G,Cog,Cig,T,aff", [(6, 1, 64, 128, 1000, True), (6, 2, 128, 128, 512, False), (3, 1, 128, 64, 776, True),
                                                (3, 1, 64, 64, 520, False)])
def test_polyphase_backward_on_phase_major_gradients(B, G, Cog, Cig, T, aff):
    """The whole backward of y = conv1d(upsample2(relu(bn(x))), w) at half resolution: the BatchNorm-backward pass above writes
    the gradient phase-major (ops.bn_relu_bwd(phase_major=True): [.., 2C, T/2]), the backward-data pass is a plain conv over it +
    row-end terms, the weight gradient a split-fp16 weight gradient over (half-resolution x with clamped ends, phase rows) folded
    back onto the three taps (nef_poly_wgrad_fold) -- against fp64 autograd through the reference's ops
    (codes/network/model_nefnet.py:102-105) and against the interleaved forms."""
    o = ops()
    from electrocardio_panorama_amd.ops import GV
    if not (o.poly_fwd_ok(G, Cog, Cig, T) and o.poly_bwd_ok(G, Cog, Cig, T) and o.poly_w_ok(B, G, Cog, Cig, T)):
        pytest.skip("polyphase forms switched off / shape outside the split-fp16 kernels")
    x = g(rnd(B, G * Cig, T // 2, seed=1220))
    w = g(rnd(G * Cog, Cig, 3, seed=1221, scale=0.05))
    pa, pb = g(rnd(3, G * Cig, seed=1222) + 1.0), g(rnd(3, G * Cig, seed=1223, scale=0.3))
    pro = (3, pa, pb, B // 3) if aff else (2, None, None, 1)
    # the gradient at the conv output comes out of a BatchNorm-backward pass: run it in both layouts
    gy_in, cc = g(rnd(B, G * Cog, T, seed=1224)), g(rnd(B, G * Cog, T, seed=1225))
    gamma, beta = g(rnd(G * Cog, seed=1226) + 1.2), g(rnd(G * Cog, seed=1227, scale=0.3))
    mean, invstd, a_, b_ = o.bn_train_stats(cc, gamma, beta, torch.zeros(G * Cog, device=DEV), torch.ones(G * Cog, device=DEV), 3)
    gy = o.bn_relu_bwd(gy_in, cc, gamma, mean, invstd, a_, b_, 3)[0]
    gy_pm = o.bn_relu_bwd(gy_in, cc, gamma, mean, invstd, a_, b_, 3, phase_major=True)[0]
    assert gy_pm.shape == (B, 2 * G * Cog, T // 2)
    assert torch.equal(gy_pm.view(B, G * Cog, 2, T // 2).permute(0, 1, 3, 2).reshape(B, G * Cog, T), gy)
    # fp64 truth
    x64 = x.double().requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    xp = x64
    if aff:
        xp = torch.relu(x64 * pa.double().repeat_interleave(B // 3, 0)[:, :, None] + pb.double().repeat_interleave(B // 3, 0)[:, :, None])
    xp.retain_grad()
    u = torch.nn.functional.interpolate(xp, scale_factor=2, mode="linear", align_corners=False)
    torch.nn.functional.conv1d(u, w64, padding=1, groups=G).backward(gy.double())
    y = o.conv_poly_fwd(GV.dense(x, G), w, Cog, pro=pro, save_edge=True)
    gx = o.conv_bwd_data_poly(GV.dense(gy_pm, G), w, Cig, phase_major=True)
    gx_il = o.conv_bwd_data_poly(GV.dense(gy, G), w, Cig)
    gw = o.conv_bwd_weight_poly(GV.dense(x, G), gy_pm, Cog, pro, y.nef_xedge)
    gw_up = o.conv_bwd_weight(GV.dense(x, G), GV.dense(gy, G), 3, pro=pro)
    e_gx, e_il = rel(gx.double(), xp.grad), rel(gx_il.double(), xp.grad)
    e_gw, e_up = rel(gw.double(), w64.grad), rel(gw_up.double(), w64.grad)
    print(f"polyphase backward B={B} G={G} {Cig}->{Cog} T={T} aff={aff}: vs fp64 gx {e_gx:.2e} (interleaved input {e_il:.2e}), "
          f"gw {e_gw:.2e} (upsampling-prologue form {e_up:.2e})")
    assert e_gx < max(2.0 * e_il, 1e-6) and e_gw < max(2.0 * e_up, 1e-6)
    for k in range(3):
        assert rel(gw[:, :, k].double(), w64.grad[:, :, k]) < max(2.0 * rel(gw_up[:, :, k].double(), w64.grad[:, :, k]), 1e-6)


@pytest.mark.parametrize("B,G,Cog,Cig,T", [(6, 1, 64, 128, 1000), (3, 1, 64, 128, 520), (6, 2, 128, 128, 512), (3, 1, 128, 64, 776)])
def test_conv_bwd_data_polyphase_through_upsampling(B, G, Cog, Cig, T):
    """Backward-data through conv1d(upsample2(x), w) at HALF resolution (ops.conv_bwd_data_poly: phase-stacked input, pro_mode 4,
    + the row-end terms) against fp64 autograd through the reference's own ops (codes/network/model_nefnet.py:102-105), against the
    two-pass form (full-resolution backward-data, then the upsampling's adjoint), and its BatchNorm-backward slot sums against
    bn_relu_bwd's own reduction.  fp32-class: the bar is the two-pass form's own distance from fp64."""
    o = ops()
    from electrocardio_panorama_amd.ops import GV
    if not o.poly_bwd_ok(G, Cog, Cig, T):
        pytest.skip("polyphase form switched off / shape outside the split-fp16 kernel")
    gy = g(rnd(B, G * Cog, T, seed=1200))
    w = g(rnd(G * Cog, Cig, 3, seed=1201, scale=0.05))
    x64 = torch.zeros(B, G * Cig, T // 2, dtype=torch.float64, device=DEV, requires_grad=True)
    u = torch.nn.functional.interpolate(x64, scale_factor=2, mode="linear", align_corners=False)
    y = torch.nn.functional.conv1d(u, w.double(), padding=1, groups=G)
    y.backward(gy.double())
    want = x64.grad
    two = o.upsample2_bwd(o.conv(GV.dense(gy, G), o.pack_weight(w, G, flip=True, T=T, f4=True), Cig, 3, role="conv_bwd_data"))
    got = o.conv_bwd_data_poly(GV.dense(gy, G), w, Cig)
    e_two, e_got = rel(two.double(), want), rel(got.double(), want)
    print(f"polyphase backward-data B={B} G={G} {Cog}->{Cig} T={T}: rel-L2 vs fp64 {e_got:.2e} (two-pass form {e_two:.2e})")
    assert e_got < max(2.0 * e_two, 1e-6)
    for col in (0, 1, T // 2 - 2, T // 2 - 1):      # the row ends, where the phase form needs its correction
        assert rel(got[:, :, col].double(), want[:, :, col]) < 2e-6
    if G == 1:
        c_below = g(rnd(B, Cig, T // 2, seed=1202))
        gamma, beta = g(rnd(Cig, seed=1203) + 1.2), g(rnd(Cig, seed=1204, scale=0.3))
        mean, invstd, a_, b_ = o.bn_train_stats(c_below, gamma, beta, torch.zeros(Cig, device=DEV), torch.ones(Cig, device=DEV), 3)
        wsyn = o.pack_weight(o.poly_weights(w), 1, flip=True, T=T // 2)
        slots = o.conv_stats_buffer(wsyn, B, 1, Cig, T // 2, gy.device)
        slots[0].fill_(float("nan"))
        gv = o.conv_bwd_data_poly(GV.dense(gy, 1), w, Cig, bnb=(c_below, mean, invstd, a_, b_, B // 3, slots))
        assert torch.equal(gv, got)
        want_b = o.bn_relu_bwd(gv, c_below, gamma, mean, invstd, a_, b_, 3, with_chan_sum=True)
        got_b = o.bn_relu_bwd(gv, c_below, gamma, mean, invstd, a_, b_, 3, with_chan_sum=True, slots=slots)
        for x_, y_ in zip(got_b[:3], want_b[:3]):
            assert rel(x_, y_) < 2e-6


def test_bn_relu_bwd_combine3_equals_two_calls():
    o = ops()
    Bp, C, L = 2, 16, 301
    N = 3 * Bp
    x, gy = rnd(N, C, L, seed=170).to(DEV), rnd(N, C, L, seed=171).to(DEV)
    gamma, beta = (rnd(C, seed=172) + 1.2).to(DEV), rnd(C, seed=173, scale=0.3).to(DEV)
    mean, invstd, a, b = o.bn_train_stats(x, gamma, beta, torch.zeros(C, device=DEV), torch.ones(C, device=DEV), 3)
    gx, gg, gb, gs = o.bn_relu_bwd(gy, x, gamma, mean, invstd, a, b, 3, with_chan_sum=True)
    got = o.bn_relu_bwd_combine3(gy, x, mean, invstd, a, b)
    assert torch.equal(got[0], o.pass_combine_bwd(gx))
    assert torch.equal(got[1], gg) and torch.equal(got[2], gb) and rel(got[3], gs) < 1e-6


@pytest.mark.parametrize("reg", ["l1_loss", "l2_loss"])
def test_loss(reg):
    o = ops()
    from oracle import nefnet_oracle as orc
    n = (4, 1, 500)
    pred, pp, pl, tgt = (torch.sigmoid(rnd(*n, seed=s)) for s in (48, 49, 50, 51))
    pr, ppr, plr = (t.clone().requires_grad_(True) for t in (pred, pp, pl))
    ref = orc.loss_v1(pr, ppr, plr, tgt, (0.5, 0.5, 1.0), (1, 2, 3), reg)
    L4 = o.loss_fwd(g(pred), g(pp), g(pl), g(tgt), (0.5, 0.5, 1.0), reg == "l2_loss", 7)
    assert maxabs(L4, torch.stack([r.detach() for r in ref])) < 1e-6
    ref[0].backward()
    gs = torch.ones(4, device=DEV)
    g_pred, g_p, g_l = o.loss_bwd(g(pred), g(pp), g(pl), g(tgt), gs, (0.5, 0.5, 1.0), reg == "l2_loss", 7)
    assert rel(g_pred, pr.grad) < 1e-5 and rel(g_p, ppr.grad) < 1e-5 and rel(g_l, plr.grad) < 1e-5


def test_mse_lead_value_and_gradient():
    """Reference losses.py:53-64: mean over leads of nn.MSELoss per lead -- value AND gradient (round-2 advisor: the
    returned element used to be the one no gradient flows through)."""
    from electrocardio_panorama_amd.network.loss.losses import MSELead
    x, tgt = rnd(3, 4, 200, seed=54), rnd(3, 4, 200, seed=55)
    xr = x.clone().requires_grad_(True)
    ref = torch.stack([F.mse_loss(xr[:, i], tgt[:, i]) for i in range(x.size(1))]).mean()
    ref.backward()
    xd = g(x).requires_grad_(True)
    got = MSELead()(xd, g(tgt))
    assert abs(float(got) - float(ref)) < 1e-6 * max(1.0, abs(float(ref)))
    (2.5 * got).backward()
    assert xd.grad is not None and float(xd.grad.abs().max()) > 0
    assert rel(xd.grad, 2.5 * xr.grad) < 1e-5


def test_sgd_momentum():
    o = ops()
    p, buf = rnd(1000, seed=52), torch.zeros(1000)
    pd, bd = g(p), g(buf)
    pr, br = p.clone(), None
    for step in range(3):
        gr = rnd(1000, seed=53 + step)
        o.sgd_momentum(pd, g(gr), bd, 0.1, 0.9, 1.0, False)
        br = gr.clone() if br is None else br * 0.9 + gr
        pr = pr - 0.1 * br
    assert rel(pd, pr) < 1e-6


@pytest.mark.parametrize("prefix,K,G,Cig,T", [("W_encoder.layer1.0", 7, 2, 128, 130), ("w_conv.0", 3, 2, 128, 130),
                                               ("z1_conv.0", 3, 2, 64, 130), ("z2_conv2.2", 3, 14, 64, 32)])
def test_basic_block(prefix, K, G, Cig, T):
    from electrocardio_panorama_amd import engine
    from electrocardio_panorama_amd.ops import GV
    from oracle import nefnet_oracle as orc
    B, Cog = 3, 128
    P = {prefix + ".conv1.weight": rnd(G * Cog, Cig, K, seed=60, scale=(Cig * K) ** -0.5),
         prefix + ".conv2.weight": rnd(G * Cog, Cog, K, seed=61, scale=(Cog * K) ** -0.5),
         prefix + ".residual_conv.weight": rnd(G * Cog, Cig, 1, seed=62, scale=Cig ** -0.5),
         prefix + ".residual_conv.bias": rnd(G * Cog, seed=63)}
    x = rnd(B, G * Cig, T, seed=64)
    mask = (rnd(B, G * Cog, T, seed=65) > -0.6).to(torch.uint8)
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    xr = x.clone().requires_grad_(True)
    ref = orc.res_block(xr, Pr, prefix, G, K, True, {prefix: mask}, 0.2)
    Pd = {k: g(v) for k, v in P.items()}
    drop = engine.DropCfg(True, 0.2, {prefix: g(mask)})
    y, saved = engine.block_fwd(GV.dense(g(x), G), Pd, prefix, K, Cog, drop)
    assert rel(y, ref) < FWD_TOL
    gy = rnd(*ref.shape, seed=66)
    ref.backward(gy)
    grads = {}
    gx = engine.block_bwd(saved, g(gy), Pd, grads)
    assert rel(gx, xr.grad) < GRAD_TOL
    for k, v in grads.items():
        assert rel(v, Pr[k].grad) < GRAD_TOL, k


@pytest.mark.parametrize("T", [125, 24])
def test_decoder_three_passes(T):
    """Three stacked BatchNorm passes (T=125: fused prologue path; T=24: short-sequence unfused path).  Four BN layers make this gradient ill-conditioned on random data (torch's own
    fp32 result sits ~8e-3 from its fp64 result here), so the yardstick is the fp64 oracle and the bar is the fp32
    oracle's own distance from it."""
    from electrocardio_panorama_amd import engine
    from oracle import hashweights as hw
    from oracle import nefnet_oracle as orc
    B = 2
    P = {k: v for k, v in hw.hashed_params(1).items() if k.startswith("decoder.")}
    Bf = hw.hashed_buffers()
    D = rnd(3 * B, 256, T, seed=67)
    gy = rnd(3 * B, 1, 4 * T, seed=68)

    def oracle(dt):
        Pr = {k: v.clone().to(dt).requires_grad_(True) for k, v in P.items()}
        Bfr = {k: (v.clone().to(dt) if v.dtype.is_floating_point else v.clone()) for k, v in Bf.items()}
        Dr = D.clone().to(dt).requires_grad_(True)
        ref = torch.cat([orc.decoder(Dr[p * B:(p + 1) * B], Pr, Bfr, True) for p in range(3)], 0)
        ref.backward(gy.to(dt))
        return ref.detach(), Dr.grad, {k: v.grad for k, v in Pr.items()}, Bfr

    ref, gD32, g32, Bfr = oracle(torch.float32)
    _, gD64, g64, _ = oracle(torch.float64)
    Pd, Bfd = {k: g(v) for k, v in P.items()}, {k: g(v) for k, v in Bf.items()}
    out, dsv = engine.decoder_fwd(g(D), Pd, Bfd, 3, True, True)
    assert rel(out, ref) < FWD_TOL
    for k in Bf:
        assert rel(Bfd[k].float(), Bfr[k].float()) < 1e-5, k
    grads = {}
    gD, up, _ = engine.decoder_bwd(dsv, g(gy), Pd, grads)
    if up:                                   # the first upsampling's adjoint is left to the consumer (mix_bwd)
        gD = ops().upsample2_bwd(gD)
    assert rel(gD, gD64) < GRAD_TOL + 2 * rel(gD32, gD64), (rel(gD, gD64), rel(gD32, gD64))
    for k, v in grads.items():
        if k.endswith("double_conv.0.bias") or k.endswith("double_conv.3.bias"):
            assert maxabs(v, g64[k]) < 1e-5, k          # analytically zero
        else:
            assert rel(v, g64[k]) < GRAD_TOL + 2 * rel(g32[k], g64[k]), (k, rel(v, g64[k]), rel(g32[k], g64[k]))


WINO_CASES = [  # K, G, Cig, Cog, T, B
    (3, 3, 128, 128, 128, 2), (3, 3, 128, 128, 250, 3), (3, 3, 64, 128, 130, 2), (3, 1, 256, 128, 250, 3),
    (3, 2, 128, 128, 1250, 2), (3, 1, 128, 64, 500, 2), (3, 1, 64, 64, 260, 3), (3, 1, 128, 64, 256, 2),
    (3, 1, 64, 128, 5000, 2), (3, 1, 128, 256, 250, 2),
    (7, 3, 128, 128, 128, 2), (7, 1, 128, 128, 300, 2), (7, 2, 128, 128, 1250, 2), (7, 1, 128, 64, 500, 2),
    (7, 3, 128, 128, 130, 3), (7, 1, 64, 128, 256, 2),
]


@pytest.mark.parametrize("f4", [False, True])
@pytest.mark.parametrize("K,G,Cig,Cog,T,B", WINO_CASES)
def test_conv_winograd(K, G, Cig, Cog, T, B, f4):
    """K = 3 and K = 7 (taps split 3 + 3 + 1) through Winograd F(2,3) (conv_wino_kernel): forward and backward-data
    against F.conv1d at the forward / gradient bars, and the distance from exact (fp64) arithmetic next to the direct
    kernel's."""
    o = ops()
    from electrocardio_panorama_amd.ops import GV
    assert o.wino_ok(K, Cig, Cog, T) and o.wino_ok(K, Cog, Cig, T) == (Cig % 128 == 0 or T >= 256)
    x = rnd(B, G * Cig, T, seed=5)
    w = rnd(G * Cog, Cig, K, seed=6, scale=(Cig * K) ** -0.5)
    xr = x.clone().requires_grad_(True)
    ref = F.conv1d(xr, w, None, 1, K // 2, 1, G)
    ref64 = F.conv1d(x.double(), w.double(), None, 1, K // 2, 1, G)
    xd, wd = g(x), g(w)
    wpw = o.pack_weight(wd, G, T=T, f4=f4)
    form = fwd_form(o, K, Cig, Cog, T, f4)
    assert form in (1, 2, 3) and wpw.nef_wino == form and wpw.numel() == packed_floats(o, form, K, G, Cog, Cig)
    y = o.conv(GV.dense(xd, G), wpw, Cog, K)
    yd = o.conv(GV.dense(xd, G), o.pack_weight(wd, G), Cog, K)
    assert rel(y, ref) < FWD_TOL, "forward"
    e_w, e_d, e_t = rel(y, ref64), rel(yd, ref64), rel(ref, ref64)
    assert e_w < 4 * max(e_d, e_t) + 1e-7, (e_w, e_d, e_t)       # the transforms cost a small constant factor, not more
    gy = rnd(*ref.shape, seed=7)
    ref.backward(gy)
    if o.wino_ok(K, Cog, Cig, T):
        wf = o.pack_weight(wd, G, flip=True, T=T, f4=f4)
        assert wf.nef_wino == fwd_form(o, K, Cog, Cig, T, f4)
        gx = o.conv(GV.dense(g(gy), G), wf, Cig, K)
        assert rel(gx, xr.grad) < GRAD_TOL, "bwd-data"


def test_conv_winograd_k7_block_epilogues():
    """The two conv launches of an encoder BasicBlock (resnet_1d.py:42-53) and their backward-data counterparts on the
    K = 7 Winograd path: ReLU + replayed dropout; residual + ReLU; gated backward-data with a residual gradient."""
    o = ops()
    from electrocardio_panorama_amd.ops import GV
    B, V, T, K = 2, 3, 250, 7
    x = rnd(B, 128 * V, T, seed=21)
    w1, w2 = rnd(128 * V, 128, K, seed=22, scale=0.03), rnd(128 * V, 128, K, seed=23, scale=0.03)
    keep = (rnd(B, 128 * V, T, seed=24) > -0.6).to(torch.uint8)
    xr, w1r, w2r = (t.clone().requires_grad_(True) for t in (x, w1, w2))
    h = F.relu(F.conv1d(xr, w1r, None, 1, 3, 1, V)) * keep / 0.8
    y = F.relu(F.conv1d(h, w2r, None, 1, 3, 1, V) + xr)
    gy = rnd(B, 128 * V, T, seed=25)
    y.backward(gy)
    xd, w1d, w2d, kd = g(x), g(w1), g(w2), g(keep)
    hd = o.conv(GV.dense(xd, V), o.pack_weight(w1d, V, T=T), 128, K, relu=True, mask=kd, drop_scale=1.25)
    yd = o.conv(GV.dense(hd, V), o.pack_weight(w2d, V, T=T), 128, K, res=GV.dense(xd, V), relu=True)
    assert rel(hd, h) < FWD_TOL and rel(yd, y) < FWD_TOL
    g2 = o.gate(g(gy), yd)
    gc1 = o.conv(GV.dense(g2, V), o.pack_weight(w2d, V, flip=True, T=T), 128, K, gate=GV.dense(hd, V), gate_scale=1.25)
    gx = o.conv(GV.dense(gc1, V), o.pack_weight(w1d, V, flip=True, T=T), 128, K, res=GV.dense(g2, V))
    assert rel(gx, xr.grad) < GRAD_TOL


@pytest.mark.parametrize("f4", [False, True])
def test_conv_winograd_epilogue_views_and_rng(f4):
    """Every epilogue / operand option of the conv entry point on the Winograd path: bias, residual, ReLU, replayed and
    counter-RNG dropout, gate, in_scale, strided input and output views."""
    o = ops()
    from electrocardio_panorama_amd.ops import GV
    B, V, T, K = 2, 3, 250, 3
    enc = rnd(B, 128 * V, T, seed=8)
    w = rnd(128 * V, 64, K, seed=9, scale=0.1)
    bias, res = rnd(128 * V, seed=10), rnd(B, 128 * V, T, seed=11)
    gate = rnd(B, 128 * V, T, seed=12)
    mask = (rnd(B, 128 * V, T, seed=13) > -0.6).to(torch.uint8)
    scale = rnd(B, 128 * V, seed=14)
    for which in (0, 1):
        xin = enc.view(B, V, 2, 64, T)[:, :, which].reshape(B, 64 * V, T)
        sc = scale.view(B, V, 2, 64)[:, :, which].reshape(B, 64 * V)
        ref = F.conv1d(xin * sc[:, :, None], w, bias, 1, 1, 1, V) + res
        ref = F.relu(ref) * mask / 0.8
        ref = torch.where(gate > 0, ref * 1.25, torch.zeros_like(ref))
        encd, scd = g(enc), g(scale)
        wp = o.pack_weight(g(w), V, T=T, f4=f4)
        assert wp.nef_wino
        y = o.conv(GV.half(encd, V, which), wp, 128, K, bias=g(bias),
                   in_scale=(scd.view(-1)[which * 64:], 128 * V, 128), res=GV.dense(g(res), V), gate=GV.dense(g(gate), V),
                   gate_scale=1.25, relu=True, mask=g(mask), drop_scale=1.25)
        assert rel(y, ref) < FWD_TOL
    # bwd-data (128 -> 64 channels per group needs T >= 256) written into one half of a full gradient tensor
    T2 = 300
    gy = rnd(B, 128 * V, T2, seed=15)
    for which in (0, 1):
        full = torch.full((B, 128 * V, T2), 7.0, device=DEV)
        xr = rnd(B, 64 * V, T2, seed=16).requires_grad_(True)
        F.conv1d(xr, w, None, 1, 1, 1, V).backward(gy)
        wf = o.pack_weight(g(w), V, flip=True, T=T2, f4=f4)
        assert wf.nef_wino
        o.conv(GV.dense(g(gy), V), wf, 64, K, out=GV.half(full, V, which))
        got = full.cpu().view(B, V, 2, 64, T2)
        assert rel(got[:, :, which].reshape(B, 64 * V, T2), xr.grad) < GRAD_TOL
        assert torch.all(got[:, :, 1 - which] == 7.0)
    # counter-RNG dropout: the keep decision is keyed by the dense element index, so both kernels drop the same elements
    xd, wd = g(rnd(B, 128, T, seed=17)), g(rnd(128, 128, 3, seed=18, scale=0.05))
    kw = dict(relu=False, drop_p=0.2, drop_scale=1.25, seed=1234)
    a = o.conv(GV.dense(xd, 1), o.pack_weight(wd, 1, T=T, f4=f4), 128, 3, **kw)
    d = o.conv(GV.dense(xd, 1), o.pack_weight(wd, 1), 128, 3, **kw)
    assert torch.equal(a == 0, d == 0) and 0.15 < float((a == 0).float().mean()) < 0.25 and rel(a, d) < 1e-6


@pytest.mark.parametrize("f4", [False, True])
@pytest.mark.parametrize("mode", [1, 2, 3])
@pytest.mark.parametrize("Cig,Cog,T_out", [(256, 128, 250), (128, 64, 500), (64, 64, 260), (128, 128, 128)])
def test_conv_winograd_prologue_modes(mode, Cig, Cog, T_out, f4):
    o = ops()
    from electrocardio_panorama_amd.ops import GV
    P, Bp = 3, 2
    N = P * Bp
    Tin = T_out // 2 if mode & 2 else T_out
    x = rnd(N, Cig, Tin, seed=80)
    a, b = rnd(P, Cig, seed=81) + 1.2, rnd(P, Cig, seed=82) * 0.5
    w, bias = rnd(Cog, Cig, 3, seed=83, scale=(3 * Cig) ** -0.5), rnd(Cog, seed=84)
    xin = x
    if mode & 1:
        xin = F.relu(x * a.repeat_interleave(Bp, 0)[:, :, None] + b.repeat_interleave(Bp, 0)[:, :, None])
    if mode & 2:
        xin = F.interpolate(xin, scale_factor=2, mode="linear", align_corners=False)
    ref = F.conv1d(xin, w, bias, 1, 1)
    pro = (mode, g(a) if mode & 1 else None, g(b) if mode & 1 else None, Bp)
    wp = o.pack_weight(g(w), 1, T=T_out, f4=f4)
    assert wp.nef_wino
    y = o.conv(GV.dense(g(x), 1), wp, Cog, 3, bias=g(bias), pro=pro)
    assert y.shape == ref.shape and rel(y, ref) < FWD_TOL


@pytest.mark.parametrize("K,G,Cig,Cog,T,B", [
    (3, 3, 128, 128, 128, 2), (3, 3, 64, 128, 130, 2), (3, 1, 256, 128, 250, 3), (3, 1, 128, 64, 500, 2),
    (3, 1, 64, 64, 260, 3), (3, 2, 128, 128, 1250, 2), (3, 1, 32, 128, 64, 3), (3, 1, 128, 256, 66, 2),
    (7, 3, 128, 128, 128, 2), (7, 1, 128, 128, 300, 2), (7, 2, 128, 128, 1250, 2), (7, 1, 128, 64, 500, 2),
    (7, 1, 64, 128, 70, 5),
])
@pytest.mark.parametrize("form", [4])
def test_conv_bwd_weight_winograd(K, G, Cig, Cog, T, B, form):
    """The weight gradient through the transposed Winograd forms -- F(3,4) for K = 3, the 4 + 3 split through F(4,4) + F(3,4) for
    K = 7 -- against autograd, next to the direct kernel, and all against fp64: within 8x (transform entries up to 8 and 1/24)
    of the larger of the direct kernel's and torch-CPU's own distance from exact."""

    o = ops()
    from electrocardio_panorama_amd.ops import GV
    x = rnd(B, G * Cig, T, seed=31)
    w = rnd(G * Cog, Cig, K, seed=32, scale=(Cig * K) ** -0.5)
    gy = rnd(B, G * Cog, T, seed=33)
    wr = w.clone().requires_grad_(True)
    F.conv1d(x, wr, None, 1, K // 2, 1, G).backward(gy)
    w64 = w.double().requires_grad_(True)
    F.conv1d(x.double(), w64, None, 1, K // 2, 1, G).backward(gy.double())
    xd, gyd = g(x), g(gy)
    gw = o.conv_bwd_weight(GV.dense(xd, G), GV.dense(gyd, G), K, wino=form)
    gd = o.conv_bwd_weight(GV.dense(xd, G), GV.dense(gyd, G), K, wino=False)
    assert rel(gw, wr.grad) < GRAD_TOL and rel(gd, wr.grad) < GRAD_TOL
    e_w, e_d, e_t = rel(gw, w64.grad), rel(gd, w64.grad), rel(wr.grad, w64.grad)
    assert e_w < (8 if form == 4 else 4) * max(e_d, e_t) + 1e-7, (e_w, e_d, e_t)
    assert torch.equal(gw, o.conv_bwd_weight(GV.dense(xd, G), GV.dense(gyd, G), K, wino=form))      # deterministic


@pytest.mark.parametrize("form", [4])
def test_conv_bwd_weight_winograd_views_scale_and_prologues(form):
    o = ops()
    from electrocardio_panorama_amd.ops import GV
    B, V, T, K = 2, 3, 250, 3
    enc, scale = rnd(B, 128 * V, T, seed=8), rnd(B, 128 * V, seed=14)
    w = rnd(128 * V, 64, K, seed=9, scale=0.1)
    gy = rnd(B, 128 * V, T, seed=15)
    encd, scd, gyd = g(enc), g(scale), g(gy)
    for which in (0, 1):        # strided half-views + in_scale (the z1 / z2 split)
        xin = enc.view(B, V, 2, 64, T)[:, :, which].reshape(B, 64 * V, T)
        sc = scale.view(B, V, 2, 64)[:, :, which].reshape(B, 64 * V)
        wr = w.clone().requires_grad_(True)
        F.conv1d(xin * sc[:, :, None], wr, None, 1, 1, 1, V).backward(gy)
        gw = o.conv_bwd_weight(GV.half(encd, V, which), GV.dense(gyd, V), K,
                               in_scale=(scd.view(-1)[which * 64:], 128 * V, 128), wino=form)
        assert rel(gw, wr.grad) < GRAD_TOL
    P, Bp = 3, 2
    for mode in (1, 2, 3):      # decoder prologues
        for Cig, Cog, T_out in ((256, 128, 250), (128, 64, 500), (64, 64, 260), (128, 128, 128)):
            Tin = T_out // 2 if mode & 2 else T_out
            x = rnd(P * Bp, Cig, Tin, seed=80)
            a, b = rnd(P, Cig, seed=81) + 1.2, rnd(P, Cig, seed=82) * 0.5
            xin = x
            if mode & 1:
                xin = F.relu(x * a.repeat_interleave(Bp, 0)[:, :, None] + b.repeat_interleave(Bp, 0)[:, :, None])
            if mode & 2:
                xin = F.interpolate(xin, scale_factor=2, mode="linear", align_corners=False)
            wr = rnd(Cog, Cig, 3, seed=83, scale=(3 * Cig) ** -0.5).requires_grad_(True)
            gy2 = rnd(P * Bp, Cog, T_out, seed=85)
            F.conv1d(xin, wr, None, 1, 1).backward(gy2)
            pro = (mode, g(a) if mode & 1 else None, g(b) if mode & 1 else None, Bp)
            gw = o.conv_bwd_weight(GV.dense(g(x), 1), GV.dense(g(gy2), 1), 3, pro=pro, wino=form)
            assert rel(gw, wr.grad) < GRAD_TOL, (mode, Cig, Cog, T_out)


@pytest.mark.parametrize("K,T,B,G,C", [(3, 64, 1, 1, 64), (3, 66, 1, 1, 64), (3, 98, 2, 1, 64), (7, 64, 1, 1, 64), (7, 66, 1, 1, 64),
                                       (7, 70, 2, 1, 64), (7, 130, 1, 2, 64), (3, 1250, 1, 1, 128), (7, 1250, 1, 1, 128)])
def test_conv_bwd_weight_dma_edges(K, T, B, G, C):
    """The LDS-DMA weight-gradient kernel (conv_bww_glds.hip) at the ends of a sample and of the whole operand: tiles whose
    image reaches outside [0, T) are patched (zeros), chunks that reach outside the tensor are not fetched and their valid
    elements come in lane by lane.  Tiny operands, so one lost or foreign element shows as >= 1e-3; both operands carry
    large values in their first and last columns, and the row after / before each sample row is the neighbour whose
    columns a missing patch would pick up."""
    o = ops()
    from electrocardio_panorama_amd.ops import GV
    x, gy = rnd(B, G * C, T, seed=90), rnd(B, G * C, T, seed=91)
    for t in (x, gy):
        t[:, :, :4] *= 8.0
        t[:, :, -4:] *= 8.0
    w64 = torch.zeros(G * C, C, K, dtype=torch.float64, requires_grad=True)
    F.conv1d(x.double(), w64, None, 1, K // 2, 1, G).backward(gy.double())
    gw = o.conv_bwd_weight(GV.dense(g(x), G), GV.dense(g(gy), G), K, wino=4)
    assert rel(gw, w64.grad) < 2e-5
    # the same rows as the second half of a twice-as-tall tensor: the kernel's operand now ends exactly where the
    # allocation's view does, and starts after foreign rows
    x2 = torch.cat([rnd(B, G * C, T, seed=92) * 100.0, x], 0)
    gy2 = torch.cat([rnd(B, G * C, T, seed=93) * 100.0, gy], 0)
    gw2 = o.conv_bwd_weight(GV.dense(g(x2)[B:], G), GV.dense(g(gy2)[B:], G), K, wino=4)
    assert torch.equal(gw2, gw)


@pytest.mark.parametrize("K,T", [(3, 66), (7, 66), (3, 70), (7, 98)])
def test_conv_bwd_weight_dma_stays_inside_the_operand(K, T):
    """An out-of-bounds READ does not change results, so it is made fatal instead: both operands are exact multiples of
    2 MiB, allocated after the cache was emptied -- each is a segment of its own whose last byte is the last byte the
    driver mapped.  (This is how a fetch of x[T] on the operand's last row was found: 256 x 8 x 5000 happened to end on a
    page boundary.)  T = 66 / 70 / 98: the tile BEFORE the last one already fetches a chunk that crosses T."""
    o = ops()
    from electrocardio_panorama_amd.ops import GV
    B, G, C = 256, 16, 64
    assert (B * G * C * T * 4) % (2 << 20) == 0
    torch.cuda.empty_cache()
    x = torch.randn(B, G * C, T, device=DEV)
    gy = torch.randn(B, G * C, T, device=DEV)
    gw = o.conv_bwd_weight(GV.dense(x, G), GV.dense(gy, G), K, wino=4)
    torch.cuda.synchronize()
    want = o.conv_bwd_weight(GV.dense(x, G), GV.dense(gy, G), K, wino=False)
    assert rel(gw, want) < 1e-4


@pytest.mark.parametrize("T,B,Cig,Cog", [(64, 1, 64, 64), (68, 2, 64, 64), (100, 1, 64, 128), (132, 2, 128, 64), (2500, 1, 256, 128)])
def test_conv_bwd_weight_dma_upsampling_prologue(T, B, Cig, Cog):
    """The x2-upsampling prologue inside the LDS-DMA weight gradient: the half-resolution samples are staged and a quad's six
    inputs interpolated from four of them; the two clamped columns (t = 0, T - 1), the conv's zero padding and the ends of
    the operand are handled by index in the edge tiles.  Against F.interpolate + autograd in fp64, tiny operands."""
    o = ops()
    from electrocardio_panorama_amd.ops import GV
    x, gy = rnd(B, Cig, T // 2, seed=95), rnd(B, Cog, T, seed=96)
    for t in (x, gy):
        t[:, :, :3] *= 8.0
        t[:, :, -3:] *= 8.0
    w64 = torch.zeros(Cog, Cig, 3, dtype=torch.float64, requires_grad=True)
    xin = F.interpolate(x.double(), scale_factor=2, mode="linear", align_corners=False)
    F.conv1d(xin, w64, None, 1, 1).backward(gy.double())
    pro = (2, None, None, 1)
    gw = o.conv_bwd_weight(GV.dense(g(x), 1), GV.dense(g(gy), 1), 3, pro=pro, wino=4)
    assert rel(gw, w64.grad) < 2e-5
    x2 = torch.cat([rnd(B, Cig, T // 2, seed=97) * 100.0, x], 0)
    gy2 = torch.cat([rnd(B, Cog, T, seed=98) * 100.0, gy], 0)
    gw2 = o.conv_bwd_weight(GV.dense(g(x2)[B:], 1), GV.dense(g(gy2)[B:], 1), 3, pro=pro, wino=4)
    assert torch.equal(gw2, gw)


def test_conv_bwd_weight_dma_upsampling_stays_inside_the_operand():
    """As test_conv_bwd_weight_dma_stays_inside_the_operand, for the half-resolution operand of the upsampling prologue."""
    o = ops()
    from electrocardio_panorama_amd.ops import GV
    B, C, T = 256, 1024, 132
    assert (B * C * (T // 2) * 4) % (2 << 20) == 0
    torch.cuda.empty_cache()
    x = torch.randn(B, C, T // 2, device=DEV)
    gy = torch.randn(B, 64, T, device=DEV)
    pro = (2, None, None, 1)
    gw = o.conv_bwd_weight(GV.dense(x, 1), GV.dense(gy, 1), 3, pro=pro, wino=4)
    torch.cuda.synchronize()
    want = o.conv_bwd_weight(GV.dense(x, 1), GV.dense(gy, 1), 3, pro=pro, wino=False)
    assert rel(gw, want) < 1e-4


# ------------------------------------------------------------------------------------------------ split-fp16 direct convs
def test_conv_h2_is_fp32_class(conv_algo):
    """csrc/conv_h2.hip against fp64: forward and backward-data of a 128-channel K = 7 and K = 3 layer land within 3x of
    torch's own fp32 conv (measured ~1.7x; the Winograd F(4,.) forms: 4..5x) -- and the operand scale makes that hold for
    operands of ANY magnitude: a 1e-7-sized gradient operand (explicit power-of-two x_scale, and the measured scale of a
    scope-less launch) is as accurate as an O(1) one."""
    if conv_algo != "h2":
        pytest.skip("split-fp16 path")
    o = ops()
    from electrocardio_panorama_amd.ops import GV
    for K, G, Cig, Cog, B, T in [(7, 3, 128, 128, 3, 1250), (3, 1, 256, 128, 2, 500), (3, 1, 64, 64, 3, 514), (1, 3, 64, 128, 3, 1250)]:
        x = F.relu(rnd(B, G * Cig, T, seed=301))
        w = rnd(G * Cog, Cig, K, seed=302, scale=0.05)
        ref64 = F.conv1d(x.double(), w.double(), None, 1, K // 2, 1, G)
        e32 = rel(F.conv1d(x, w, None, 1, K // 2, 1, G), ref64)
        wp = o.pack_weight(g(w), G, T=T, f4=True)
        assert wp.nef_wino == 3 and wp.numel() == packed_floats(o, 3, K, G, Cog, Cig)
        y = o.conv(GV.dense(g(x), G), wp, Cog, K)
        assert rel(y.double().cpu(), ref64) < 3 * e32 + 1e-8, (K, rel(y.double().cpu(), ref64), e32)
        # backward-data, operand 7 orders of magnitude below 1
        gy = rnd(B, G * Cog, T, seed=303) * 1e-7
        gref64 = torch.nn.grad.conv1d_input(x.shape, w.double(), gy.double(), padding=K // 2, groups=G)
        g32 = rel(torch.nn.grad.conv1d_input(x.shape, w, gy, padding=K // 2, groups=G), gref64)
        wf = o.pack_weight(g(w), G, flip=True, T=T, f4=True)
        assert wf.nef_wino == 3
        gx_meas = o.conv(GV.dense(g(gy), G), wf, Cig, K)                      # no scope: the launch measures its operand first
        gx_expl = o.conv(GV.dense(g(gy), G), wf, Cig, K, x_scale=2.0 ** 30)    # the caller's power of two
        for gx in (gx_meas, gx_expl):
            assert rel(gx.double().cpu(), gref64) < 3 * g32 + 1e-8, (K, rel(gx.double().cpu(), gref64), g32)
        # without a scale the low terms of such an operand fall below fp16's range: the reason the scale exists
        gx_raw = o.conv(GV.dense(g(gy), G), wf, Cig, K, x_scale=1.0)
        assert rel(gx_raw.double().cpu(), gref64) > 1e-4


def _dist_operand(kind, shape, seed):
    """Operands the split-fp16 format finds hard: `lognormal` = exp(4 N(0,1)) with random signs (34 binades between the median and
    the largest of 10^6 elements), `outlier` = uniform [-1, 1] with ONE element 10^4 x the rms, `rows` = every second sample scaled
    by 2^-12 (per-tensor scaling cannot serve both halves at full precision)."""
    gen = torch.Generator().manual_seed(seed)
    if kind == "lognormal":
        x = torch.exp(4.0 * torch.randn(*shape, generator=gen)) * torch.sign(torch.rand(*shape, generator=gen) - 0.5)
    else:
        x = torch.rand(*shape, generator=gen) * 2 - 1
        if kind == "outlier":
            x.reshape(-1)[x.numel() // 3] = 1e4 * float(x.pow(2).mean().sqrt())
        elif kind == "rows":
            x[1::2] *= 2.0 ** -12
    return x


def _region_err(y, ref64):
    """rel-L2 over all elements, and over the half of the elements whose reference magnitude is below the median."""
    y, ref64 = y.double().cpu().reshape(-1), ref64.reshape(-1)
    small = ref64.abs() <= ref64.abs().median()
    return rel(y, ref64), rel(y[small], ref64[small])


@pytest.mark.parametrize("kind", ["lognormal", "outlier", "rows"])
@pytest.mark.parametrize("K", [3, 7])
def test_conv_h2_operand_distributions(conv_algo, kind, K):
    """The split-fp16 kernels as a FORMAT (one power-of-two scale per tensor, two fp16 terms per element): forward, backward-data and
    weight gradient on heavy-tailed, single-outlier and mixed-row-scale operands against fp64 -- flat rel-L2 (bar: 3 x torch's own
    fp32 error, as for uniform operands) AND rel-L2 over the small-magnitude half of the outputs, where the format's absolute floor
    (2^-25 of the scaled tensor's largest element) shows: bars from the analysis in conv_h2.hip's header -- an element 2^-k below
    the largest keeps 22 - max(0, k - 11) bits, so outputs fed by inputs 2^-13 (outlier) / 2^-12 (rows) below the largest
    are good to ~2^-19 / 2^-20 relative -- fp32-CLASS on the flat norm, not per region (bars and measurements at the asserts).
    No launch may clamp."""
    if conv_algo != "h2":
        pytest.skip("split-fp16 path")
    o = ops()
    from electrocardio_panorama_amd.ops import GV
    o.h2_clamped()
    B, G, C, T = 6, 2, 128, 512
    x = _dist_operand(kind, (B, G * C, T), 700 + K)
    w = rnd(G * C, C, K, seed=701, scale=0.05)
    gy = _dist_operand(kind, (B, G * C, T), 702 + K)
    # forward
    r64 = F.conv1d(x.double(), w.double(), None, 1, K // 2, 1, G)
    e32 = _region_err(F.conv1d(x, w, None, 1, K // 2, 1, G), r64)
    y = o.conv(GV.dense(g(x), G), o.pack_weight(g(w), G, T=T), C, K)
    e = _region_err(y, r64)
    # backward-data
    d64 = torch.nn.grad.conv1d_input(x.shape, w.double(), gy.double(), padding=K // 2, groups=G)
    d32 = _region_err(torch.nn.grad.conv1d_input(x.shape, w, gy, padding=K // 2, groups=G), d64)
    gx = o.conv(GV.dense(g(gy), G), o.pack_weight(g(w), G, flip=True, T=T), C, K, role="conv_bwd_data")
    ed = _region_err(gx, d64)
    # weight gradient (both operands split)
    w64 = torch.nn.grad.conv1d_weight(x.double(), w.shape, gy.double(), padding=K // 2, groups=G)
    w32 = _region_err(torch.nn.grad.conv1d_weight(x, w.shape, gy, padding=K // 2, groups=G), w64)
    gw = o.conv_bwd_weight(GV.dense(g(x), G), GV.dense(g(gy), G), K, h2=True)
    ew = _region_err(gw, w64)
    import conftest
    conftest.report(f"split-fp16 format, {kind} K={K}: flat / small-half rel-L2 fwd {e[0]:.1e} / {e[1]:.1e} (torch fp32 {e32[0]:.1e} / {e32[1]:.1e}), "
                    f"bwd-data {ed[0]:.1e} / {ed[1]:.1e} ({d32[0]:.1e} / {d32[1]:.1e}), weight grad {ew[0]:.1e} / {ew[1]:.1e} ({w32[0]:.1e} / {w32[1]:.1e})")
    assert o.h2_clamped() == 0
    # flat norm: fp32-class.  4 x torch's own fp32 error for one split operand; the weight gradient splits BOTH operands and
    # under heavy tails on both its flat error reaches 1.2e-6 (30 x fp32's, still 80 x under the 1e-4 gradient bar)
    assert e[0] < 4 * e32[0] + 1e-8 and ed[0] < 4 * d32[0] + 1e-8, (kind, e, e32, ed, d32)
    assert ew[0] < (4e-6 if kind == "lognormal" else 4 * w32[0] + 1e-8), (kind, ew, w32)
    # small-magnitude half of the outputs: where the ABSOLUTE floor of the format shows (measured in round 5, bars = 3 x).
    #   outlier: 1.4e-6 (fp32 3..5e-7): the bulk sits 13 binades under the outlier and keeps ~19 bits;
    #   rows:    as fp32 itself (1.5e-5 vs 1.4e-5): the small rows' outputs are small because their INPUTS are, for both;
    #   lognormal: forward / backward-data 4e-6 .. 3e-5 (fp32 3..4e-7), weight gradient 5..6e-3 (fp32 8e-7) -- a product of two
    #   elements that are BOTH far below their tensor's largest carries an absolute error no per-tensor scale can remove: the
    #   format is fp32-class on the norm, NOT elementwise, and a model whose gradients live in such a tail needs NEF_H2=0.
    small = {"outlier": (5e-6, 5e-6, 6e-6), "rows": (3 * e32[1] + 1e-8, 3 * d32[1] + 1e-8, 3 * w32[1] + 1e-8),
             "lognormal": (1e-4, 2e-5, 2e-2)}[kind]
    assert e[1] < small[0] and ed[1] < small[1] and ew[1] < small[2], (kind, e, ed, ew)


def test_round6_library_ops_match_the_torch_ops_they_replace(conv_algo):
    """The launches that took the last torch kernels out of the replayed train step (round 6): nef_flatten against torch.cat (odd
    sizes, a 4-byte-aligned destination, more than 64 tensors), nef_regroup_halves against the view / permute it replaces (+ inverse),
    nef_amax_roll against the follow-up rule it implements (ops.H2_FOLLOW_UP / _DOWN, `follow` mode), and a pack whose polyphase
    weights are formed INSIDE the pack kernel (nef_pack_desc.src_mode 1) against packing nef_poly_weights' output: same bytes."""
    if conv_algo != "h2":
        pytest.skip("library-level: once")
    o = ops()
    from electrocardio_panorama_amd import _lib
    L = _lib.load()
    # flatten
    ts = [g(rnd(*shp, seed=800 + i)) for i, shp in enumerate([(5, 3), (7,), (2, 2, 2), (1,), (384, 128, 7), (13,)] + [(3,)] * 70)]
    buf = torch.zeros(sum(t.numel() for t in ts) + 5, device=DEV)
    for off in (4, 1):      # 16-byte aligned and 4-byte aligned destinations
        out = buf[off:off + sum(t.numel() for t in ts)]
        out.zero_()
        o.flatten_into(ts, out)
        assert torch.equal(out, torch.cat([t.reshape(-1) for t in ts]))
    # regroup halves
    w = g(rnd(128, 256, 3, seed=801))
    wg = o.regroup_halves(w)
    assert torch.equal(wg, w.view(128, 2, 128, 3).permute(1, 0, 2, 3).contiguous().view(256, 128, 3))
    assert torch.equal(o.regroup_halves(wg, inverse=True), w)
    # amax roll
    gen = torch.Generator().manual_seed(5)
    cur0 = torch.exp(3 * torch.randn(1000, generator=gen)) * (torch.rand(1000, generator=gen) > 0.2)
    nxt0 = torch.exp(3 * torch.randn(1000, generator=gen)) * (torch.rand(1000, generator=gen) > 0.3)
    for follow in (0, 1):
        cur, nxt = g(cur0.clone()), g(nxt0.clone())
        _lib.check(L.nef_amax_roll(cur.data_ptr(), nxt.data_ptr(), 900, o.H2_FOLLOW_UP, o.H2_FOLLOW_DOWN, follow, torch.cuda.current_stream().cuda_stream))
        upd = (nxt0 > 0) & ((nxt0 > 0) if follow else ((cur0 <= 0) | (nxt0 > o.H2_FOLLOW_UP * cur0) | (nxt0 * o.H2_FOLLOW_DOWN < cur0)))
        want = torch.where(upd, nxt0, cur0)
        want[900:] = cur0[900:]
        assert torch.equal(cur.cpu(), want)
        assert float(nxt[:900].abs().max()) == 0.0 and torch.equal(nxt[900:].cpu(), nxt0[900:])
    # polyphase weights formed inside the pack
    for G, Cog, Cig, T in ((1, 64, 128, 2500), (2, 128, 128, 1250)):
        wc = g(rnd(G * Cog, Cig, 3, seed=802, scale=0.05))
        for flip, Cr in ((False, Cog), (True, 0)):
            ref = o.pack_weight(o.poly_weights(wc, Cr), G, flip=flip, T=T, plain=flip)
            got = o.pack_weight(wc, G, flip=flip, T=T, plain=flip, src=("poly", Cr))
            assert ref.nef_wino == got.nef_wino == 3 and torch.equal(ref, got), (G, Cog, flip)
            # ... and through the step-level multi-descriptor launch
            o.pack_many([(wc, G, flip, T, False, ("poly", Cr), flip, None)])
            assert torch.equal(o.pack_weight(wc, G, flip=flip, T=T, plain=flip, src=("poly", Cr)), ref)


def test_conv_h2_heavy_tail_sites_are_counted(conv_algo):
    """The runtime guard behind the format's per-region limit (round 6): a call site whose operand, when the site measures it, has
    more than ops.H2_TAIL_FRAC = 90 % of its nonzero elements below 2^-11 of its largest counts itself in ops.h2_tail_sites() --
    log-normal operands do (99 %: forward site and both operands' weight-gradient site); uniform ones, ReLU outputs (half zeros) and a
    tensor with a geometrically decaying fringe (73 % of the nonzero elements tiny: what the edge of a beat's zero tail and this model's
    gradient tensors look like -- its worst site has 74 %) do not.  (The model's own tensors: 0, asserted in test_model_gpu / reported by bench.py.)"""
    if conv_algo != "h2":
        pytest.skip("split-fp16 path")
    o = ops()
    from electrocardio_panorama_amd.ops import GV
    B, G, C, T, K = 4, 1, 128, 512, 3
    w = g(rnd(G * C, C, K, seed=771, scale=0.05))
    o.h2_tail_sites()
    for kind, want in (("uniform", 0), ("relu", 0), ("fringe", 0), ("lognormal", 2)):
        if kind == "fringe":      # 1/4 of the row carries the signal, the rest decays by 2^-1 per position down to 1e-30
            xf = rnd(B, G * C, T, seed=775)
            xf[:, :, T // 4:] *= torch.pow(0.5, torch.arange(T - T // 4, dtype=torch.float32)).clamp_min(1e-30)
            x = g(xf)
        else:
            x = g(F.relu(rnd(B, G * C, T, seed=774)) if kind == "relu" else _dist_operand(kind, (B, G * C, T), 772))
        gy = g(_dist_operand("uniform" if kind in ("relu", "fringe") else kind, (B, G * C, T), 773))
        with o.amax_scope((o.new_amax_scope(), True)):
            o.conv(GV.dense(x, G), o.pack_weight(w, G, T=T), C, K)
            o.conv_bwd_weight(GV.dense(x, G), GV.dense(gy, G), K, site=w.data_ptr())
        assert o.h2_tail_sites() == want, kind


@pytest.mark.parametrize("K", [3, 7])
def test_conv_h2_range_rescue(conv_algo, K):
    """A launch whose scale is wrong by far more than the headroom (explicit x_scale 2^14 on data of magnitude 10^3 .. 10^5: scaled
    operands of 10^7 .. 10^9 against fp16's 65504) still returns the fp32-class result: every workgroup whose tile does not fit redoes
    it with the scale its own data asks for -- forward with mixed tiles (one sample 100 x larger than the rest, one all-zero),
    backward-data, and the weight gradient with either operand or both out of range (both kernel forms).  Nothing clamps."""
    if conv_algo != "h2":
        pytest.skip("split-fp16 path")
    o = ops()
    from electrocardio_panorama_amd.ops import GV
    o.h2_clamped()
    B, G, C, T = 6, 2, 128, 700
    x = rnd(B, G * C, T, seed=740) * 1e3
    x[2] *= 100.0
    x[4] = 0.0
    w = rnd(G * C, C, K, seed=741, scale=0.05)
    r64 = F.conv1d(x.double(), w.double(), None, 1, K // 2, 1, G)
    y = o.conv(GV.dense(g(x), G), o.pack_weight(g(w), G, T=T), C, K, x_scale=2.0 ** 14)
    assert rel(y, r64) < 5e-7 and float(y[4].abs().max()) == 0.0
    for b in (0, 2):      # per sample: the small samples are not drowned by the large one (tiles rescale independently)
        assert rel(y[b], r64[b]) < 5e-7, b
    gy = rnd(B, G * C, T, seed=742) * 1e4
    d64 = torch.nn.grad.conv1d_input(x.shape, w.double(), gy.double(), padding=K // 2, groups=G)
    gx = o.conv(GV.dense(g(gy), G), o.pack_weight(g(w), G, flip=True, T=T), C, K, role="conv_bwd_data", x_scale=2.0 ** 14)
    assert rel(gx, d64) < 5e-7
    w64 = torch.nn.grad.conv1d_weight(x.double(), w.shape, gy.double(), padding=K // 2, groups=G)
    for sx, sg in ((2.0 ** 14, 2.0 ** -4), (2.0 ** -6, 2.0 ** 14), (2.0 ** 14, 2.0 ** 14)):
        gw = o.conv_bwd_weight(GV.dense(g(x), G), GV.dense(g(gy), G), K, h2=True, x_scale=sx, gy_scale=sg)
        assert rel(gw, w64) < 1e-6, (sx, sg)
    # the first-form kernel (64-channel layers)
    x6, g6 = x[:, :64].contiguous(), gy[:, :64].contiguous()
    w6 = torch.nn.grad.conv1d_weight(x6.double(), (64, 64, K), g6.double(), padding=K // 2)
    gw6 = o.conv_bwd_weight(GV.dense(g(x6), 1), GV.dense(g(g6), 1), K, h2=True, x_scale=2.0 ** 14, gy_scale=2.0 ** 14)
    assert rel(gw6, w6) < 1e-6
    assert o.h2_clamped() == 0


def test_conv_h2_scale_follows_a_drifting_operand(conv_algo):
    """ops.amax_roll: an operand that grows x1.5 per pass for ten passes and then jumps x3 never clamps (the scale follows up as
    soon as the operand doubled: H2_HEADROOM = 64 x per pass whatever the history; round 4 followed at 64 x only, so this
    sequence -- x57 of drift, then x3 -- ran into the clamp); a x200 jump is beyond the headroom and is RESCUED inside the launch
    (no clamp, fp32-class result); shrinking operands keep their scale (sticky) and their precision."""
    if conv_algo != "h2":
        pytest.skip("split-fp16 path")
    o = ops()
    from electrocardio_panorama_amd.ops import GV
    B, G, C, T, K = 4, 1, 128, 512, 3
    x0 = rnd(B, G * C, T, seed=720)
    w = g(rnd(G * C, C, K, seed=721, scale=0.05))
    wp = o.pack_weight(w, G, T=T)
    o.h2_clamped()

    def one_pass(scale):
        o.amax_roll()
        x = g(x0 * scale)
        y = o.conv(GV.dense(x, G), wp, C, K)
        gw = o.conv_bwd_weight(GV.dense(x, G), GV.dense(g(x0), G), K, site=w.data_ptr())
        r64 = F.conv1d((x0 * scale).double(), w.double().cpu(), None, 1, 1, 1, G)
        return rel(y, r64), gw

    with o.amax_scope(o.new_amax_scope()):
        s = 1.0
        for i in range(10):
            assert one_pass(s)[0] < 5e-7
            s *= 1.5
        s *= 3.0
        assert one_pass(s)[0] < 5e-7
        assert o.h2_clamped() == 0
        # beyond the headroom: the tiles whose data does not fit are redone with their own scale inside the launch (range rescue,
        # conv_h2.hip / conv_h2w.hip) -- nothing clamps, nothing is wrong; the pass after it has followed
        assert one_pass(s * 200.0)[0] < 5e-7 and o.h2_clamped() == 0
        assert one_pass(s * 200.0)[0] < 5e-7 and o.h2_clamped() == 0
        assert one_pass(s * 200.0 / 50.0)[0] < 5e-7                         # 50 x smaller: the scale stays (sticky), precision holds
        # 5000 x below the reference (100 x below the last pass): this pass still runs on the old scale -- 13 binades down the
        # absolute floor of the low term starts to show -- and the next one has followed down
        assert one_pass(s * 200.0 / 50.0 / 100.0)[0] < 2e-6
        assert one_pass(s * 200.0 / 50.0 / 100.0)[0] < 5e-7 and o.h2_clamped() == 0


def test_unscoped_conv_launches_do_not_touch_a_sites_scale(conv_algo):
    """unscoped conv() used one magnitude slot and unscoped conv_bwd_weight() two under the SAME key, so the second
    overwrote the slot of whichever real site was allocated next.  They now own separate anonymous slots."""
    if conv_algo != "h2":
        pytest.skip("split-fp16 path")
    o = ops()
    from electrocardio_panorama_amd.ops import GV
    B, G, C, T, K = 4, 1, 64, 256, 3
    x, w = g(rnd(B, C, T, seed=730)), g(rnd(C, C, K, seed=731, scale=0.05))
    st = o._amax_state(x.device)
    o.conv(GV.dense(x, G), o.pack_weight(w, G, T=T), C, K)                                  # unscoped conv first
    with o.amax_scope(o.new_amax_scope()):
        wp = o.pack_weight(w, G, T=T)
        o.amax_roll()
        o.conv(GV.dense(x * 1000.0, G), wp, C, K)                                           # a real site, large operand
        site = [k for k in st["index"] if k[0] is not None and k[0] == o.AMAX_SCOPE][-1]
        i = st["index"][site]
        before = float(st["cur"][i])
        o.conv_bwd_weight(GV.dense(x, G), GV.dense(x, G), K, h2=True, site=None)        # unscoped weight gradient (two slots)
        assert float(st["cur"][i]) == before and before > 100.0


H2P_CASES = [  # K, G, Cig, Cog, B, T_out, pro_mode, extras
    (7, 2, 48, 64, 9, 386, 0, "relu,drop"), (7, 1, 64, 128, 3, 1250, 0, "gate"), (3, 3, 64, 64, 13, 1250, 0, "relu,res,sc"),
    (1, 2, 64, 128, 5, 700, 0, "bias"), (3, 1, 64, 64, 5, 512, 3, "bias,stats"), (3, 2, 64, 128, 6, 2500, 2, "bias"),
    (3, 1, 128, 64, 6, 1000, 1, "bias,stats"), (3, 1, 64, 64, 6, 1000, 0, "bnb"), (3, 1, 64, 128, 6, 1000, 0, "bnbup"),
    (3, 1, 64, 64, 17, 256, 0, "relu"), (3, 1, 48, 64, 2, 130, 1, "bias"),
]


@pytest.mark.parametrize("K,G,Cig,Cog,B,T,pm,extra", H2P_CASES)
def test_conv_h2_producer_consumer_form_is_bit_identical(conv_algo, K, G, Cig, Cog, B, T, pm, extra):
    """tools/experiments/conv_h2p.hip (producer / consumer waves, persistent twelve-wave workgroup; nef_set_option(NEF_OPT_H2_FORM, 1);
    only in libraries built with `csrc/build.py --with-experiments`, skipped otherwise) does
    conv_h2_kernel's arithmetic in the same order: outputs, BatchNorm slot sums and BatchNorm-backward sums must be bit-identical
    -- every prologue mode, every epilogue option, first / interior / last tiles of a row (quad loads vs checked loads), batches
    that are not a multiple of 8 (empty tiles of the walk), 3 and 8 stages."""
    if conv_algo != "h2":
        pytest.skip("split-fp16 path")
    o = ops()
    from electrocardio_panorama_amd import _lib
    from electrocardio_panorama_amd.ops import GV
    L = _lib.load()
    if not hasattr(L, "nef_debug_h2p_occupancy"):
        pytest.skip("library built without the experimental kernel forms (csrc/build.py --with-experiments)")
    ex = set(extra.split(","))
    Tin = T // 2 if pm & 2 else T
    x = g(rnd(B, G * Cig, Tin, seed=901))
    w = g(rnd(G * Cog, Cig, K, seed=902, scale=0.05))
    wp = o.pack_weight(w, G, T=T)
    assert wp.nef_wino == 3
    kw = dict(x_scale=16.0, relu="relu" in ex)
    if "bias" in ex:
        kw["bias"] = g(rnd(G * Cog, seed=903))
    if "drop" in ex:
        kw.update(drop_p=0.2, drop_scale=1.25, seed=77)
    if "sc" in ex:
        kw["in_scale"] = (g(rnd(B, G * Cig, seed=904) + 1.5), G * Cig, Cig)
    if "res" in ex:
        kw["res"] = GV.dense(g(rnd(B, G * Cog, T, seed=905)), G)
    if "gate" in ex:
        kw.update(gate=GV.dense(g(rnd(B, G * Cog, T, seed=906)), G), gate_scale=1.25, role="conv_bwd_data")
    P = 3 if B % 3 == 0 else 1
    if pm:
        kw["pro"] = (pm, g(rnd(P, G * Cig, seed=907) + 0.5), g(rnd(P, G * Cig, seed=908) * 0.3), B // P)
    bn = None
    if "bnb" in ex or "bnbup" in ex:
        up = "bnbup" in ex
        bn = [g(rnd(B, G * Cog, T // 2 if up else T, seed=909)), g(rnd(P, G * Cog, seed=910)), g(rnd(P, G * Cog, seed=911) + 1.5),
              g(rnd(P, G * Cog, seed=912) + 0.5), g(rnd(P, G * Cog, seed=913) * 0.3)]
        kw["role"] = "conv_bwd_data"

    def run():
        k2, slots = dict(kw), None
        if "stats" in ex:
            slots = k2["stats"] = o.conv_stats_buffer(wp, B, G, Cog, T, DEV)
        if bn is not None:
            slots = o.conv_stats_buffer(wp, B, G, Cog, T, DEV)
            k2["bnb"] = (*bn, B // P, slots, int("bnbup" in ex))
        if slots is not None:
            slots[0].fill_(-7.0)
        return o.conv(GV.dense(x, G), wp, Cog, K, **k2), (None if slots is None else slots[0])

    assert L.nef_get_option(_lib.OPT_H2_FORM) == 0
    y0, s0 = run()
    L.nef_set_option(_lib.OPT_H2_FORM, 1)
    try:
        y1, s1 = run()
        y2, _ = run()
    finally:
        L.nef_set_option(_lib.OPT_H2_FORM, 0)
    assert torch.equal(y0, y1) and torch.equal(y1, y2)
    assert s0 is None or torch.equal(s0, s1)
    assert float(y0.abs().max()) > 0


@pytest.mark.parametrize("K,G,Cig,Cog,B,T", [(3, 21, 128, 128, 30, 16), (3, 3, 64, 128, 5, 32), (1, 21, 64, 128, 27, 16), (3, 2, 128, 64, 14, 8),
                                             (3, 1, 128, 128, 3, 64), (3, 7, 128, 128, 13, 20)])
def test_conv_h2_packed_short_rows(conv_algo, K, G, Cig, Cog, B, T):
    """Short rows (8 <= T <= 64, T % 4 == 0) on the split-fp16 kernel: several samples per 256-position tile at a pitch of T + 4,
    zeros staged between them (conv_h2_kernel, PACK).  Forward with bias + residual + ReLU + a dropout mask, and backward-data with
    a ReLU gate, against fp64 -- including a last tile that is not full (B not a multiple of the samples per tile) and exact
    zeros wherever the reference has them; the measured-scale launch and fp32-class accuracy as for the long rows."""
    if conv_algo != "h2":
        pytest.skip("split-fp16 path")
    o = ops()
    from electrocardio_panorama_amd.ops import GV
    x = F.relu(rnd(B, G * Cig, T, seed=341))
    w = rnd(G * Cog, Cig, K, seed=342, scale=0.05)
    bias, res = rnd(G * Cog, seed=343), rnd(B, G * Cog, T, seed=344)
    mask = (torch.rand(B, G * Cog, T, generator=torch.Generator().manual_seed(345)) > 0.2).to(torch.uint8)
    wp = o.pack_weight(g(w), G, T=T, f4=True)
    assert wp.nef_wino == 3
    def ref(dt):
        y = F.relu(F.conv1d(x.to(dt), w.to(dt), bias.to(dt), 1, K // 2, 1, G) + res.to(dt))
        return y * mask.to(dt) * 1.25
    r64 = ref(torch.float64)
    e32 = rel(ref(torch.float32), r64)
    y = o.conv(GV.dense(g(x), G), wp, Cog, K, bias=g(bias), res=GV.dense(g(res), G), relu=True, mask=g(mask), drop_scale=1.25)
    e = rel(y.double().cpu(), r64)
    assert e < 3 * e32 + 1e-8, (e, e32)
    assert torch.equal((y.cpu() == 0), (r64 == 0))                       # the reference's exact zeros (ReLU, dropped elements)
    assert torch.equal(y, o.conv(GV.dense(g(x), G), wp, Cog, K, bias=g(bias), res=GV.dense(g(res), G), relu=True, mask=g(mask), drop_scale=1.25))
    # backward-data through the same layer, gated by the ReLU of its input
    gy = rnd(B, G * Cog, T, seed=346) * 1e-5
    gref = torch.nn.grad.conv1d_input(x.shape, w.double(), gy.double(), padding=K // 2, groups=G) * (x.double() > 0)
    g32 = rel(torch.nn.grad.conv1d_input(x.shape, w, gy, padding=K // 2, groups=G) * (x > 0), gref)
    wf = o.pack_weight(g(w), G, flip=True, T=T, f4=True)
    assert wf.nef_wino == 3
    gx = o.conv(GV.dense(g(gy), G), wf, Cig, K, gate=GV.dense(g(x), G), gate_scale=1.0, role="conv_bwd_data")
    assert rel(gx.double().cpu(), gref) < 3 * g32 + 1e-8
    # against the fp32 kernels the step used before, element by element
    o.H2 = False
    try:
        y0 = o.conv(GV.dense(g(x), G), o.pack_weight(g(w), G, T=T, f4=True), Cog, K, bias=g(bias), res=GV.dense(g(res), G), relu=True,
                    mask=g(mask), drop_scale=1.25)
    finally:
        o.H2 = True
    assert float((y - y0).abs().max()) < 1e-4 * float(y0.abs().max())


def test_conv_h2_sites_are_sticky_and_scoped(conv_algo):
    """Inside a scope (ops.amax_scope, what Model_nefnet sets) a call site measures once, then keeps its power-of-two operand
    scale while the operand stays inside the window (cur / H2_FOLLOW_DOWN, H2_FOLLOW_UP x cur] of what it measured
    (bit-identical repeats, also after a 1.5 x / 10 x-down change of magnitude and back), follows a larger change at the next
    amax_roll() -- upward already at 2 x, so that H2_HEADROOM x of growth per pass never clamps -- and a new scope starts from
    scratch."""
    if conv_algo != "h2":
        pytest.skip("split-fp16 path")
    o = ops()
    from electrocardio_panorama_amd.ops import GV
    K, G, C, B, T = 3, 1, 128, 2, 384
    x, w = g(rnd(B, C, T, seed=311)), g(rnd(C, C, K, seed=312, scale=0.05))
    wp = o.pack_weight(w, G, T=T, f4=True)
    ref = F.conv1d(x.double(), w.double(), None, 1, 1).float()

    def slot():
        st = o._amax_state(x.device)
        (i,) = [v for k, v in st["index"].items() if k[0] is not None and k[0] == tok]
        return float(st["cur"][i])
    tok = o.new_amax_scope()
    with o.amax_scope(tok):
        y1 = o.conv(GV.dense(x, G), wp, C, K)
        m0 = slot()
        assert abs(m0 - float(x.abs().max())) < 1e-6 * m0
        o.amax_roll()
        y2 = o.conv(GV.dense(x, G), wp, C, K)
        assert torch.equal(y1, y2) and slot() == m0
        o.amax_roll()
        y3 = o.conv(GV.dense(x * 1.5, G), wp, C, K)           # inside the window, upward: same scale, nothing re-measured
        assert slot() == m0 and rel(y3, 1.5 * ref) < 1e-6
        o.amax_roll()
        assert slot() == m0
        y3 = o.conv(GV.dense(x * 0.1, G), wp, C, K)           # inside the window, downward
        assert slot() == m0 and rel(y3, 0.1 * ref) < 1e-6
        o.amax_roll()
        assert slot() == m0 and torch.equal(o.conv(GV.dense(x, G), wp, C, K), y1)
        assert o.h2_clamped() == 0
        o.amax_roll()
        y4 = o.conv(GV.dense(x * 50, G), wp, C, K)            # outside: this launch still runs on the old scale (in range: the
        assert rel(y4, 50 * ref) < 1e-6                       # scale leaves H2_HEADROOM = 64 x, clamping starts beyond that) ...
        o.amax_roll()
        assert abs(slot() - 50 * m0) < 1e-3 * m0 * 50         # ... and the next pass follows
        y5 = o.conv(GV.dense(x * 100, G), wp, C, K)           # x 2 on top of it: still inside the window of the NEW reference
        assert rel(y5, 100 * ref) < 1e-6 and o.h2_clamped() == 0
        # a jump past fp16's range from one pass to the next: every tile is redone with its own scale inside the launch -- right
        # result, no clamp; only data that is not finite still counts itself
        o.amax_roll()
        y7 = o.conv(GV.dense(x * 1e6, G), wp, C, K)
        assert torch.isfinite(y7).all() and rel(y7, 1e6 * ref) < 1e-6 and o.h2_clamped() == 0
        xb = (x * 1e6).clone()
        xb[0, 0, 0] = float("inf")
        o.conv(GV.dense(xb, G), wp, C, K)
        assert o.h2_clamped() > 0 and o.h2_clamped() == 0
        o.amax_roll()
        y8 = o.conv(GV.dense(x * 1e6, G), wp, C, K)           # the next pass has followed
        assert rel(y8, 1e6 * ref) < 1e-6 and o.h2_clamped() == 0
        o.amax_roll()
    tok2 = o.new_amax_scope()
    with o.amax_scope(tok2):
        tok = tok2
        y6 = o.conv(GV.dense(x, G), wp, C, K)
        assert slot() == m0 and torch.equal(y6, y1)


@pytest.mark.parametrize("Cig,Cog,T,mode", [(64, 64, 5000, 0), (128, 64, 5000, 3), (128, 128, 2500, 2)])
def test_conv_h2_full_size_every_lane_arrives(conv_algo, Cig, Cog, T, mode):
    """The decoder launches at configs[1]'s size (768 samples): an earlier build lost 16 lanes of one accumulator row to exact 0.0 a
    few hundred times per launch, on a loaded chip only and not reproducibly (packed-fp32 instructions created by SLP vectorisation
    on registers a ds_read_b128 had just returned; every matrix-core source is built with -fno-slp-vectorize since).  Ten launches
    at full load must be BIT-IDENTICAL to each other and to the same kernel run eight samples at a time on an idle chip (a tile's
    arithmetic does not depend on what else is resident), and agree with the fp32 Winograd kernels element by element."""
    if conv_algo != "h2":
        pytest.skip("split-fp16 path")
    o = ops()
    from electrocardio_panorama_amd.ops import GV
    B = 768
    Tin = T // 2 if mode & 2 else T
    x = torch.randn(B, Cig, Tin, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
    w, bias = g(rnd(Cog, Cig, 3, seed=321, scale=0.05)), g(rnd(Cog, seed=322, scale=0.1))
    pa, pb = g(rnd(3, Cig, seed=323).abs() + 0.5), g(rnd(3, Cig, seed=324) * 0.2)
    pro = (mode, pa, pb, B // 3) if mode & 1 else (mode, None, None, 1)
    o.H2 = False
    y0 = o.conv(GV.dense(x, 1), o.pack_weight(w, 1, T=T, f4=True), Cog, 3, bias=bias, pro=pro)
    o.H2 = True
    wp = o.pack_weight(w, 1, T=T, f4=True)
    assert wp.nef_wino == 3
    y1 = o.conv(GV.dense(x, 1), wp, Cog, 3, bias=bias, pro=pro, x_scale=16.0)
    assert int(((y1 - y0).abs() > 1e-3).sum()) == 0
    del y0
    for _ in range(9):
        y2 = o.conv(GV.dense(x, 1), wp, Cog, 3, bias=bias, pro=pro, x_scale=16.0)
        assert torch.equal(y1, y2)
        del y2
    torch.cuda.synchronize()
    for b0 in (0, 8, 248, 256, 504, 512, 760):            # idle-chip reference: 8 samples per launch, each inside one pass
        p = b0 // (B // 3)
        pro_s = (mode, pa[p:p + 1].contiguous(), pb[p:p + 1].contiguous(), 8) if mode & 1 else (mode, None, None, 1)
        ys = o.conv(GV.dense(x[b0:b0 + 8].contiguous(), 1), wp, Cog, 3, bias=bias, pro=pro_s, x_scale=16.0)
        torch.cuda.synchronize()
        assert torch.equal(ys, y1[b0:b0 + 8]), b0


@pytest.mark.parametrize("K,G,Cig,Cog,B,T,mode,insc", [
    (7, 3, 128, 128, 2, 300, 0, False), (3, 1, 64, 128, 3, 130, 0, True), (3, 1, 128, 64, 6, 500, 1, False),
    (3, 2, 128, 128, 6, 256, 2, False), (3, 1, 128, 64, 6, 504, 3, False), (7, 1, 64, 64, 2, 1250, 0, False),
    (1, 3, 64, 128, 3, 1250, 0, False), (3, 1, 64, 64, 3, 66, 0, False),
    # the producer / consumer form (Cout_g % 128 == 0): every prologue, T % 4 == 2, one-tile rows, shares of zero and one tile
    (3, 1, 128, 128, 6, 500, 1, False), (3, 1, 128, 128, 6, 1004, 3, False), (3, 1, 128, 128, 3, 190, 2, False),
    (7, 1, 128, 128, 3, 1250, 0, False), (3, 1, 64, 256, 1, 64, 0, False), (3, 2, 128, 128, 3, 1250, 0, True),
    (7, 2, 64, 128, 1, 66, 0, False)])
def test_conv_bwd_weight_h2(conv_algo, K, G, Cig, Cog, B, T, mode, insc):
    """csrc/conv_h2w.hip -- the weight gradient on exact fp16 splits of both operands -- against fp64 autograd, with a gradient
    operand of magnitude 1e-4, every input prologue (BatchNorm affine + ReLU, x2 upsampling, both) and the channel scale: within
    3x of torch's own fp32 result (measured: below it), and closer to fp64 than the fp32 transposed-Winograd path; deterministic."""
    if conv_algo != "h2":
        pytest.skip("split-fp16 path")
    o = ops()
    from electrocardio_panorama_amd.ops import GV
    Tin = T // 2 if mode & 2 else T
    x, gy = rnd(B, G * Cig, Tin, seed=331), rnd(B, G * Cog, T, seed=332) * 1e-4
    pa, pb, sc = rnd(3, G * Cig, seed=333).abs() + 0.5, rnd(3, G * Cig, seed=334) * 0.2, rnd(B, G * Cig, seed=335)
    def inputs(dt):
        xin = x.to(dt)
        if mode & 1:
            Bp = B // 3
            xin = F.relu(xin * pa.to(dt).repeat_interleave(Bp, 0)[:, :, None] + pb.to(dt).repeat_interleave(Bp, 0)[:, :, None])
        if mode & 2:
            xin = F.interpolate(xin, scale_factor=2, mode="linear", align_corners=False)
        if insc:
            xin = xin * sc.to(dt)[:, :, None]
        return xin
    refs = {}
    for dt in (torch.float64, torch.float32):
        w = torch.zeros(G * Cog, Cig, K, dtype=dt, requires_grad=True)
        F.conv1d(inputs(dt), w, None, 1, K // 2, 1, G).backward(gy.to(dt))
        refs[dt] = w.grad
    pro = (mode, g(pa), g(pb), B // 3) if mode & 1 else ((mode, None, None, 1) if mode else None)
    kw = dict(in_scale=(g(sc), G * Cig, Cig) if insc else None, pro=pro)
    assert o.h2w_ok(K, Cig, Cog, T, mode, insc)
    xv, gv = GV.dense(g(x), G), GV.dense(g(gy), G)
    got = o.conv_bwd_weight(xv, gv, K, **kw)                      # default path = the split-fp16 kernel; no scope: it measures first
    e32 = rel(refs[torch.float32], refs[torch.float64])
    e = rel(got.double().cpu(), refs[torch.float64])
    assert e < 3 * e32 + 1e-8, (e, e32)
    assert torch.equal(got, o.conv_bwd_weight(xv, gv, K, **kw))
    if K != 1:
        e_w = rel(o.conv_bwd_weight(xv, gv, K, h2=False, **kw).double().cpu(), refs[torch.float64])
        assert e < e_w * 1.2 + 1e-8, (e, e_w)
