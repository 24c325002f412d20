"""GPU: the half-precision panorama decoder (csrc/pano_h.hip; SURVEY 8-f2, BASELINE configs 4/5).

The reference computes the view sweep in fp32 only, so there is no reduced-precision behaviour to match: the kernels
are checked (a) exactly-ish against fp64 torch-CPU math on the SAME fp16-rounded operands (what remains is fp32
accumulation order and the final rounding of each output to fp16: <= 2^-11 relative per element), and (b) end to end
against the fp32 product path, the CPU oracle and the reference's golden outputs at the 2e-3 rel-L2 gate SURVEY 8c names.
"""
import glob
import os
import random

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from util import maxabs, rel, rnd

pytestmark = pytest.mark.gpu
DEV = "cuda"
HALF_TOL = 2e-3      # rel-L2 gate for the fp16 decoder vs fp32 results (SURVEY 8c "fp16 (C5) ... <= 2e-3 rel-L2")


def ops():
    from electrocardio_panorama_amd import ops as o
    return o


from test_model_gpu import batch_t, hashed_model  # noqa: E402  (same hashed weights / synthetic batches)


def up2(x):   # [N, C, T] fp64, nn.Upsample(scale_factor=2, mode='linear', align_corners=False)
    return F.interpolate(x, scale_factor=2, mode="linear", align_corners=False)


def test_from_f32_is_a_rounding_transpose():
    x = rnd(3, 70, 45, seed=1, scale=4.0)
    y = ops().pano_h_from_f32(x.to(DEV))
    assert y.dtype == torch.float16 and y.shape == (3, 45, 70)
    assert torch.equal(y.cpu(), x.transpose(1, 2).to(torch.float16))


@pytest.mark.parametrize("T,NB,nq", [(256, 3, 4), (40, 2, 3), (254, 70, 5), (258, 2, 2), (504, 2, 3), (1000, 3, 2), (2500, 2, 3), (1250, 40, 7),
                                     (2, 1, 1), (6, 2, 3), (252, 3, 2), (506, 2, 2), (758, 1, 3)])      # (padding rows / out-of-sequence slots of ragged tiles)
def test_fused_layer_pair_is_bit_identical_to_two_launches(T, NB, nq):
    """nef_pano_h_conv_pair (layers 1 + 2 with c1 on chip) against nef_pano_h_conv twice: same k order per output and
    the same fp16 rounding of the intermediate, so the bytes must match; more pairs than CUs in the third case, so a
    block walks several pairs and re-uses its c1 rows.  Sequences longer than 256 rows (round 6: configs[4]'s 2500) run in tiles of
    252 output rows with recomputed halo slots: tile starts, exact multiples, ragged ends, more (pair, tile) items than CUs."""
    o = ops()
    N, Tin = NB * nq, T // 2
    xh = rnd(NB, Tin, 256, seed=11).to(torch.float16).to(DEV)
    w1 = (rnd(128, 256, 3, seed=12) * (2.0 / (3 * 256)) ** 0.5).to(DEV)
    w2 = (rnd(128, 128, 3, seed=13) * (2.0 / (3 * 128)) ** 0.5).to(DEV)
    b1, b2 = rnd(128, seed=14, scale=0.1).to(DEV), rnd(128, seed=15, scale=0.1).to(DEV)
    sc = rnd(NB, nq + 2, 256, seed=16, scale=1.5).to(DEV)
    scale = (sc[:, 1:], (nq + 2) * 256, 256)
    wp1, wp2 = o.pano_h_pack_weight(w1), o.pano_h_pack_weight(w2)
    c1 = o.pano_h_conv(xh, wp1, b1, 128, N=N, upsample=True, scale=scale, x_div=nq, nq=nq)
    ref = o.pano_h_conv(c1, wp2, b2, 128)
    y = o.pano_h_conv_pair(xh, wp1, b1, scale, wp2, b2, N, nq, nq)
    assert y.shape == ref.shape == (N, T, 128)
    assert torch.equal(y, ref), float((y.float() - ref.float()).abs().max())


@pytest.mark.parametrize("T,NB,nq", [(512, 3, 4), (80, 2, 3), (508, 70, 5), (256, 2, 2),
                                     (514, 2, 2), (1016, 2, 3), (1100, 3, 2), (5000, 2, 3), (2500, 40, 7),
                                     (2, 1, 1), (6, 2, 3), (510, 3, 2), (1018, 2, 2), (1526, 1, 3)])      # (padding rows / out-of-sequence slots of ragged tiles)
def test_fused_tail_matches_two_launches(T, NB, nq):
    """nef_pano_h_conv_tail (layers 3 + 4 + last conv + sigmoid, c3 / c4 on chip; round 6) against nef_pano_h_conv(upsample) +
    nef_pano_h_conv_outconv: same k order per output and the same fp16 roundings of c3 and c4; the last conv runs on the matrix cores
    here (fp32 weights as two fp16 terms) and as a fused-multiply-add chain there, so the views agree to fp32 round-off of a 192-term
    sum, not bit for bit -- whole tile, short and ragged sequences, more pairs than CUs (a block walks several pairs and re-uses
    both row buffers).  Sequences longer than 512 rows (configs[4]: 5000) run in tiles of 508 output rows with recomputed halo rows: tile
    starts, exact multiples, ragged ends, more (pair, tile) items than CUs."""
    o = ops()
    N, Tin = NB * nq, T // 2
    c2 = F.relu(rnd(N, Tin, 128, seed=21)).to(torch.float16).to(DEV)
    w3 = (rnd(64, 128, 3, seed=22) * (2.0 / (3 * 128)) ** 0.5).to(DEV)
    w4 = (rnd(64, 64, 3, seed=23) * (2.0 / (3 * 64)) ** 0.5).to(DEV)
    b3, b4 = rnd(64, seed=24, scale=0.1).to(DEV), rnd(64, seed=25, scale=0.1).to(DEV)
    wout, bout = (rnd(1, 64, 3, seed=26) * 0.2).to(DEV), rnd(1, seed=27, scale=0.1).to(DEV)
    wp3, wp4 = o.pano_h_pack_weight(w3), o.pano_h_pack_weight(w4)
    ref = torch.full((NB, nq + 1, T), -7.0, device=DEV)
    got = torch.full((NB, nq + 1, T), -7.0, device=DEV)
    c3 = o.pano_h_conv(c2, wp3, b3, 64, upsample=True)
    o.pano_h_conv_outconv(c3, wp4, b4, wout, bout, ref[:, 1:], nq, (nq + 1) * T, T)
    o.pano_h_conv_tail(c2, wp3, b3, wp4, b4, wout, bout, got[:, 1:], nq, (nq + 1) * T, T)
    assert torch.equal(got[:, 0], ref[:, 0])                          # the view in front of the addressed ones is untouched
    assert float((got - ref).abs().max()) < 5e-7, float((got - ref).abs().max())
    assert float(got[:, 1:].min()) > 0.0 and float(got[:, 1:].max()) < 1.0


@pytest.mark.parametrize("Cin,Cout,T,N,upsample,scaled", [
    (128, 128, 256, 2, False, False),      # whole tiles
    (128, 128, 300, 3, False, False),      # ragged last tile
    (64, 64, 520, 2, False, False),        # 256-column tiles, ragged
    (128, 64, 512, 2, True, False),        # x2 upsample while staging
    (128, 64, 200, 1, True, False),
    (256, 128, 256, 6, True, True),        # layer 1: shared latent, per-(sample, angle) channel scale
    (256, 128, 40, 3, True, True),         # shorter than one tile
    (256, 128, 600, 6, True, True),        # layer 1 over several 256-column tiles: blended halo rows, ragged tail
    (256, 128, 512, 3, False, True),       # query scaling without the upsampling
    (128, 128, 700, 2, False, False),      # 128 -> 128 over three tiles
])
def test_h_conv_matches_fp64_on_the_same_operands(Cin, Cout, T, N, upsample, scaled):
    o = ops()
    Tin = T // 2 if upsample else T
    nq = 3 if scaled else 1                 # angles per sample
    NB = N // nq if scaled else N           # stored x rows
    xh = rnd(NB, Tin, Cin, seed=2).to(torch.float16)
    w = (rnd(Cout, Cin, 3, seed=3) * (2.0 / (3 * Cin)) ** 0.5).to(torch.float16).float()
    bias = rnd(Cout, seed=4, scale=0.1)
    sc = rnd(NB, 5, Cin, seed=5, scale=1.5) if scaled else None          # [sample][angle slot][ci], uses slots 1..3
    wp = o.pano_h_pack_weight(w.to(DEV))
    scale = (sc.to(DEV)[:, 1:], 5 * Cin, Cin) if scaled else None
    y = o.pano_h_conv(xh.to(DEV), wp, bias.to(DEV), Cout, N=N, upsample=upsample, scale=scale,
                      x_div=nq if scaled else 1, nq=nq)
    assert y.shape == (N, T, Cout) and y.dtype == torch.float16
    # the same pipeline in fp64: blend (fp32 in the kernel) -> scale -> round to fp16 -> conv -> +bias -> ReLU
    x = xh.double().transpose(1, 2)                                       # [NB, Cin, Tin]
    if upsample:
        x = up2(x)
    if scaled:
        x = torch.stack([x[n // nq] * sc[n // nq, 1 + n % nq].double()[:, None] for n in range(N)])
    x = x.float().to(torch.float16).double()                              # staging rounds once, from fp32
    ref = F.relu(F.conv1d(x, w.double(), bias.double(), padding=1)).transpose(1, 2)
    err = (y.cpu().double() - ref).abs()
    # fp16 output rounding (2^-11 relative) + fp32 accumulation noise; staging double-rounding can flip one input ulp
    bound = 2.0 ** -10 * ref.abs() + 2e-3
    assert bool((err <= bound).all()), float((err - bound).max())
    assert rel(y, ref) < 5e-4


def test_h_outconv():
    o = ops()
    N, T, nq = 6, 700, 3
    xh = rnd(N, T, 64, seed=6, scale=2.0).to(torch.float16)
    w, b = rnd(1, 64, 3, seed=7, scale=0.2), rnd(1, seed=8)
    out = torch.full((2, 5, T), -1.0, device=DEV)
    o.pano_h_outconv(xh.to(DEV), w.to(DEV), b.to(DEV), out[:, 1:], nq, 5 * T, T)
    ref = torch.sigmoid(F.conv1d(xh.double().transpose(1, 2), w.double(), b.double(), padding=1) / 3.0)[:, 0]
    got = out.cpu()
    assert torch.all(got[:, 0] == -1.0) and torch.all(got[:, 4] == -1.0)      # untouched angle slots
    assert rel(got[:, 1:4].reshape(N, T), ref) < 1e-6


@pytest.mark.parametrize("T", [256, 700, 5000, 40])
def test_h_conv_outconv_fused_equals_two_calls(T):
    """Layer 4 + last conv in one pass (the 64-channel tile never reaches memory) against the two-call sequence: the
    intermediate is rounded to fp16 identically, only the summation order of the 192-term last conv differs."""
    o = ops()
    N, nq = 6, 3
    xh = rnd(N, T, 64, seed=21).to(torch.float16).to(DEV)
    w4 = (rnd(64, 64, 3, seed=22) * 0.1).to(DEV)
    b4 = rnd(64, seed=23, scale=0.1).to(DEV)
    wo, bo = rnd(1, 64, 3, seed=24, scale=0.2).to(DEV), rnd(1, seed=25).to(DEV)
    wp = o.pano_h_pack_weight(w4)
    two = torch.full((2, 5, T), -1.0, device=DEV)
    c4 = o.pano_h_conv(xh, wp, b4, 64)
    o.pano_h_outconv(c4, wo, bo, two[:, 1:], nq, 5 * T, T)
    one = torch.full((2, 5, T), -1.0, device=DEV)
    o.pano_h_conv_outconv(xh, wp, b4, wo, bo, one[:, 1:], nq, 5 * T, T)
    assert torch.all(one[:, 0] == -1.0) and torch.all(one[:, 4] == -1.0)
    assert rel(one[:, 1:4], two[:, 1:4]) < 1e-6
    assert maxabs(one[:, 1:4], two[:, 1:4]) < 1e-6
    again = torch.full((2, 5, T), 7.0, device=DEV)                # stale contents at the tile edges must not leak in
    o.pano_h_conv_outconv(xh, wp, b4, wo, bo, again[:, 1:], nq, 5 * T, T)
    assert torch.equal(again[:, 1:4], one[:, 1:4])                # and the result is run-to-run identical


def _logit3(o):
    o = torch.as_tensor(o).double().cpu().clamp(1e-12, 1 - 1e-12)
    return 3.0 * torch.log(o / (1 - o))


@pytest.mark.parametrize("B,V,L,Q", [(2, 3, 512, 5), (3, 1, 1000, 7), (2, 8, 512, 4)])
def test_sweep_fp16_vs_fp32_path(B, V, L, Q):
    """Both product paths on the same inputs: outputs and pre-sigmoid logits within the half-precision gate."""
    m = hashed_model(V).eval()
    b = batch_t(B, V, L, 11, Q)
    random.seed(0)
    ref = m(b["data"], b["input_theta"], b["target_theta"], b["rois"], rest_theta=b["rest_theta"], phase="test")
    m.panorama_dtype = "fp16"
    random.seed(0)
    got = m(b["data"], b["input_theta"], b["target_theta"], b["rois"], rest_theta=b["rest_theta"], phase="test")
    for a, r in zip(got[:3], ref[:3]):
        assert torch.equal(a, r)                      # the three training-style outputs stay on the fp32 path
    assert got[3].shape == (B, Q, L) and got[3].dtype == torch.float32
    assert rel(got[3], ref[3]) < HALF_TOL, rel(got[3], ref[3])
    assert rel(_logit3(got[3]), _logit3(ref[3])) < 2 * HALF_TOL, rel(_logit3(got[3]), _logit3(ref[3]))
    z1, z2 = m(b["data"], b["input_theta"], b["target_theta"], b["rois"], phase="gen")
    g16 = m.gen_ecg(z1, z2, b["rest_theta"], b["rois"])
    m.panorama_dtype = "fp32"
    g32 = m.gen_ecg(z1, z2, b["rest_theta"], b["rois"])
    assert rel(g16, g32) < HALF_TOL
    assert torch.equal(g32, ref[3])


def test_sweep_fp16_vs_reference_golden(golden_dir):
    """Against the reference's own outputs (tests/golden/eval_*.npz, written by oracle/make_golden.py)."""
    files = sorted(glob.glob(os.path.join(golden_dir, "eval_*.npz")))
    assert files
    for f in files:
        z = np.load(f)
        B, V, L, Q, seed = (int(z[k]) for k in ("B", "V", "L", "Q", "seed"))
        b = batch_t(B, V, L, seed, Q)
        m = hashed_model(V).eval()
        m.panorama_dtype = "fp16"
        random.seed(seed)
        rest = m(b["data"], b["input_theta"], b["target_theta"], b["rois"], rest_theta=b["rest_theta"], phase="test")[3]
        assert rel(rest, z["rest_out"]) < HALF_TOL, (os.path.basename(f), rel(rest, z["rest_out"]))
        z1, z2 = m(b["data"], b["input_theta"], b["target_theta"], b["rois"], phase="gen")
        assert rel(m.gen_ecg(z1, z2, b["rest_theta"], b["rois"]), z["gen_ecg"]) < HALF_TOL


def test_sweep_fp16_chunking_is_invisible():
    """Angle chunks (bounded intermediates) must not change a single bit of the result."""
    from electrocardio_panorama_amd import engine
    m = hashed_model(1).eval()
    b = batch_t(4, 1, 512, 5, 9)
    z1, z2 = m(b["data"], b["input_theta"], b["target_theta"], b["rois"], phase="gen")
    P, Bf = dict(m.named_parameters()), dict(m.named_buffers())
    with torch.no_grad():
        z2r = ops().roi_unpool_fwd(z2.contiguous(), b["rois"], z1.shape[2])
        latent = ops().lead_mean(z1, z2r, 1)
        whole = engine.sweep_eval_h(P, Bf, latent, b["rest_theta"], pair_budget=1 << 20)
        parts = engine.sweep_eval_h(P, Bf, latent, b["rest_theta"], pair_budget=8)       # 2 angles per chunk, ragged end
    assert torch.equal(whole, parts)


def test_panorama_dtype_is_validated():
    m = hashed_model(1).eval()
    m.panorama_dtype = "bf16"
    b = batch_t(2, 1, 512, 5, 3)
    with pytest.raises(ValueError):
        m(b["data"], b["input_theta"], b["target_theta"], b["rois"], rest_theta=b["rest_theta"], phase="test")


def _oracle_rest(V, b, rows, seed):
    """The CPU oracle's eval-mode sweep (reference model_nefnet.py:181-192) on the selected rows of a host batch."""
    from oracle import hashweights as hw
    from oracle import nefnet_oracle as orc
    P, Bf = hw.hashed_params(V), hw.hashed_buffers()
    sub = {k: v[rows] for k, v in b.items()}
    with torch.no_grad():
        random.seed(seed)
        return orc.forward(P, Bf, sub["data"], sub["input_theta"], sub["target_theta"], sub["rois"],
                           rest_theta=sub["rest_theta"], phase="test", training=False)[3]


def test_config3_full_size_sweep_rows_vs_oracle():
    """BASELINE configs[3] AT SIZE: batch 1024, 1 lead in -> 360 queried angles, len 512, fp16 decoder -- the whole
    sweep runs (chunked by the default pair budget), rows 0 / 511 / 1023 x all 360 angles are held to the CPU oracle at the
    half-precision gate, and the same rows of the fp32 product path to 1e-5 (eval mode is batch-independent: running
    BatchNorm statistics)."""
    from electrocardio_panorama_amd import synth
    B, V, L, Q, seed = 1024, 1, 512, 360, 77
    host = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.make_batch(B, V, L, seed=seed, Q=Q).items()
            if k in ("data", "input_theta", "target_theta", "rois", "rest_theta")}
    d = {k: v.to(DEV) for k, v in host.items()}
    m = hashed_model(V).eval()
    m.panorama_dtype = "fp16"
    random.seed(seed)
    rest = m(d["data"], d["input_theta"], d["target_theta"], d["rois"], rest_theta=d["rest_theta"], phase="test")[3]
    assert rest.shape == (B, Q, L) and rest.dtype == torch.float32 and bool(torch.isfinite(rest).all())
    rows = [0, 511, 1023]
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ref = _oracle_rest(V, host, rows, seed)
    for i, r in enumerate(rows):
        assert rel(rest[r], ref[i]) < HALF_TOL, (r, rel(rest[r], ref[i]))
    m.panorama_dtype = "fp32"            # the reference-arithmetic path on the same three rows
    random.seed(seed)
    sub = {k: v[rows] for k, v in d.items()}
    r32 = m(sub["data"], sub["input_theta"], sub["target_theta"], sub["rois"], rest_theta=sub["rest_theta"], phase="test")[3]
    assert rel(r32, ref) < 1e-5, rel(r32, ref)
    # rows of one sweep do not depend on which other samples share the batch: a 3-row batch takes other tile shapes in
    # the fp32 encoder (1e-7 apart), which flips a few fp16 roundings in the decoder -- nothing more
    m.panorama_dtype = "fp16"
    random.seed(seed)
    r16 = m(sub["data"], sub["input_theta"], sub["target_theta"], sub["rois"], rest_theta=sub["rest_theta"], phase="test")[3]
    assert rel(r16, rest[rows]) < 1e-4, rel(r16, rest[rows])


def test_config4_full_share_gen_ecg_rows_vs_oracle():
    """BASELINE configs[4] AT ITS PER-GPU SIZE: one GPU's share of the global batch 4096 = 512 samples x 3 leads x 12 angles,
    len 5000, fp16 decoder (what bench.py's `secondary` times): the whole share runs through gen_ecg, rows 0 / 255 / 511 x all
    12 angles are held to the CPU oracle's sweep at the half-precision gate, and the fp32 product path on the same rows to 1e-5
    (reference model_nefnet.py:196-218)."""
    from electrocardio_panorama_amd import synth
    B, V, L, Q, seed = 512, 3, 5000, 12, 79
    host = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.make_batch(B, V, L, seed=seed, Q=Q).items()
            if k in ("data", "input_theta", "target_theta", "rois", "rest_theta")}
    d = {k: v.to(DEV) for k, v in host.items()}
    m = hashed_model(V).eval()
    with torch.no_grad():
        z1, z2 = m(d["data"], d["input_theta"], d["target_theta"], d["rois"], phase="gen")
        m.panorama_dtype = "fp16"
        g16 = m.gen_ecg(z1, z2, d["rest_theta"], d["rois"])
    assert g16.shape == (B, Q, L) and g16.dtype == torch.float32 and bool(torch.isfinite(g16).all())
    rows = [0, 255, 511]
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ref = _oracle_rest(V, host, rows, seed)
    for i, r in enumerate(rows):
        assert rel(g16[r], ref[i]) < HALF_TOL, (r, rel(g16[r], ref[i]))
    m.panorama_dtype = "fp32"
    with torch.no_grad():
        g32 = m.gen_ecg(z1[rows].contiguous(), z2[rows].contiguous(), d["rest_theta"][rows].contiguous(),
                        d["rois"][rows].contiguous())
    assert rel(g32, ref) < 1e-5, rel(g32, ref)
    assert rel(_logit3(g16[rows]), _logit3(g32)) < 2 * HALF_TOL


def test_config4_gen_ecg_fp16_len5000_vs_fp32_and_oracle():
    """BASELINE configs[4] shape (decoder-only synthesis, len 5000, 3 leads, 12 angles, fp16) at a batch the CPU oracle
    finishes in seconds: gen_ecg through the fp16 decoder (multi-tile sequences, both x2 upsamplings, ROI un-pooling at
    len 5000) against the fp32 product path and the oracle's sweep (reference model_nefnet.py:196-218)."""
    B, V, L, Q, seed = 4, 3, 5000, 12, 78
    host = batch_t(B, V, L, seed, Q, dev="cpu")
    d = {k: v.to(DEV) for k, v in host.items()}
    m = hashed_model(V).eval()
    z1, z2 = m(d["data"], d["input_theta"], d["target_theta"], d["rois"], phase="gen")
    m.panorama_dtype = "fp16"
    g16 = m.gen_ecg(z1, z2, d["rest_theta"], d["rois"])
    m.panorama_dtype = "fp32"
    g32 = m.gen_ecg(z1, z2, d["rest_theta"], d["rois"])
    assert g16.shape == (B, Q, L) and bool(torch.isfinite(g16).all())
    ref = _oracle_rest(V, host, list(range(B)), seed)
    assert rel(g32, ref) < 1e-5, rel(g32, ref)
    assert rel(g16, g32) < HALF_TOL, rel(g16, g32)
    assert rel(g16, ref) < HALF_TOL, rel(g16, ref)
    assert rel(_logit3(g16), _logit3(g32)) < 2 * HALF_TOL
