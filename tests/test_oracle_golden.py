"""CPU: the oracle restatement reproduces the fixtures that oracle/make_golden.py captured from the reference."""
import glob
import os
import random

import numpy as np
import torch

from util import maxabs, rel, stats, sub


def _batch(B, V, L, seed, Q=0):
    from electrocardio_panorama_amd import synth
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.make_batch(B, V, L, seed=seed, Q=Q).items()}


def test_theta_and_roi_fixtures(golden_dir):
    from oracle import hashweights as hw
    from oracle import nefnet_oracle as orc
    z = np.load(os.path.join(golden_dir, "theta_table.npz"))
    assert maxabs(orc.angular_encoding(torch.from_numpy(z["theta"])), z["enc"]) == 0.0
    z = np.load(os.path.join(golden_dir, "roi_cases.npz"))
    for name in sorted({k.split(":")[0] for k in z.files}):
        L = int(z[f"{name}:L"])
        rois = torch.from_numpy(z[f"{name}:rois"])
        Bn, T, C = rois.shape[0], L // 4, 6
        zz = torch.from_numpy(hw.unit_noise("roi-z:" + name, Bn * C * T).reshape(Bn, C, T).astype(np.float32))
        zs = torch.from_numpy(hw.unit_noise("roi-s:" + name, Bn * C * 7 * 32).reshape(Bn, C, 7, 32).astype(np.float32))
        start, length = orc.roi_segment_table(rois)
        assert np.array_equal(start.numpy(), z[f"{name}:seg_start"]) and np.array_equal(length.numpy(), z[f"{name}:seg_len"])
        assert np.array_equal(start.numpy(), z[f"{name}:rois"][..., 0] // 4)          # == rois // 4
        assert rel(orc.roi_align_mid(zz, rois), z[f"{name}:align"]) < 1e-7
        assert rel(orc.roi_unpool(zs, rois), z[f"{name}:unpool"]) < 1e-7


def test_eval_fixture_small(golden_dir):
    from oracle import hashweights as hw
    from oracle import nefnet_oracle as orc
    f = os.path.join(golden_dir, "eval_B2_V1_L512_Q5.npz")
    z = np.load(f)
    B, V, L, Q, seed = (int(z[k]) for k in ("B", "V", "L", "Q", "seed"))
    b = _batch(B, V, L, seed, Q)
    P, Bf = hw.hashed_params(V), hw.hashed_buffers()
    with torch.no_grad():
        random.seed(seed)
        out = orc.forward(P, Bf, b["data"], b["input_theta"], b["target_theta"], b["rois"], rest_theta=b["rest_theta"],
                          phase="test", training=False)
        z1, z2 = orc.forward(P, Bf, b["data"], b["input_theta"], b["target_theta"], b["rois"], phase="gen", training=False)
    for got, key in zip(out, ("out", "shuf_p", "shuf_l", "rest_out")):
        assert rel(got, z[key]) < 1e-6, key
    assert rel(sub(z1), z["z1_sub"]) < 1e-6 and rel(stats(z2), z["z2_stats"]) < 1e-6


def test_train_fixture_small(golden_dir):
    from oracle import hashweights as hw
    from oracle import nefnet_oracle as orc
    z = np.load(os.path.join(golden_dir, "train_B2_V1_L512_l1_loss.npz"))
    B, V, L, seed = (int(z[k]) for k in ("B", "V", "L", "seed"))
    b = _batch(B, V, L, seed)
    P, Bf = orc.require_grad(hw.hashed_params(V)), hw.hashed_buffers()
    random.seed(seed)
    outs = orc.forward(P, Bf, b["data"], b["input_theta"], b["target_theta"], b["rois"], phase="train", training=True,
                       masks=hw.hashed_masks(V, B, L // 4))
    losses = orc.loss_v1(outs[0], outs[1], outs[2], b["target_view"].unsqueeze(1))
    losses[0].backward()
    assert rel(outs[0], z["out"]) < 1e-6
    assert maxabs(torch.stack([l_.detach() for l_ in losses]), z["losses"]) < 1e-6
    for k, p in P.items():
        if k in orc.DEAD_PARAMS:
            assert p.grad is None
        elif not (k.endswith("double_conv.0.bias") or k.endswith("double_conv.3.bias")):
            assert rel(sub(p.grad, 256), z["gsub:" + k]) < 1e-4, k
    assert int(Bf["decoder.1.double_conv.1.num_batches_tracked"]) == 3


def test_tie_free_fixture_pins_oracle(golden_dir):
    """Round 6: the tie-free train fixtures (seeds on which the reference's fp32 run, its fp64 restatement and the HIP path take
    identical ReLU / L1 decisions: oracle/tie_search.py, tests/screen_tie_free.py) -- the oracle restates the reference's run on one of
    them to fp32 round-off, gradients included, and the fp64 run of the oracle takes the SAME decisions (what makes the fixture's
    gradients a valid target at the plain bars)."""
    from oracle import hashweights as hw
    from oracle import nefnet_oracle as orc
    from oracle import tie_search as ts
    files = sorted(glob.glob(os.path.join(golden_dir, "train_*_tf*.npz")))
    assert len(files) >= 3 and glob.glob(os.path.join(golden_dir, "nefnet2_*_tf*.npz"))
    z = np.load([f for f in files if "V3_L512" in f][0])
    assert int(z["tie_free"]) == 1
    B, V, L, seed = (int(z[k]) for k in ("B", "V", "L", "seed"))
    b = _batch(B, V, L, seed)
    P, Bf = orc.require_grad(hw.hashed_params(V)), hw.hashed_buffers()
    random.seed(seed)
    outs = orc.forward(P, Bf, b["data"], b["input_theta"], b["target_theta"], b["rois"], phase="train", training=True,
                       masks=hw.hashed_masks(V, B, L // 4))
    losses = orc.loss_v1(outs[0], outs[1], outs[2], b["target_view"].unsqueeze(1), reg_loss=str(z["reg"]))
    losses[0].backward()
    assert rel(outs[0], z["out"]) < 1e-6 and maxabs(torch.stack([l_.detach() for l_ in losses]), z["losses"]) < 1e-6
    nsub = int(z["nsub"])
    for k, p in P.items():
        if k not in orc.DEAD_PARAMS and not (k.endswith("double_conv.0.bias") or k.endswith("double_conv.3.bias")):
            assert rel(sub(p.grad, nsub), z["gsub:" + k]) < 1e-4, k
    r64 = ts.run(V, B, L, seed, str(z["reg"]), True, torch.float64)
    r32 = ts.run(V, B, L, seed, str(z["reg"]), True, torch.float32)
    assert all(torch.equal(r64.own[k], r32.own[k]) for k in r64.own)
    assert sum(v.numel() for v in r64.own.values()) == int(z["decisions"])


def test_fixture_inventory(golden_dir):
    names = {os.path.basename(f) for f in glob.glob(os.path.join(golden_dir, "*.npz"))}
    assert {"theta_table.npz", "roi_cases.npz", "sgd_B4_V3_L512.npz"} <= names
    assert sum(n.startswith("eval_") for n in names) >= 4 and sum(n.startswith("train_") for n in names) >= 4


def test_baseline_config0_cpu_plumbing():
    """BASELINE.json configs[0]: nef_net.yml surface on CPU, synthetic 12-lead-table batch=4 len=2048, 1 input view ->
    1 target view: the oracle trains for two steps and the loss goes down (plumbing check, no GPU)."""
    from oracle import nefnet_oracle as orc
    b = _batch(4, 1, 2048, seed=7)
    P, Bf = orc.require_grad(orc.reference_style_init(1, seed=123)), orc.fresh_buffers()
    opt = orc.SGDState(0.1)
    random.seed(0)
    l0 = orc.train_step(P, Bf, opt, b, p=0.0)
    l1 = orc.train_step(P, Bf, opt, b, p=0.0)
    l2 = orc.train_step(P, Bf, opt, b, p=0.0)
    assert all(np.isfinite(v) for v in l0 + l1 + l2) and l2[0] < l0[0]
    assert l0[1] == 0.0 and l0[2] == 0.0          # one lead: the Standin passes equal the prediction
    assert int(Bf["decoder.3.double_conv.4.num_batches_tracked"]) == 9


def test_real_recordings_fixture(golden_dir):
    """G7: the two recordings bundled with the reference, through its own dataset class (inputs stored in the fixture)."""
    from oracle import hashweights as hw
    from oracle import nefnet_oracle as orc
    z = np.load(os.path.join(golden_dir, "real_tianchi_B2_V3.npz"))
    t = {k: torch.from_numpy(z[k]) for k in ("data", "rois", "input_theta", "target_theta", "rest_theta")}
    assert t["data"].shape == (2, 3, 512) and t["rois"].shape == (2, 7, 2) and int(t["rois"][0, 6, 1]) == 512
    P, Bf = hw.hashed_params(3), hw.hashed_buffers()
    random.seed(int(z["seed"]))
    with torch.no_grad():
        outs = orc.forward(P, Bf, t["data"], t["input_theta"], t["target_theta"], t["rois"], rest_theta=t["rest_theta"],
                           phase="test", training=False)
    for got, key in zip(outs, ("out", "shuf_p", "shuf_l", "rest_out")):
        assert rel(got, z[key]) < 1e-6, key


def test_nefnet2_fixture(golden_dir):
    """f4: the oracle's Model_nefnet2 restatement (shared single-lead encoder, lead loop) against the reference's own
    outputs and gradients (dropout off)."""
    from oracle import hashweights as hw
    from oracle import nefnet_oracle as orc
    z = np.load(os.path.join(golden_dir, "nefnet2_B2_V3_L512_Q4.npz"))
    B, V, L, Q, seed = (int(z[k]) for k in ("B", "V", "L", "Q", "seed"))
    b = _batch(B, V, L, seed, Q)
    choice = tuple(int(c) for c in z["lead_choice"])
    with torch.no_grad():
        out = orc.forward2(hw.hashed_params2(), hw.hashed_buffers(), b["data"], b["input_theta"], b["target_theta"],
                           b["rois"], rest_theta=b["rest_theta"], phase="test", training=False, lead_choice=choice)
        z1m, z2m = orc.forward2(hw.hashed_params2(), hw.hashed_buffers(), b["data"], b["input_theta"], b["target_theta"],
                                b["rois"], phase="gen", training=False)
    for got, key in zip(out, ("out", "shuf_p", "shuf_l", "rest_out")):
        assert rel(got, z[key]) < 1e-6, key
    assert z1m.shape == (B, 128, L // 4) and rel(sub(z1m), z["z1m_sub"]) < 1e-6 and rel(stats(z2m), z["z2m_stats"]) < 1e-6
    P, Bf = orc.require_grad(hw.hashed_params2()), hw.hashed_buffers()
    outs = orc.forward2(P, Bf, b["data"], b["input_theta"], b["target_theta"], b["rois"], phase="train", training=True,
                        p=0.0, lead_choice=choice)
    losses = orc.loss_v1(outs[0], outs[1], outs[2], b["target_view"].unsqueeze(1), reg_loss=str(z["reg"]))
    losses[0].backward()
    assert maxabs(torch.stack([l_.detach() for l_ in losses]), z["losses"]) < 1e-6
    for k in ("single_conv_z1.0.weight", "single_conv_z2.0.bias", "W_encoder.conv1.weight", "mlp1.weight"):
        assert rel(sub(P[k].grad, 256), z["gsub:" + k]) < 1e-5, k
    assert all(P[k].grad is None for k in orc.DEAD_PARAMS)
