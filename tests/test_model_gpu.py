"""Whole-path parity on the GPU: the HIP-backed Model_nefnet / losswrapper / Solver against (1) the golden
fixtures produced by the reference itself (tests/golden, made by oracle/make_golden.py) and (2) the CPU oracle
run live on the same inputs; plus size-independent properties at BASELINE-sized time axes."""
import glob
import os
import random

import numpy as np
import pytest
import torch

from decisions import FLAT_TOL, TENSOR_TOL, assert_flips_are_ties, assert_grad_parity, gpu_decisions
from util import FWD_TOL, maxabs, rel, stats, sub

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(autouse=True, params=["h2", "auto"])
def conv_path(request):
    """Every model-level test runs with the split-fp16 convs FORCED wherever their shape rules hold (`h2`: the arithmetic of the
    full-size step, at the fixtures' small batches) and with the product's own choice (`auto`: batches too small to fill the chip
    stay on the fp32 Winograd kernels, ops._h2_fills).  Tests at full size run once: there both are the same path."""
    from electrocardio_panorama_amd import ops as o
    name = request.node.name
    full = any(k in name for k in ("full_size", "full_baseline", "config3", "config4", "long_sequences", "main_entry"))
    if request.param == "auto" and full:
        pytest.skip("full-size: `auto` is the split-fp16 path already")
    saved = o._H2_MIN_WGS
    o._H2_MIN_WGS = 0 if request.param == "h2" else saved
    yield request.param
    o._H2_MIN_WGS = saved
# Caps on the measured allowance of the fixture-gradient checks (test_train_golden, test_nefnet2_golden).  The allowance is the
# distance between the decision-replaying fp64 oracle and the fixture (the reference's own fp32 run); it is large (up to ~1.9e-3 on
# the flat gradient of these tiny shapes, tools/debug_tie.py) exactly when a ReLU / L1 argument sits within fp32 round-off of its
# switching point and the REFERENCE resolved it differently from exact arithmetic.  Two ways to get there: the HIP path resolved
# it like the reference (then the oracle replays a flip, which assert_flips_are_ties certifies as a tie), or the HIP path resolved
# it like exact arithmetic (no flip to replay: the split-fp16 convs land there on nefnet2_B3_V1_L1000_Q3 -- HIP equals the fp64
# oracle to the fixed bars of oracle_replaying() and the fixture is the one that is 1.7e-3 away).  Either way the HIP path is held to
# the fixed bars against the oracle FIRST; the fixture comparison then gets the measured distance, capped so that nothing can
# widen its own bar beyond what a tie moves.
SLACK_CAPS = (3e-3, 1e-2)       # (flat, per tensor); largest seen: 1.87e-3 flat (nefnet2_B3_V1_L1000_Q3, 3 ties)


def slack_caps(dec):
    return SLACK_CAPS


TIE_FREE_FACTS = []      # (fixture, HIP-to-fixture rel-L2 on the flat statistic) of every tie-free fixture compared so far


def _report_tie_free():
    """One line in the run's `---- parity facts ----` tail (latest state: the line is replaced, not repeated)."""
    import conftest
    if not TIE_FREE_FACTS:
        return
    names = sorted({n for n, _ in TIE_FREE_FACTS})
    worst = max(TIE_FREE_FACTS, key=lambda t: t[1])
    conftest.REPORT[:] = [l for l in conftest.REPORT if not l.startswith("tie-free fixtures:")]
    conftest.report(f"tie-free fixtures: {len(names)} (HIP vs the reference's own gradients at the plain bars, zero allowance, no replayed "
                    f"decision, both conv paths), worst flat {worst[1]:.2e} <= {FLAT_TOL:g} ({worst[0]})")


class Cfg(dict):
    __getattr__ = dict.__getitem__


def make_cfg(V, reg="l1_loss", lr=0.1, noise=False):
    return Cfg(MODEL=Cfg(model="model_nefnet", theta_L=1, loss="v1", resume=""),
               DATA=Cfg(lead_num=V, noise=noise),
               SOLVER=Cfg(optim="sgd", lr=lr, scheduler="MultiStep", lr_step=[50, 100], reg_loss=reg,
                          loss_using=[1, 2, 3], loss_factor=[0.5, 0.5, 1], epochs=1),
               output_dir="/tmp/nef_test", desc="debug")


def hashed_model(V):
    from electrocardio_panorama_amd.network import build_model
    from oracle import hashweights as hw
    m = build_model(make_cfg(V)).float()
    m.load_state_dict({**hw.hashed_params(V), **hw.hashed_buffers()})
    return m.to(DEV)


def batch_t(B, V, L, seed, Q=0, dev=DEV):
    from electrocardio_panorama_amd import synth
    b = synth.make_batch(B, V, L, seed=seed, Q=Q)
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in b.items()}


def golden(golden_dir, pattern):
    files = sorted(glob.glob(os.path.join(golden_dir, pattern)))
    assert files, pattern
    return files


def test_state_dict_surface():
    from oracle import nefnet_oracle as orc
    m = hashed_model(3)
    exp = {**orc.param_shapes(3), **orc.buffer_shapes()}
    sd = m.state_dict()
    assert set(sd) == set(exp)
    assert all(tuple(sd[k].shape) == tuple(exp[k]) for k in sd)
    with pytest.raises(KeyError):
        b = batch_t(2, 3, 512, 1)
        m(b["data"], b["input_theta"], b["target_theta"], b["rois"], phase="bogus")
    with pytest.raises(ValueError):
        from electrocardio_panorama_amd.network import build_model
        c = make_cfg(3)
        c.MODEL.model = "nope"
        build_model(c)


def test_eval_golden(golden_dir):
    for f in golden(golden_dir, "eval_*.npz"):
        z = np.load(f)
        B, V, L, Q, seed = (int(z[k]) for k in ("B", "V", "L", "Q", "seed"))
        b = batch_t(B, V, L, seed, Q)
        m = hashed_model(V).eval()
        random.seed(seed)
        out, sp, sl, rest = m(b["data"], b["input_theta"], b["target_theta"], b["rois"], rest_theta=b["rest_theta"],
                              phase="test")
        name = os.path.basename(f)
        for got, key in ((out, "out"), (sp, "shuf_p"), (sl, "shuf_l"), (rest, "rest_out")):
            assert rel(got, z[key]) < FWD_TOL, (name, key, rel(got, z[key]))
        rois_before = b["rois"].clone()
        z1, z2 = m(b["data"], b["input_theta"], b["target_theta"], b["rois"], phase="gen")
        assert torch.equal(rois_before, b["rois"])
        assert z1.shape == (B, 128 * V, L // 4) and z2.shape == (B, 128 * V, 7, 32)
        assert rel(sub(z1), z["z1_sub"]) < FWD_TOL and rel(sub(z2), z["z2_sub"]) < FWD_TOL
        assert rel(stats(z1), z["z1_stats"]) < 1e-5 and rel(stats(z2), z["z2_stats"]) < 1e-5
        gen = m.gen_ecg(z1, z2, b["rest_theta"], b["rois"])
        assert rel(gen, z["gen_ecg"]) < FWD_TOL, (name, "gen_ecg")
        assert not m.training and m.segment_status() == 0


def _train_once(V, B, L, seed, reg, masked):
    from electrocardio_panorama_amd.network import build_loss
    from oracle import hashweights as hw
    cfg = make_cfg(V, reg)
    m = hashed_model(V).train()
    m.keep_saved = True
    if masked:
        m.dropout_masks = {k: v.to(DEV) for k, v in hw.hashed_masks(V, B, L // 4).items()}
    else:
        m.dropout_p = 0.0
    b = batch_t(B, V, L, seed)
    random.seed(seed)
    outs = m(b["data"], b["input_theta"], b["target_theta"], b["rois"], phase="train")
    losses = build_loss(cfg)(outs[0], outs[1], outs[2], b["target_view"].unsqueeze(1), cfg)
    losses[0].backward()
    return m, outs, losses


def test_side_stream_modes_are_bit_identical(monkeypatch):
    """NEF_SIDE_STREAM: weight gradients on a second stream (forced on here; `auto` keeps launch-bound shapes like this
    one on a single stream) or inline -- the same kernels on the same operands, so every gradient is bit-identical."""
    grads = {}
    for mode in ("1", "0", "auto"):
        monkeypatch.setenv("NEF_SIDE_STREAM", mode)
        m, outs, losses = _train_once(3, 3, 512, 11, "l1_loss", True)
        torch.cuda.synchronize()
        grads[mode] = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    for mode in ("0", "auto"):
        assert grads[mode].keys() == grads["1"].keys()
        for k in grads["1"]:
            assert torch.equal(grads[mode][k], grads["1"][k]), (mode, k)


def oracle_replaying(m, outs, b, V, seed, masks=None, dt=torch.float32, reg="l1_loss", model2=False, fold=None,
                     lead_choice=None):
    """Run the CPU oracle on the same inputs while it REPLAYS the discrete decisions (ReLU on/off, L1 signs) the HIP
    path took in `m`'s last forward (tests/decisions.py); assert every decision the oracle would have taken differently
    is a tie; assert the gradients agree at the tie-free bars (flat <= 1e-4, per tensor <= 1e-3).  Returns
    (oracle outputs, oracle losses, Decisions, flat gradient rel-L2)."""
    from oracle import hashweights as hw
    from oracle import nefnet_oracle as orc
    tgt = b["target_view"].unsqueeze(1)
    dec = orc.Decisions(gpu_decisions(m, outs, tgt, keep_masks=masks, reg_l1=(reg == "l1_loss"), fold=fold))
    src = hw.hashed_params2() if model2 else hw.hashed_params(V)
    P = orc.require_grad({k: v.to(dt) for k, v in src.items()})
    Bf = {k: (v.to(dt) if v.dtype.is_floating_point else v) for k, v in hw.hashed_buffers().items()}
    random.seed(seed)
    fwd = orc.forward2 if model2 else orc.forward
    ref = fwd(P, Bf, b["data"].to(dt), b["input_theta"].to(dt), b["target_theta"].to(dt), b["rois"], phase="train",
              training=True, masks=masks, p=0.2 if masks is not None else 0.0, dec=dec, lead_choice=lead_choice)
    rl = orc.loss_v1(ref[0], ref[1], ref[2], tgt.to(dt), reg_loss=reg, dec=dec)
    rl[0].backward()
    assert_flips_are_ties(dec)
    named = {k: p.grad for k, p in m.named_parameters()}
    flat, worst = assert_grad_parity(named, P, dead=orc.DEAD_PARAMS, n_terms=3.0 * b["data"].shape[0] * b["data"].shape[2])
    dec.oracle_buffers, dec.oracle_params = Bf, P
    return ref, rl, dec, flat


# Gradient bars (tests/decisions.py): flat gradient <= 1e-4, every tensor <= 1e-3 rel-L2 -- unconditionally.  The one
# legitimate source of larger deviations, a ReLU / L1 argument within fp32 round-off of its switching point (one such tie
# moved the flat gradient by 1.3e-3 at these tiny shapes in round 1, tools/debug_tie.py), is handled by making the
# oracle replay the HIP path's decisions and asserting that each replayed difference IS a tie.


def test_train_golden(golden_dir):
    """Outputs, losses, BN buffers against the reference's own fixtures; gradients against the fixtures when the step
    had no tie, and against the decision-replaying oracle always."""
    from oracle import hashweights as hw
    from oracle import nefnet_oracle as orc
    n_fixture_grad = 0
    for f in golden(golden_dir, "train_*.npz"):
        z = np.load(f)
        B, V, L, seed = (int(z[k]) for k in ("B", "V", "L", "seed"))
        name = os.path.basename(f)
        masked = bool(int(z["masked"]))
        m, outs, losses = _train_once(V, B, L, seed, str(z["reg"]), masked)
        for got, key in zip(outs, ("out", "shuf_p", "shuf_l")):
            assert rel(got, z[key]) < FWD_TOL, (name, key, rel(got, z[key]))
        assert maxabs(torch.stack([l_.detach() for l_ in losses]), z["losses"]) < 1e-6, name
        masks = hw.hashed_masks(V, B, L // 4) if masked else None
        _, _, dec, _ = oracle_replaying(m, outs, batch_t(B, V, L, seed, dev="cpu"), V, seed, masks=masks, reg=str(z["reg"]),
                                        dt=torch.float64)
        # The reference's own (fp32) gradients, for EVERY fixture: |hip - fixture| <= |hip - x64| + |x64 - fixture| with x64
        # the decision-replaying fp64 oracle.  The second term is measured, not assumed: it holds the fixture's own
        # distance from exact arithmetic (up to 2e-4 on this statistic) and, in a step with a ReLU / L1 tie, the effect
        # of the decisions the reference took differently -- so no fixture is skipped.
        # TIE-FREE fixtures (round 6; `tie_free` = 1: seeds on which the reference's fp32 run, its fp64 restatement and -- screened on
        # the GPU, tests/screen_tie_free.py -- the HIP path take identical decisions): the reference's own gradients at the PLAIN
        # bars, zero allowance, and the replaying oracle must have had nothing to replay
        tie_free = "tie_free" in z.files and bool(int(z["tie_free"]))
        nsub = int(z["nsub"]) if "nsub" in z.files else 256
        if tie_free:
            assert dec.total_flips() == 0, (name, "a tie-free fixture saw replayed decisions", dec.total_flips())
        if True:
            n_fixture_grad += 1
            sq, sq64, got_all, ref_all, x64_all = 0.0, 0.0, [], [], []
            SLACK_CAP_FLAT, SLACK_CAP_TENSOR = (0.0, 0.0) if tie_free else slack_caps(dec)
            for k, p in m.named_parameters():
                if k in orc.DEAD_PARAMS:
                    assert p.grad is None, k
                    continue
                ref_sub = z["gsub:" + k]
                if k.endswith("double_conv.0.bias") or k.endswith("double_conv.3.bias"):
                    assert maxabs(sub(p.grad, nsub), ref_sub) < 1e-6, (name, k)      # analytically zero
                    continue
                x64 = sub(dec.oracle_params[k].grad, nsub)
                # bar = the fixed tolerance + the MEASURED distance between the decision-replaying fp64 oracle and the fixture,
                # capped: a GPU-side decision bug must not be able to widen its own bar (the replaying oracle follows the
                # GPU's decisions); the uncapped, fixed-bar check against that oracle is oracle_replaying() above
                slack = min(rel(x64, ref_sub), SLACK_CAP_TENSOR)
                assert rel(sub(p.grad, nsub), ref_sub) < TENSOR_TOL + slack, (
                    name, k, rel(sub(p.grad, nsub), ref_sub), f"bar {TENSOR_TOL} + measured oracle-to-fixture distance {slack:.2e}")
                w_k = (p.grad.numel() / len(ref_sub)) ** 0.5        # the fixture holds <= 256 entries per tensor:
                got_all.append(w_k * sub(p.grad, nsub))               # weight them back to the tensor's size, so the
                ref_all.append(w_k * ref_sub)                        # statistic estimates the FLAT gradient's rel-L2
                x64_all.append(w_k * x64)
                sq += float((p.grad.double() ** 2).sum())
                sq64 += float((dec.oracle_params[k].grad.double() ** 2).sum())
            ref_norm = float(z["flat_grad_norm"])       # same rule for the norm: bar + the replaying oracle's own distance
            nslack = min(abs(sq64 ** 0.5 - ref_norm), SLACK_CAP_FLAT * ref_norm)
            assert abs(sq ** 0.5 - ref_norm) < FLAT_TOL * ref_norm + nslack, (
                name, f"bar {FLAT_TOL} + measured oracle-to-fixture distance {nslack / ref_norm:.2e} (capped at {SLACK_CAP_FLAT})")
            ref_cat = np.concatenate(ref_all)
            fslack = min(rel(np.concatenate(x64_all), ref_cat), SLACK_CAP_FLAT)
            assert rel(np.concatenate(got_all), ref_cat) < FLAT_TOL + fslack, (
                name, f"bar {FLAT_TOL} + measured oracle-to-fixture distance {fslack:.2e} (capped at {SLACK_CAP_FLAT})")
            if tie_free:
                TIE_FREE_FACTS.append((name, rel(np.concatenate(got_all), ref_cat)))
            if dec.total_flips() > 0 or fslack > 3e-4:
                import conftest
                conftest.report(f"{name}: {dec.total_flips()} replayed tie(s); replaying fp64 oracle to fixture {fslack:.2e}, "
                                f"HIP to fixture {rel(np.concatenate(got_all), ref_cat):.2e} on the flat statistic")
        sd = m.state_dict()
        for k in sd:
            if "running" in k:
                assert rel(sd[k], z["buf:" + k]) < 1e-5, (name, k)
            if k.endswith("num_batches_tracked"):
                assert int(sd[k]) == 3, k
    assert n_fixture_grad == len(golden(golden_dir, "train_*.npz")) >= 1
    _report_tie_free()


def test_train_vs_oracle_live():
    """Full flat gradient against the fp64 oracle run on the host (dropout masks replayed), several seeds."""
    from oracle import hashweights as hw
    B, V, L = 2, 3, 520
    for seed in (31, 32, 33, 34):
        m, outs, losses = _train_once(V, B, L, seed, "l1_loss", True)
        b = batch_t(B, V, L, seed, dev="cpu")
        ref, rl, dec, flat = oracle_replaying(m, outs, b, V, seed, masks=hw.hashed_masks(V, B, L // 4), dt=torch.float64)
        for a, r in zip(outs, ref):
            assert rel(a, r) < FWD_TOL
        assert flat < 2e-5, flat         # fp32 round-off against fp64 once ties are out of the picture


@pytest.mark.parametrize("graph", [False, True], ids=["eager", "graphed"])
def test_sgd_steps_golden(golden_dir, graph):
    """Three iterations of Solver.run_one_epoch(phase='train') vs the reference Solver's trajectory -- eagerly, and through
    the captured hipGraph the Solver uses at launch-bound shapes (cfg.SOLVER.graph; FusedSGD's flat buffers stepped)."""
    from electrocardio_panorama_amd import synth
    from electrocardio_panorama_amd.solver import Solver
    from electrocardio_panorama_amd.solver.optim_scheduler import get_optimizer
    from oracle import hashweights as hw
    from oracle import nefnet_oracle as orc
    z = np.load(golden(golden_dir, "sgd_*.npz")[0])
    B, V, L, seed, steps = (int(z[k]) for k in ("B", "V", "L", "seed", "steps"))
    cfg = make_cfg(V, lr=float(z["lr"]))
    cfg.SOLVER["graph"] = bool(graph)
    sol = Solver(cfg, use_tensorboardx=False)
    sol.model.load_state_dict({**hw.hashed_params(V), **hw.hashed_buffers()})
    sol.model.dropout_p = 0.0
    batches = [synth.make_batch(B, V, L, seed=seed + s, Q=2) for s in range(steps)]
    opt = get_optimizer(cfg, sol.model.parameters())
    random.seed(seed)
    losses = sol.run_one_epoch(batches, "train", opt, collect_views=not graph)[0]
    assert (getattr(sol, "_graph_stepper", None) is not None) == bool(graph)
    if graph:      # the momentum lives in the optimiser's state (checkpoints see it), not in a private buffer
        assert sol._graph_stepper.calls == steps and sol._graph_stepper.flat_buf is opt._flat[0]["buf"]
    assert np.abs(np.array(losses) - z["losses"]).max() < 2e-5, (losses, z["losses"])
    sd = sol.model.state_dict()
    worst = (0.0, None)
    for k in orc.param_shapes(V):
        tol = 1e-6 if k in orc.DEAD_PARAMS else 2e-4
        e = rel(sub(sd[k], 128), z["psub:" + k])
        assert e < tol, (k, e)
        if k not in orc.DEAD_PARAMS and e > worst[0]:
            worst = (e, k)
    import conftest
    conftest.report(f"3-step SGD trajectory vs the reference Solver ({'graphed' if graph else 'eager'}): worst parameter "
                    f"{worst[1]} rel-L2 {worst[0]:.2e} (bar 2e-4)")
    with open(os.path.join(os.environ.get("NEF_TEST_LOG_DIR", "/tmp"), "sgd_trajectory.txt"), "w") as fh:
        fh.write(f"3-step SGD trajectory vs the reference Solver: worst parameter {worst[1]} rel-L2 {worst[0]:.2e} (bar 2e-4)\n")
    for k in orc.buffer_shapes():
        if "running" in k:
            assert rel(sd[k], z["buf:" + k]) < 1e-4, k
    assert int(sd["decoder.1.double_conv.1.num_batches_tracked"]) == 3 * steps


def test_dropout_rng_train_mode_runs_and_is_seeded():
    torch.manual_seed(7)
    m = hashed_model(3).train()
    b = batch_t(2, 3, 512, 5)
    random.seed(1)
    a = m(b["data"], b["input_theta"], b["target_theta"], b["rois"], phase="train")
    m._drop_calls = 0
    random.seed(1)
    c = m(b["data"], b["input_theta"], b["target_theta"], b["rois"], phase="train")
    assert all(torch.equal(x.detach(), y.detach()) for x, y in zip(a, c))
    random.seed(1)
    d = m(b["data"], b["input_theta"], b["target_theta"], b["rois"], phase="train")
    assert not torch.equal(a[0].detach(), d[0].detach())
    assert all(torch.isfinite(x).all() for x in d)


def test_properties_at_long_sequences():
    """Size-independent properties on the BASELINE time axis (L=5000, V=3): eval outputs do not depend on the other
    samples of the batch; a train step is run-to-run deterministic; the loss follows its directional derivative."""
    from electrocardio_panorama_amd.network import build_loss
    V, L, B = 3, 5000, 6
    m = hashed_model(V).eval()
    b = batch_t(B, V, L, 77)
    random.seed(3)
    full = m(b["data"], b["input_theta"], b["target_theta"], b["rois"], phase="train")
    random.seed(3)
    one = m(b["data"][4:5], b["input_theta"][4:5], b["target_theta"][4:5], b["rois"][4:5], phase="train")
    for a, c in zip(full, one):
        assert a.shape == (B, 1, L) and rel(a[4:5], c) < 1e-6
    cfg = make_cfg(V, reg="l2_loss")
    cfg.SOLVER.loss_using = [3]       # the Standin terms stop the gradient through `out`, so only the reconstruction
    lossf = build_loss(cfg)           # term is the derivative of the value it reports

    def step(model):
        random.seed(9)
        model.zero_grad()
        outs = model(b["data"], b["input_theta"], b["target_theta"], b["rois"], phase="train")
        ls = lossf(outs[0], outs[1], outs[2], b["target_view"].unsqueeze(1), cfg)
        ls[0].backward()
        return ls[0].detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}

    m.train()
    m.dropout_p = 0.0
    l1, g1 = step(m)
    l2, g2 = step(m)
    assert torch.equal(l1, l2) and all(torch.equal(g1[k], g2[k]) for k in g1)
    # directional derivative along the normalised gradient of the dominant encoder weight
    k = "W_encoder.layer1.1.conv2.weight"
    p = dict(m.named_parameters())[k]
    d = g1[k] / g1[k].norm()
    eps = 5e-3
    with torch.no_grad():
        p.add_(d, alpha=eps)
    lp, _ = step(m)
    with torch.no_grad():
        p.add_(d, alpha=-2 * eps)
    lm, _ = step(m)
    fd = float(lp - lm) / (2 * eps)
    an = float((g1[k] * d).sum())
    assert abs(fd - an) < 0.05 * abs(an) + 1e-6, (fd, an)


def test_full_baseline_batch_vs_oracle_rows():
    """BASELINE configs[1] at its FULL size (256 samples x 3 leads x 5000) on the GPU; the CPU oracle cannot run that
    batch in seconds, but in eval mode samples are independent, so the oracle decodes three rows of it (first, middle,
    last) and those rows of the batch-256 result must match at the forward bar.  The full-size train step is also run
    twice from the same state: bit-identical losses and gradients (dropout by replayed RNG seed)."""
    from electrocardio_panorama_amd.network import build_loss
    from oracle import hashweights as hw
    from oracle import nefnet_oracle as orc
    B, V, L = 256, 3, 5000
    m = hashed_model(V).eval()
    b = batch_t(B, V, L, 314)
    random.seed(5)
    with torch.no_grad():
        outs = m(b["data"], b["input_theta"], b["target_theta"], b["rois"], phase="train")
    assert all(o.shape == (B, 1, L) and bool(torch.isfinite(o).all()) for o in outs)
    rows = [0, 131, 255]
    P, Bf = hw.hashed_params(V), hw.hashed_buffers()
    random.seed(5)
    c1, c2 = random.randint(0, V - 1), random.randint(0, V - 1)
    idx = torch.tensor(rows)
    cpu = {k: v[idx.to(v.device)].cpu() for k, v in b.items()}
    with torch.no_grad():
        ref = orc.forward(P, Bf, cpu["data"], cpu["input_theta"], cpu["target_theta"], cpu["rois"], phase="train",
                          training=False, lead_choice=(c1, c2))
    for got, want in zip(outs, ref):
        assert rel(got[idx.to(got.device)], want) < FWD_TOL, rel(got[idx.to(got.device)], want)
    assert m.segment_status() == 0
    # full-size train step, twice
    cfg = make_cfg(V)
    lossf = build_loss(cfg)
    m.train()

    def step():
        torch.manual_seed(11)
        m._drop_calls = 0
        random.seed(9)
        m.zero_grad()
        o = m(b["data"], b["input_theta"], b["target_theta"], b["rois"], phase="train")
        ls = lossf(o[0], o[1], o[2], b["target_view"].unsqueeze(1), cfg)
        ls[0].backward()
        return torch.stack([x.detach() for x in ls]), torch.cat([p.grad.reshape(-1) for p in m.parameters() if p.grad is not None])

    l1, g1 = step()
    l2, g2 = step()
    assert torch.equal(l1, l2) and torch.equal(g1, g2)
    assert bool(torch.isfinite(g1).all()) and float(g1.norm()) > 0


def test_config3_shard_full_size_is_deterministic_and_fits():
    """BASELINE configs[2]: the per-GPU shard of the 8-GPU run (256 samples x 8 leads x 5000, Standin losses and dropout
    on) -- the train step runs, is finite, bit-identical when repeated from the same state, and its memory footprint is
    reported (the CPU oracle covers this shape at small batch in test_baseline_config_shapes_vs_oracle and with two
    ranks in tests/test_dp_gpu.py)."""
    from electrocardio_panorama_amd.network import build_loss
    B, V, L = 256, 8, 5000
    cfg = make_cfg(V)
    m = hashed_model(V).train()
    lossf = build_loss(cfg)
    b = batch_t(B, V, L, 2718)
    torch.cuda.reset_peak_memory_stats()

    def step():
        torch.manual_seed(11)
        m._drop_calls = 0
        random.seed(9)
        m.zero_grad()
        o = m(b["data"], b["input_theta"], b["target_theta"], b["rois"], phase="train")
        ls = lossf(o[0], o[1], o[2], b["target_view"].unsqueeze(1), cfg)
        ls[0].backward()
        return torch.stack([x.detach() for x in ls]), torch.cat([p.grad.reshape(-1) for p in m.parameters() if p.grad is not None])

    l1, g1 = step()
    l2, g2 = step()
    assert torch.equal(l1, l2) and torch.equal(g1, g2)
    assert g1.numel() == 18831041 and bool(torch.isfinite(g1).all()) and float(g1.norm()) > 0 and m.segment_status() == 0
    gib = torch.cuda.max_memory_reserved() / 2 ** 30
    print(f"configs[2] shard (256 x 8 x 5000): peak reserved {gib:.1f} GiB, peak allocated "
          f"{torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB")
    assert gib < 200


@pytest.mark.parametrize("B", [256], ids=["B256"])
def test_full_size_train_gradients_vs_oracle(record_property, B):
    """BASELINE configs[1] at FULL size in TRAIN mode (256 x 3 x 5000, dropout masks replayed, batch-statistics
    BatchNorm over 3.84 M elements per channel, split-K weight gradients over 320 k columns): outputs, losses, BN running
    statistics and EVERY gradient tensor against the CPU oracle on the same batch, at the tie-free bars.  The oracle
    needs ~40 GB of host memory for this batch; a host with less than 90 GB available SKIPS (it does not shrink the batch:
    the test id says which size ran, and the run's tail repeats it with the measured numbers)."""
    import conftest
    import psutil
    from electrocardio_panorama_amd.network import build_loss
    from oracle import hashweights as hw
    V, L, seed = 3, 5000, 271
    avail = psutil.virtual_memory().available
    if avail < 90e9:
        conftest.report(f"full-size train parity (B={B}): SKIPPED, host has {avail / 1e9:.0f} GB available, the oracle needs 90")
        pytest.skip(f"the CPU oracle needs ~90 GB of host memory at B={B}; {avail / 1e9:.0f} GB available")
    record_property("batch", B)            # junit / --report-log: which size really ran (256 = configs[1])
    T, C = L // 4, 128 * V
    g = torch.Generator().manual_seed(seed)
    shapes = {"W_encoder.layer1.0": (B, C, T), "W_encoder.layer1.1": (B, C, T), "W_encoder.layer1.2": (B, C, T),
              "w_conv.0": (B, C, T), "z1_conv.0": (B, C, T), "z2_conv1.0": (B, C, T), "z2_conv2.0": (B, 7 * C, 16),
              "z2_conv2.2": (B, 7 * C, 32)}
    masks = {k: (torch.rand(sh, generator=g) >= 0.2).to(torch.uint8) for k, sh in shapes.items()}
    cfg = make_cfg(V)
    m = hashed_model(V).train()
    m.keep_saved = True
    m.dropout_masks = {k: v.to(DEV) for k, v in masks.items()}
    b = batch_t(B, V, L, seed, dev="cpu")
    bd = {k: v.to(DEV) for k, v in b.items()}
    random.seed(seed)
    outs = m(bd["data"], bd["input_theta"], bd["target_theta"], bd["rois"], phase="train")
    losses = build_loss(cfg)(outs[0], outs[1], outs[2], bd["target_view"].unsqueeze(1), cfg)
    losses[0].backward()
    torch.set_num_threads(min(16, os.cpu_count() or 1))      # the host's measured optimum (profiles/r02_cpu_thread_sweep.md)
    ref, rl, dec, flat = oracle_replaying(m, outs, b, V, seed, masks=masks)
    for a, r in zip(outs, ref):
        assert rel(a, r) < FWD_TOL, (f"batch {B}", rel(a, r))
    assert maxabs(torch.stack([l_.detach() for l_ in losses]), torch.stack([r.detach() for r in rl])) < 1e-6, f"batch {B}"
    sd = m.state_dict()
    for k, v in dec.oracle_buffers.items():
        if "running" in k:
            assert rel(sd[k], v) < 1e-5, (f"batch {B}", k)
    record_property("flat_gradient_rel_l2", float(flat))
    record_property("replayed_ties", int(dec.total_flips()))
    msg = (f"full-size train parity: B={B} (configs[1] = 256), flat gradient rel-L2 {flat:.2e} (bar {FLAT_TOL}), "
           f"replayed ties {dec.total_flips()}")
    print(msg)
    conftest.report(msg)
    with open(os.path.join(os.environ.get("NEF_TEST_LOG_DIR", "/tmp"), "full_size_parity.txt"), "w") as fh:
        fh.write(msg + "\n")
    m.last_saved = None


@pytest.mark.parametrize("B,V,L,Q,phase", [
    (4, 1, 2048, 0, "train"),      # BASELINE configs[0] shape: batch 4, 1 lead, len 2048
    (2, 8, 5000, 0, "train"),      # configs[2] shape (Tianchi 8-lead, len 5000), small batch
    (2, 1, 512, 360, "test"),      # configs[3] shape: 1 view in -> 360 queried angles
    (2, 3, 5000, 12, "test"),      # configs[4]-like: long sequences through the sweep / gen_ecg path
])
def test_baseline_config_shapes_vs_oracle(B, V, L, Q, phase):
    """The BASELINE.json configurations, at batch sizes the CPU oracle finishes in seconds: outputs (and for the train
    shapes the flat gradient) against the oracle run live on the same seeded inputs."""
    from oracle import hashweights as hw
    from oracle import nefnet_oracle as orc
    seed = 100 + V + Q
    b = batch_t(B, V, L, seed, Q, dev="cpu")
    if phase == "train":
        m, outs, losses = _train_once(V, B, L, seed, "l1_loss", False)
        ref, rl, dec, flat = oracle_replaying(m, outs, b, V, seed)
        for a, r in zip(outs, ref):
            assert rel(a, r) < FWD_TOL
        assert maxabs(torch.stack([l_.detach() for l_ in losses]), torch.stack([r.detach() for r in rl])) < 1e-6
    else:
        m = hashed_model(V).eval()
        bd = {k: v.to(DEV) for k, v in b.items()}
        random.seed(seed)
        outs = m(bd["data"], bd["input_theta"], bd["target_theta"], bd["rois"], rest_theta=bd["rest_theta"], phase="test")
        P, Bf = hw.hashed_params(V), hw.hashed_buffers()
        with torch.no_grad():
            random.seed(seed)
            ref = orc.forward(P, Bf, b["data"], b["input_theta"], b["target_theta"], b["rois"], rest_theta=b["rest_theta"],
                              phase="test", training=False)
        assert outs[3].shape == (B, Q, L)
        for a, r in zip(outs, ref):
            assert rel(a, r) < FWD_TOL, rel(a, r)
        z1, z2 = m(bd["data"], bd["input_theta"], bd["target_theta"], bd["rois"], phase="gen")
        gen = m.gen_ecg(z1, z2, bd["rest_theta"], bd["rois"])
        assert rel(gen, ref[3]) < FWD_TOL


@pytest.mark.parametrize("V", [2, 4, 5, 9, 12])
def test_every_supported_lead_count_vs_oracle(V):
    """The input-lead schemes the reference's datasets can emit (lead_num 1/2/3/4/5/8/9/12: tianchi.py:127-190); 1, 3 and
    8 are covered by the fixtures above, the rest here: train-phase outputs, losses and the flat gradient (dropout
    off), ragged length 520 (T=130: one full column tile + 2)."""
    test_baseline_config_shapes_vs_oracle(2, V, 520, 0, "train")


def test_solver_test_phase_vs_oracle():
    """Solver.run_one_epoch(phase='test') (reference solver.py:190-230): five losses incl. loss_unsperv on the last four
    rest views, PSNR/SSIM bookkeeping; values against the oracle on the same batches."""
    from electrocardio_panorama_amd import synth
    from electrocardio_panorama_amd.solver import Solver
    from electrocardio_panorama_amd.utils.metric import PSNR, SSIM
    from oracle import hashweights as hw
    from oracle import nefnet_oracle as orc
    V, B, L, Q = 3, 3, 512, 6
    cfg = make_cfg(V)
    cfg.DATA["super_mode"] = "IIv2v5_v4I_372"        # gen_num = 2 (reference solver.py:198-199)
    cfg.DATA["dataset"] = "tianchi"
    sol = Solver(cfg, use_tensorboardx=False)
    sol.model.load_state_dict({**hw.hashed_params(V), **hw.hashed_buffers()})
    batches = [synth.make_batch(B, V, L, seed=40 + s, Q=Q) for s in range(2)]
    random.seed(5)
    losses, rest_views, predict_views, _, mertics_all, _, single = sol.run_one_epoch(batches, "test")
    assert len(losses) == 2 and len(losses[0]) == 5 and len(mertics_all) == 2 and len(single) == 2 and len(single[0]) == 2
    P, Bf = hw.hashed_params(V), hw.hashed_buffers()
    random.seed(5)
    for i, meta in enumerate(batches):
        b = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in meta.items()}
        with torch.no_grad():
            o = orc.forward(P, Bf, b["data"], b["input_theta"], b["target_theta"], b["rois"], rest_theta=b["rest_theta"],
                            phase="test", training=False)
            ref = orc.loss_v1(o[0], o[1], o[2], b["target_view"].unsqueeze(1), rest_out=o[3][:, -4:],
                              rest_view=b["rest_view"][:, -4:].float())
        assert np.abs(np.array(losses[i]) - np.array([float(v) for v in ref])).max() < 2e-6
        ro, rv, rn = o[3].numpy(), meta["rest_view"], meta["rois"]
        want = [PSNR(ro[:, -2:], rv[:, -2:], rn), PSNR(ro[:, :-2], rv[:, :-2], rn), SSIM(ro[:, -2:], rv[:, -2:], rn),
                SSIM(ro[:, :-2], rv[:, :-2], rn)]
        assert np.abs(np.array(mertics_all[i]) - np.array(want)).max() < 1e-3
    assert len(predict_views) == 2 * B and predict_views[0].shape == (Q, L)


def test_view_metrics_kernel_vs_host():
    """nef_view_metrics (per-row PSNR / SSIM on the un-padded region) against the host restatement of mertic.py."""
    from electrocardio_panorama_amd import ops as o
    from electrocardio_panorama_amd.utils.metric import PSNR, SSIM
    rng = np.random.default_rng(3)
    B, Q, L = 5, 6, 512
    gt = rng.random((B, Q, L)).astype(np.float32)
    pred = (gt + 0.05 * rng.standard_normal((B, Q, L))).astype(np.float32)
    pred[1, 2] = gt[1, 2]                                        # exact match -> PSNR 100, SSIM 1
    rois = np.zeros((B, 7, 2), np.int64)
    rois[:, -1, 0] = [512, 300, 7, 333, 64]
    ps, ss = o.view_metrics(torch.from_numpy(pred).to(DEV), torch.from_numpy(gt).to(DEV), torch.from_numpy(rois).to(DEV))
    ps, ss = ps.cpu().numpy(), ss.cpu().numpy()
    assert ps[1, 2] == 100.0 and abs(ss[1, 2] - 1.0) < 1e-12
    for i in range(B):
        for j in range(Q):
            assert abs(ps[i, j] - PSNR(pred[i:i + 1, j:j + 1], gt[i:i + 1, j:j + 1], rois[i:i + 1])) < 1e-9
            assert abs(ss[i, j] - SSIM(pred[i:i + 1, j:j + 1], gt[i:i + 1, j:j + 1], rois[i:i + 1])) < 1e-9
    assert abs(ps.mean() - PSNR(pred, gt, rois)) < 1e-9 and abs(ss[:, -2:].mean() - SSIM(pred[:, -2:], gt[:, -2:], rois)) < 1e-9
    ps2, ss2 = o.view_metrics(torch.from_numpy(pred).to(DEV), torch.from_numpy(gt).to(DEV))
    assert abs(ps2.cpu().numpy().mean() - PSNR(pred, gt)) < 1e-9
    rois[0, -1, 0] = 5                                            # shorter than the SSIM window
    _, ss3 = o.view_metrics(torch.from_numpy(pred).to(DEV), torch.from_numpy(gt).to(DEV), torch.from_numpy(rois).to(DEV))
    assert bool(torch.isnan(ss3[0]).all()) and not bool(torch.isnan(ss3[1:]).any())


def test_solver_val_loads_checkpoints_and_reproduces_oracle_metrics(tmp_path):
    """Solver.val (reference solver.py:118-137): epoch=-1 loads best_valid.pkl, epoch=n loads epoch_n.pkl even when
    `last_checkpoint` points elsewhere (checkpointer.py:49-60); PSNR / SSIM equal the oracle's on the same batches."""
    from electrocardio_panorama_amd import synth
    from electrocardio_panorama_amd.solver import Solver
    from electrocardio_panorama_amd.utils import CheckPointer
    from electrocardio_panorama_amd.utils.metric import PSNR, SSIM
    from oracle import hashweights as hw
    from oracle import nefnet_oracle as orc
    V, B, L, Q = 3, 3, 512, 6
    cfg = make_cfg(V)
    cfg["output_dir"], cfg["desc"] = str(tmp_path), "debug"
    good = Solver(cfg, use_tensorboardx=False)
    good.model.load_state_dict({**hw.hashed_params(V), **hw.hashed_buffers()})
    ck = CheckPointer(good.model, save_dir=good.output_dir)
    ck.save("epoch_7", epoch=7, psnr_gen=1.0, psnr_reg=2.0)
    ck.save("best_valid", epoch=7, best_test_psnr_gen=1.0)
    other = Solver(cfg, use_tensorboardx=False)                   # a different (randomly initialised) model saved LAST
    CheckPointer(other.model, save_dir=other.output_dir).save("epoch_8", epoch=8)
    assert open(os.path.join(good.output_dir, "last_checkpoint")).read().strip().endswith("epoch_8.pkl")
    batches = [synth.make_batch(B, V, L, seed=60 + s, Q=Q) for s in range(2)]
    P, Bf = hw.hashed_params(V), hw.hashed_buffers()
    want = []
    random.seed(5)
    for meta in batches:
        b = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in meta.items()}
        with torch.no_grad():
            o = orc.forward(P, Bf, b["data"], b["input_theta"], b["target_theta"], b["rois"], rest_theta=b["rest_theta"],
                            phase="test", training=False)
        ro, rv, rn = o[3].numpy(), meta["rest_view"], meta["rois"]
        want.append([PSNR(ro[:, -4:], rv[:, -4:], rn), PSNR(ro[:, :-4], rv[:, :-4], rn), SSIM(ro[:, -4:], rv[:, -4:], rn),
                     SSIM(ro[:, :-4], rv[:, :-4], rn)])
    want = np.mean(want, axis=0)
    for epoch in (-1, 7):
        fresh = Solver(cfg, use_tensorboardx=False)
        random.seed(5)
        got = fresh.val(batches, epoch=epoch)
        assert np.abs(np.array(got) - want).max() < 1e-3, (epoch, got, want)
    random.seed(5)
    last = Solver(cfg, use_tensorboardx=False).val(batches, epoch=8)
    assert abs(last[0] - want[0]) > 1e-2                          # epoch_8 really is the other model
    with pytest.raises(FileNotFoundError):
        Solver(cfg, use_tensorboardx=False).val(batches, epoch=99)


def test_real_recordings_golden(golden_dir):
    """G7: a real batch (the reference's bundled Tianchi recordings via its own dataset class) against the reference's
    outputs."""
    z = np.load(os.path.join(golden_dir, "real_tianchi_B2_V3.npz"))
    t = {k: torch.from_numpy(z[k]).to(DEV) for k in ("data", "rois", "input_theta", "target_theta", "rest_theta")}
    m = hashed_model(3).eval()
    random.seed(int(z["seed"]))
    outs = m(t["data"], t["input_theta"], t["target_theta"], t["rois"], rest_theta=t["rest_theta"], phase="test")
    for got, key in zip(outs, ("out", "shuf_p", "shuf_l", "rest_out")):
        assert rel(got, z[key]) < FWD_TOL, (key, rel(got, z[key]))
    assert m.segment_status() == 0


def test_graphed_step_matches_eager(golden_dir):
    """hipGraph replay of the whole train step == the eager Solver path, step for step (same Standin draws; dropout off
    so both consume identical arithmetic), and also reproduces the reference Solver's three-step trajectory."""
    from electrocardio_panorama_amd import synth
    from electrocardio_panorama_amd.graph import GraphedTrainStep
    from oracle import hashweights as hw
    from oracle import nefnet_oracle as orc
    z = np.load(golden(golden_dir, "sgd_*.npz")[0])
    B, V, L, seed, steps = (int(z[k]) for k in ("B", "V", "L", "seed", "steps"))
    cfg = make_cfg(V, lr=float(z["lr"]))
    m = hashed_model(V).train()
    m.dropout_p = 0.0
    step = GraphedTrainStep(m, cfg)
    random.seed(seed)
    got = []
    for s in range(steps):
        b = {k: torch.from_numpy(np.ascontiguousarray(v)).to(DEV) for k, v in synth.make_batch(B, V, L, seed=seed + s, Q=2).items()}
        got.append(step(b["data"], b["input_theta"], b["target_theta"], b["rois"], b["target_view"]).cpu().numpy().copy())
    assert np.abs(np.array(got) - z["losses"]).max() < 2e-5, (got, z["losses"])
    sd = m.state_dict()
    for k in orc.param_shapes(V):
        tol = 1e-6 if k in orc.DEAD_PARAMS else 2e-4
        assert rel(sub(sd[k], 128), z["psub:" + k]) < tol, k
    assert int(sd["decoder.1.double_conv.1.num_batches_tracked"]) == 3 * steps
    # dropout on: replays keep running and use a fresh seed each step
    m.dropout_p = 0.2
    step2 = GraphedTrainStep(m, cfg)
    l1 = step2(b["data"], b["input_theta"], b["target_theta"], b["rois"], b["target_view"]).clone()
    l2 = step2(b["data"], b["input_theta"], b["target_theta"], b["rois"], b["target_view"]).clone()
    assert torch.isfinite(l1).all() and torch.isfinite(l2).all() and not torch.equal(l1, l2)


def test_graphed_step_keeps_momentum_across_shapes_and_checkpoints(conv_path):
    """Alternating input shapes (a final partial batch) replay their own captured graphs and share ONE momentum buffer:
    the trajectory equals the eager FusedSGD path step for step; state_dict() / load_state_dict() carry the momentum."""
    import copy
    from electrocardio_panorama_amd import synth
    from electrocardio_panorama_amd.graph import GraphedTrainStep
    from electrocardio_panorama_amd.network import build_loss
    from electrocardio_panorama_amd.solver.optim_scheduler import get_optimizer
    V = 3
    cfg = make_cfg(V, lr=0.05)
    shapes = [(4, 512), (2, 512), (4, 512), (2, 512)]
    batches = [{k: torch.from_numpy(np.ascontiguousarray(v)).to(DEV) for k, v in synth.make_batch(B, V, L, seed=70 + i).items()}
               for i, (B, L) in enumerate(shapes)]
    mg, me = hashed_model(V).train(), hashed_model(V).train()
    mg.dropout_p = me.dropout_p = 0.0
    step = GraphedTrainStep(mg, cfg)
    lossf, optim = build_loss(cfg), get_optimizer(cfg, me.parameters())
    random.seed(3)
    got = [step(b["data"], b["input_theta"], b["target_theta"], b["rois"], b["target_view"]).cpu().numpy().copy()
           for b in batches[:3]]
    assert len(step.slots) == 2
    random.seed(3)
    want = []
    for b in batches[:3]:
        o = me(b["data"], b["input_theta"], b["target_theta"], b["rois"], phase="train")
        ls = lossf(o[0], o[1], o[2], b["target_view"].unsqueeze(1), cfg)
        ls[0].backward()
        optim.step()
        optim.zero_grad()
        want.append(torch.stack([x.detach() for x in ls]).cpu().numpy())
    assert np.abs(np.array(got) - np.array(want)).max() < 1e-6, (got, want)
    pg, pe = dict(mg.named_parameters()), dict(me.named_parameters())
    assert max(rel(pg[k], pe[k]) for k in pg) < 1e-6
    # checkpoint the graphed path after three steps, restore into a fresh stepper on a copy of the model, take step four
    sd = step.state_dict()
    assert sd["momentum_buffer"] is not None and float(sd["momentum_buffer"].abs().sum()) > 0
    sd_model = copy.deepcopy(mg.state_dict())
    m2 = hashed_model(V).train()
    m2.dropout_p = 0.0
    m2.load_state_dict(copy.deepcopy(mg.state_dict()))
    from electrocardio_panorama_amd import ops as _ops
    h2 = copy.deepcopy(mg.h2_state())           # what CheckPointer.save stores next to the state_dict (round 6)
    assert (len(h2["keys"]) > 40) == bool(_ops.H2 and conv_path == "h2")      # (`auto`: these batches stay on the fp32 kernels -- no such sites)
    assert m2.load_h2_state(h2) == len(h2["keys"])
    step2 = GraphedTrainStep(m2, cfg)
    step2.load_state_dict(sd)
    b = batches[3]
    st = random.getstate()
    l_a = step(b["data"], b["input_theta"], b["target_theta"], b["rois"], b["target_view"]).clone()
    random.setstate(st)
    l_b = step2(b["data"], b["input_theta"], b["target_theta"], b["rois"], b["target_view"]).clone()
    # With the operand magnitudes of the split-fp16 call sites restored (Model_nefnet.h2_state / load_h2_state, carried by
    # CheckPointer) the restored stepper splits its operands with the scales the running one uses: the resumed trajectory is the
    # uninterrupted one bit for bit -- losses and every parameter.  (Rounds 4-5: the restored model measured again; a different
    # power-of-two scale wherever a magnitude sat near a binade edge left 1e-6 .. 1e-4 between the two.)
    assert torch.equal(l_a, l_b), (l_a, l_b)
    p2 = dict(m2.named_parameters())
    for k in pg:
        assert torch.equal(p2[k], pg[k]), k
    # ... and WITHOUT it: equal arithmetic at fp32-rounding level, not bit identity
    m3 = hashed_model(V).train()
    m3.dropout_p = 0.0
    m3.load_state_dict(copy.deepcopy(sd_model))
    step3 = GraphedTrainStep(m3, cfg)
    step3.load_state_dict(sd)
    random.setstate(st)
    l_c = step3(b["data"], b["input_theta"], b["target_theta"], b["rois"], b["target_view"]).clone()
    assert rel(l_a, l_c) < 1e-6, (l_a, l_c)
    p3 = dict(m3.named_parameters())
    worst = max(pg, key=lambda k: rel(p3[k], pg[k]))
    print(f"restored WITHOUT the operand magnitudes vs running stepper after one more step: worst parameter {worst} rel {rel(p3[worst], pg[worst]):.2e}")
    assert rel(p3[worst], pg[worst]) < (1e-4 if _ops.H2 else 1e-7), (worst, rel(p3[worst], pg[worst]), float(pg[worst].norm()))
    # a new learning rate does NOT re-capture (the captured SGD launch reads it from a device word) and keeps the momentum; the
    # replayed step applies it: the update of the next step is lr_new / lr_old times what the old rate would have given
    before, graphs = step.flat_buf.clone(), dict(step.slots)
    p_before = step.flat_p.clone()
    st = random.getstate()
    step(b["data"], b["input_theta"], b["target_theta"], b["rois"], b["target_view"])
    d_old = (step.flat_p - p_before).clone()
    step.flat_p.copy_(p_before), step.flat_buf.copy_(before)
    lr_old = step.lr
    step.set_lr(0.01)
    assert step.slots == graphs and torch.equal(step.flat_buf, before)
    random.setstate(st)
    step.calls -= 1
    step(b["data"], b["input_theta"], b["target_theta"], b["rois"], b["target_view"])
    d_new = step.flat_p - p_before
    assert rel(d_new, d_old * (0.01 / lr_old)) < 1e-3      # (differences of fp32 parameters: each update is rounded at the parameter's ulp)


def test_adversarial_rois_full_step_vs_oracle():
    """Zero-length and 1-sample segments, boundaries at every residue mod 4, through a full train step (forward,
    losses, flat gradient) against the oracle; the integer segment table is bit-exact."""
    from electrocardio_panorama_amd import ops as o
    from oracle import hashweights as hw
    from oracle import nefnet_oracle as orc
    V, B, L, seed = 3, 2, 512, 17
    b = batch_t(B, V, L, seed, dev="cpu")
    b["rois"] = torch.tensor([[[0, 1], [1, 2], [2, 7], [7, 9], [9, 250], [250, 251], [251, 512]],
                              [[0, 3], [3, 3], [3, 130], [130, 133], [133, 134], [134, 509], [509, 512]]])
    start, length = o.roi_segment_table(b["rois"].to(DEV))
    rs, rl = orc.roi_segment_table(b["rois"])
    assert torch.equal(start.cpu(), rs) and torch.equal(length.cpu(), rl) and int(rl.sum()) == 2 * (L // 4)
    m = hashed_model(V).train()
    m.dropout_p = 0.0
    m.keep_saved = True
    bd = {k: v.to(DEV) for k, v in b.items()}
    random.seed(seed)
    outs = m(bd["data"], bd["input_theta"], bd["target_theta"], bd["rois"], phase="train")
    from electrocardio_panorama_amd.network import build_loss
    cfg = make_cfg(V)
    losses = build_loss(cfg)(outs[0], outs[1], outs[2], bd["target_view"].unsqueeze(1), cfg)
    losses[0].backward()
    assert m.segment_status() == 0
    ref, rl_, dec, flat = oracle_replaying(m, outs, b, V, seed)
    for a, r in zip(outs, ref):
        assert rel(a, r) < FWD_TOL
    # ROIs that do not tile [0, L] are flagged on the device (the reference would fail in torch.cat)
    bad = bd["rois"].clone()
    bad[0, 3, 1] = 5
    m(bd["data"], bd["input_theta"], bd["target_theta"], bad, phase="train")
    assert m.segment_status() == 1
    with pytest.raises(TypeError):
        m(bd["data"], bd["input_theta"], bd["target_theta"], bd["rois"].float(), phase="train")


def test_main_entry_trains_and_checkpoints(tmp_path):
    """`python -m electrocardio_panorama_amd.main --config-file config/nef-net.yml ...`: the packaged equivalent of the
    reference's main.py runs two epochs on synthetic meta batches, steps the MultiStepLR schedule, writes the reference's
    checkpoint layout and resumes from it."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "electrocardio_panorama_amd.main", "--config-file", "config/nef-net.yml",
           "SOLVER.epochs", "2", "output_dir", str(tmp_path), "DATA.synthetic", "True",
           "DATA.lead_num", "3"]
    env = dict(os.environ, PYTHONPATH=root)
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("Epoch ")]
    assert len(lines) == 2
    l0, l1 = (float(ln.split("train_loss: ")[1].split(",")[0]) for ln in lines)
    assert l1 < l0
    outdir = os.path.join(str(tmp_path), "nef-net", "nef-net")
    ck = torch.load(os.path.join(outdir, "epoch_1.pkl"), map_location="cpu")
    assert {"model", "optimizer", "scheduler", "epoch", "psnr_gen", "psnr_reg"} <= set(ck)      # solver.py:103-107
    # the reference's scalar set (solver.py:82-100) through the fallback writer (no tensorboard in this image)
    import json
    rows = [json.loads(ln) for ln in open(os.path.join(str(tmp_path), "nef-net", "tf_logs", "scalars.jsonl"))]
    tags = {r_["tag"] for r_ in rows}
    assert {'train_loss_all', 'test_loss_all', 'train_loss_1', 'test_loss_1', 'train_loss_2', 'test_loss_2', 'train_3',
            'test_3', 'test_unsuperv', 'psnr_gen', 'psnr_reg', 'ssim_gen', 'ssim_reg', 'psnr_reg_lead_0',
            'ssim_reg_lead_1'} <= tags and {r_["step"] for r_ in rows} == {0, 1}
    # val_net entry (reference val_net.py:9-48): loads best_valid.pkl and prints the four metrics
    rv = subprocess.run([sys.executable, "-m", "electrocardio_panorama_amd.val_net"] + cmd[3:5] + cmd[7:], cwd=root, env=env,
                        capture_output=True, text=True, timeout=600)
    assert rv.returncode == 0 and "psnr_gen:" in rv.stdout and "the latest best_test_psnr_gen" in rv.stdout, \
        rv.stdout[-2000:] + rv.stderr[-2000:]
    from oracle import nefnet_oracle as orc
    assert set(ck["model"]) == set(orc.param_shapes(3)) | set(orc.buffer_shapes())
    assert open(os.path.join(outdir, "last_checkpoint")).read().strip().endswith(".pkl")
    r2 = subprocess.run(cmd[:5] + ["SOLVER.epochs", "3"] + cmd[7:], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-2000:]
    # auto-resume from `last_checkpoint`: like the reference (solver.py:53,62) the saved epoch index is re-run
    assert [ln.split(":")[0] for ln in r2.stdout.splitlines() if ln.startswith("Epoch ")] == ["Epoch 1", "Epoch 2"]


def test_allocator_footprint_is_stable():
    """Steady-state training must not grow the allocator's reserved memory step after step (a regression guard for
    cross-stream lifetime handling in the backward pass)."""
    from electrocardio_panorama_amd.network import build_loss
    from electrocardio_panorama_amd.solver.optim_scheduler import get_optimizer
    V, B, L = 3, 16, 2048
    cfg = make_cfg(V)
    m = hashed_model(V).train()
    lossf, optim = build_loss(cfg), get_optimizer(cfg, m.parameters())
    b = batch_t(B, V, L, 3)
    reserved = []
    for i in range(14):
        out, sp, sl = m(b["data"], b["input_theta"], b["target_theta"], b["rois"], phase="train")
        ls = lossf(out, sp, sl, b["target_view"].unsqueeze(1), cfg)
        ls[0].backward()
        optim.step()
        optim.zero_grad()
        torch.cuda.synchronize()
        reserved.append(torch.cuda.memory_reserved())
    assert reserved[-1] <= reserved[4] * 1.05, reserved
    assert torch.isfinite(ls[0])


def test_rccl_single_rank():
    """The data-parallel collectives on the real backend: torchrun, one rank, backend 'nccl' (RCCL).  Multi-GPU boxes
    are not available to the test suite; rank-count logic is covered by the gloo world-2 test in test_host_cpu.py."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29533", os.path.join(root, "tests", "rccl_worker.py")]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def hashed_model2(V):
    from electrocardio_panorama_amd.network import build_model
    from oracle import hashweights as hw
    c = make_cfg(V)
    c.MODEL.model = "model_nefnet2"
    m = build_model(c).float()
    m.load_state_dict({**hw.hashed_params2(), **hw.hashed_buffers()})
    return m.to(DEV)


def test_nefnet2_golden(golden_dir):
    """f4: Model_nefnet2 (one single-lead encoder shared by all leads + the two single convs) against the reference's
    own outputs, lead means, losses and gradients (tests/golden/nefnet2_*.npz, dropout off)."""
    from electrocardio_panorama_amd.network import build_loss
    from oracle import nefnet_oracle as orc
    n_fixture_grad = 0
    for f in golden(golden_dir, "nefnet2_*.npz"):
        z = np.load(f)
        B, V, L, Q, seed = (int(z[k]) for k in ("B", "V", "L", "Q", "seed"))
        name = os.path.basename(f)
        b = batch_t(B, V, L, seed, Q)
        m = hashed_model2(V).eval()
        random.seed(seed)
        outs = m(b["data"], b["input_theta"], b["target_theta"], b["rois"], rest_theta=b["rest_theta"], phase="test")
        for got, key in zip(outs, ("out", "shuf_p", "shuf_l", "rest_out")):
            assert rel(got, z[key]) < FWD_TOL, (name, key, rel(got, z[key]))
        z1m, z2m = m(b["data"], b["input_theta"], b["target_theta"], b["rois"], phase="gen")
        assert z1m.shape == (B, 128, L // 4) and z2m.shape == (B, 128, L // 4)
        assert rel(sub(z1m), z["z1m_sub"]) < FWD_TOL and rel(stats(z2m), z["z2m_stats"]) < 1e-5
        # train phase
        cfg = make_cfg(V, str(z["reg"]))
        m = hashed_model2(V).train()
        m.dropout_p = 0.0
        m.keep_saved = True
        random.seed(seed)
        touts = m(b["data"], b["input_theta"], b["target_theta"], b["rois"], phase="train")
        losses = build_loss(cfg)(touts[0], touts[1], touts[2], b["target_view"].unsqueeze(1), cfg)
        losses[0].backward()
        for got, key in zip(touts, ("t_out", "t_shuf_p", "t_shuf_l")):
            assert rel(got, z[key]) < FWD_TOL, (name, key, rel(got, z[key]))
        assert maxabs(torch.stack([l_.detach() for l_ in losses]), z["losses"]) < 1e-6, name
        bc = {k: v.cpu() for k, v in b.items()}
        _, _, dec, _ = oracle_replaying(m, touts, bc, V, seed, reg=str(z["reg"]), model2=True, fold=(B, V), dt=torch.float64)
        tie_free = "tie_free" in z.files and bool(int(z["tie_free"]))      # see test_train_golden
        nsub = int(z["nsub"]) if "nsub" in z.files else 256
        if tie_free:
            assert dec.total_flips() == 0, (name, "a tie-free fixture saw replayed decisions", dec.total_flips())
        if True:      # the reference's own gradients for every fixture (bars + the measured distance replaying-fp64-oracle <-> fixture)
            n_fixture_grad += 1
            sq, sq64, got_all, ref_all, x64_all = 0.0, 0.0, [], [], []
            SLACK_CAP_FLAT, SLACK_CAP_TENSOR = (0.0, 0.0) if tie_free else slack_caps(dec)
            for k, p in m.named_parameters():
                if k in orc.DEAD_PARAMS:
                    assert p.grad is None, k
                    continue
                ref_sub = z["gsub:" + k]
                if k.endswith("double_conv.0.bias") or k.endswith("double_conv.3.bias"):
                    assert maxabs(sub(p.grad, nsub), ref_sub) < 1e-6, (name, k)
                    continue
                x64 = sub(dec.oracle_params[k].grad, nsub)
                slack = min(rel(x64, ref_sub), SLACK_CAP_TENSOR)      # capped measured allowance, see test_train_golden
                assert rel(sub(p.grad, nsub), ref_sub) < TENSOR_TOL + slack, (
                    name, k, rel(sub(p.grad, nsub), ref_sub), f"bar {TENSOR_TOL} + measured oracle-to-fixture distance {slack:.2e}")
                w_k = (p.grad.numel() / len(ref_sub)) ** 0.5        # the fixture holds <= 256 entries per tensor:
                got_all.append(w_k * sub(p.grad, nsub))               # weight them back to the tensor's size, so the
                ref_all.append(w_k * ref_sub)                        # statistic estimates the FLAT gradient's rel-L2
                x64_all.append(w_k * x64)
                sq += float((p.grad.double() ** 2).sum())
                sq64 += float((dec.oracle_params[k].grad.double() ** 2).sum())
            ref_norm = float(z["flat_grad_norm"])       # same rule for the norm: bar + the replaying oracle's own distance
            nslack = min(abs(sq64 ** 0.5 - ref_norm), SLACK_CAP_FLAT * ref_norm)
            assert abs(sq ** 0.5 - ref_norm) < FLAT_TOL * ref_norm + nslack, (
                name, f"bar {FLAT_TOL} + measured oracle-to-fixture distance {nslack / ref_norm:.2e} (capped at {SLACK_CAP_FLAT})")
            ref_cat = np.concatenate(ref_all)
            fslack = min(rel(np.concatenate(x64_all), ref_cat), SLACK_CAP_FLAT)
            assert rel(np.concatenate(got_all), ref_cat) < FLAT_TOL + fslack, (
                name, f"bar {FLAT_TOL} + measured oracle-to-fixture distance {fslack:.2e} (capped at {SLACK_CAP_FLAT})")
            if tie_free:
                TIE_FREE_FACTS.append((name, rel(np.concatenate(got_all), ref_cat)))
            if dec.total_flips() > 0 or fslack > 3e-4:
                import conftest
                conftest.report(f"{name}: {dec.total_flips()} replayed tie(s); replaying fp64 oracle to fixture {fslack:.2e}, "
                                f"HIP to fixture {rel(np.concatenate(got_all), ref_cat):.2e} on the flat statistic")
        sd = m.state_dict()
        for k in sd:
            if "running" in k:
                assert rel(sd[k], z["buf:" + k]) < 1e-5, (name, k)
    assert n_fixture_grad == len(golden(golden_dir, "nefnet2_*.npz")) >= 1
    _report_tie_free()


def test_nefnet2_sgd_step_runs_through_solver_api():
    """Model_nefnet2 with FusedSGD: shared-weight gradients reach the optimiser, parameters move, dead ones do not."""
    from electrocardio_panorama_amd.network import build_loss
    from electrocardio_panorama_amd.solver.optim_scheduler import get_optimizer
    from oracle import nefnet_oracle as orc
    cfg = make_cfg(3)
    m = hashed_model2(3).train()
    optim = get_optimizer(cfg, m.parameters())
    b = batch_t(4, 3, 512, 3)
    before = {k: v.detach().clone() for k, v in m.named_parameters()}
    random.seed(1)
    outs = m(b["data"], b["input_theta"], b["target_theta"], b["rois"], phase="train")
    build_loss(cfg)(outs[0], outs[1], outs[2], b["target_view"].unsqueeze(1), cfg)[0].backward()
    optim.step()
    optim.zero_grad()
    for k, v in m.named_parameters():
        if k.endswith("double_conv.0.bias") or k.endswith("double_conv.3.bias"):
            continue                                   # gradient analytically zero: may or may not move
        moved = not torch.equal(v.detach(), before[k])
        assert moved == (k not in orc.DEAD_PARAMS), k
