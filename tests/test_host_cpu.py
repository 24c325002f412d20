"""CPU: C-ABI surface, config loader, module surface, synthetic batches, data-parallel plumbing (gloo, world 2)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported():
    """The shared library loads and exports every function include/nefnet_hip.h declares (no compute calls)."""
    from electrocardio_panorama_amd import _lib
    from electrocardio_panorama_amd.csrc import build
    build.build(verbose=False)
    hdr = open(os.path.join(ROOT, "include", "nefnet_hip.h")).read()
    declared = set(re.findall(r"\b(nef_[a-z0-9_]+)\s*\(", hdr)) - {"nef_conv_args", "nef_stream_t"}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert _lib.load().nef_abi_version() == 18
    assert ctypes.sizeof(_lib.ConvArgs) == 384 == _lib.load().nef_conv_args_bytes()


def test_rejects_bad_calls_without_touching_the_gpu():
    from electrocardio_panorama_amd import _lib
    L = _lib.load()
    assert L.nef_stem_fwd(None, None, None, 1, 1, 512, None) == -2            # NEF_E_NULL
    assert L.nef_conv_bwd_weight_ws_bytes(2, 100, 3, 100, 128, 3) == 0        # unsupported channel count
    assert L.nef_conv_bwd_weight_ws_bytes(2, 100, 3, 128, 128, 7) > 0
    a = _lib.ConvArgs()
    assert L.nef_conv_fwd(ctypes.byref(a), None) == -2


def test_config_surface():
    from electrocardio_panorama_amd.config import get_defaults, resolve_config_path
    cfg = get_defaults()
    cfg.merge_from_file(resolve_config_path("config/nef-net.yml"))        # hyphen spelling of README.md:32
    assert cfg.SOLVER.lr == 0.1 and isinstance(cfg.SOLVER.lr, float)
    assert cfg.SOLVER.loss_factor == [0.5, 0.5, 1] and cfg.SOLVER.lr_step == [50, 100]
    assert cfg.DATA.lead_num == 3 and cfg.MODEL.model == "model_nefnet" and cfg.SOLVER.reg_loss == "l1_loss"
    with pytest.raises(KeyError):
        cfg.merge_from_list(["SOLVER.not_a_key", 1])
    with pytest.raises(ValueError):
        cfg.merge_from_list(["SOLVER.lr", "'text'"])
    cfg.merge_from_list(["SOLVER.lr", "1e-2", "DATA.lead_num", 8])
    assert cfg.SOLVER.lr == 0.01 and cfg.DATA.lead_num == 8


def test_module_surface_on_cpu():
    from electrocardio_panorama_amd.config import get_defaults
    from electrocardio_panorama_amd.network import build_loss, build_model, losswrapper
    from oracle import nefnet_oracle as orc
    cfg = get_defaults()
    cfg.MODEL.model = "model_nefnet"
    cfg.DATA.lead_num = 2
    m = build_model(cfg)
    exp = {**orc.param_shapes(2), **orc.buffer_shapes()}
    sd = m.state_dict()
    assert list(sd.keys()) == list(dict(m.named_parameters()).keys() | sd.keys()) or set(sd) == set(exp)
    assert all(tuple(sd[k].shape) == tuple(exp[k]) for k in exp)
    assert build_loss(cfg) is losswrapper
    cfg.MODEL.model = "other"
    with pytest.raises(ValueError):
        build_model(cfg)
    cfg.MODEL.loss = "other"
    with pytest.raises(ValueError):
        build_loss(cfg)
    # no CPU path: the product refuses to compute without a HIP device
    x = torch.zeros(1, 2, 512)
    with pytest.raises(RuntimeError):
        m(x, torch.zeros(1, 2, 2), torch.zeros(1, 2), torch.zeros(1, 7, 2, dtype=torch.int64))
    # reference-style init statistics (resnet_1d.py:114-120)
    w = sd["W_encoder.layer1.0.conv1.weight"]
    assert abs(float(w.std()) - (2.0 / (49 * 256)) ** 0.5) < 5e-4


def test_synthetic_batch_schema():
    from electrocardio_panorama_amd import synth
    b = synth.make_batch(5, 3, 5000, seed=1, Q=4)
    assert b["data"].shape == (5, 3, 5000) and b["data"].dtype == np.float32
    assert b["rois"].shape == (5, 7, 2) and b["rois"].dtype == np.int64
    r = b["rois"]
    assert (r[:, 0, 0] == 0).all() and (r[:, 6, 1] == 5000).all() and (r[:, 1:, 0] == r[:, :-1, 1]).all()
    assert (np.diff(r.reshape(5, -1), axis=1) >= 0).all()
    assert b["data"].min() >= 0 and b["data"].max() <= 1
    for i in range(5):
        assert (b["data"][i, :, r[i, 6, 0]:] == 0).all()
    assert b["rest_theta"].shape == (5, 4, 2) and b["rest_view"].shape == (5, 4, 5000)
    b2 = synth.make_batch(5, 3, 5000, seed=1, Q=4)
    assert all(np.array_equal(b[k], b2[k]) for k in b)


def test_data_parallel_plumbing_gloo_world2(tmp_path):
    """Two CPU ranks over gloo: batch sharding covers the global batch exactly once, and the flat-gradient
    all-reduce + 1/world averaging that FusedSGD performs matches a single-process run on the full batch."""
    script = os.path.join(ROOT, "tests", "dp_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29671", PYTHONPATH=ROOT)
    procs = [subprocess.Popen([sys.executable, script, str(r), "2", str(tmp_path)], env=env) for r in range(2)]
    assert all(p.wait(timeout=300) == 0 for p in procs)
    a, b = (np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(2))
    assert np.array_equal(a["flat"], b["flat"])
    assert np.allclose(a["flat"], a["expect"], rtol=1e-6, atol=1e-7)
    assert sorted(np.concatenate([a["idx"], b["idx"]]).tolist()) == list(range(8))
    # parallel.sum_counts (round 6): a counter only rank 1 incremented is seen by BOTH ranks, so both take the same decision
    assert a["counts"].tolist() == b["counts"].tolist() == [3, 1]


def test_dry_collective_model_and_single_rank_counts():
    """The modelled all-reduce duration of bench.py --dry-collective (ring over xGMI, per-link bound: 2 (N-1)/N S / 153 GB/s + 20 us,
    SURVEY section 5) and parallel.sum_counts without a process group."""
    from electrocardio_panorama_amd import parallel
    d = parallel.DryCollective(8)
    assert abs(d.ms(75.3e6) - (2 * 7 / 8 * 75.3e6 / 153e9 * 1e3 + 0.02)) < 1e-9 and 0.85 < d.ms(75.3e6) < 0.9
    assert abs(parallel.DryCollective(2).ms(153e6) - (1.0 + 0.02)) < 1e-9
    assert parallel.sum_counts([2, 0], "cpu") == [2, 0]


def test_metrics_psnr_ssim():
    from electrocardio_panorama_amd.utils.metric import PSNR, SSIM, ssim_1d
    rng = np.random.default_rng(0)
    gt = rng.random((2, 3, 64))
    rois = np.zeros((2, 7, 2), np.int64)
    rois[:, -1, 0] = [50, 64]
    assert PSNR(gt, gt, rois) == 100.0
    pred = gt + 0.1
    assert abs(PSNR(pred, gt, rois) - 20.0) < 1e-9                       # rmse 0.1 -> 20 dB
    assert abs(SSIM(gt, gt, rois) - 1.0) < 1e-12
    x, y = rng.random(40), rng.random(40)
    # direct evaluation of the SSIM definition at one interior position (window 7, sample covariance)
    i = 20
    wx, wy = x[i - 3:i + 4], y[i - 3:i + 4]
    c1, c2 = 1e-4, 9e-4
    cov = np.cov(wx, wy, ddof=1)
    s_i = ((2 * wx.mean() * wy.mean() + c1) * (2 * cov[0, 1] + c2)) / ((wx.mean() ** 2 + wy.mean() ** 2 + c1) * (cov[0, 0] + cov[1, 1] + c2))
    from scipy.ndimage import uniform_filter
    full = ssim_1d(np.r_[x], np.r_[y])
    assert 0 < full < 1
    # the interior value our implementation averages over equals the definition
    ux, uy = uniform_filter(x, 7), uniform_filter(y, 7)
    vx = 7 / 6 * (uniform_filter(x * x, 7) - ux * ux)
    assert abs(vx[i] - cov[0, 0]) < 1e-12 and abs(ux[i] - wx.mean()) < 1e-12
    assert abs(ssim_1d(x, x) - 1.0) < 1e-12 and s_i < 1


def test_dataset_reproduces_reference_batch(tmp_path):
    """f3: the Tianchi per-beat dataset on the two recordings bundled with the reference (stored as data fixtures)
    yields, for the same seeds, exactly the batch the reference's own dataset class produced (fixture G7)."""
    import random
    from electrocardio_panorama_amd.config import get_defaults, resolve_config_path
    from electrocardio_panorama_amd.dataset import build_dataset
    gold = os.path.join(ROOT, "tests", "golden")
    listing = tmp_path / "test_jsons.txt"
    listing.write_text("11315.json\n40723.json\n")
    cfg = get_defaults()
    cfg.merge_from_file(resolve_config_path("config/nef_net.yml"))
    cfg.DATA.test_label_path = cfg.DATA.train_label_path = str(listing)
    cfg.DATA.train_data_root = cfg.DATA.train_label_root = os.path.join(gold, "tianchi")
    random.seed(3)
    np.random.seed(3)
    ds = build_dataset(cfg, "test")
    items = [ds[i] for i in range(len(ds))]
    z = np.load(os.path.join(gold, "real_tianchi_B2_V3.npz"))
    for k in ("data", "rois", "input_theta", "target_view", "target_theta", "rest_view", "rest_theta"):
        got = np.stack([np.asarray(it[k]) for it in items])
        assert got.shape == z[k].shape, k
        assert np.array_equal(got.astype(z[k].dtype), z[k]), k
    assert items[0]["rois"].dtype == np.int64 and items[0]["rois"][6, 1] == 512 and items[0]["noise"].shape == (512,)


def test_ptb_dataset_reproduces_reference_items(tmp_path):
    """f3: PTBV2 / HeartBeatList on a PTB-format tree.  No PTB recording ships with the reference, so the tree is
    synthetic (oracle/make_golden.py::case_ptb wrote it into the fixture together with every `meta` the REFERENCE's
    PTBV2 returned for five lead plans); same seeds -> identical items, bit for bit, and the cache round-trips."""
    import json
    import random
    from electrocardio_panorama_amd.config import get_defaults, resolve_config_path
    from electrocardio_panorama_amd.dataset import HeartBeat, build_dataset
    from electrocardio_panorama_amd.dataset.ptbv2 import HeartBeatList
    z = np.load(os.path.join(ROOT, "tests", "golden", "ptb_synth.npz"))
    for k in z.files:
        if k.startswith("rec/") and k.endswith("/signal"):
            _, patient, name, _ = k.split("/")
            os.makedirs(tmp_path / patient, exist_ok=True)
            np.save(tmp_path / patient / (name + ".npy"), z[k])
            (tmp_path / patient / (name + ".json")).write_text(bytes(z[k.replace("/signal", "/label")]).decode())
    (tmp_path / "patients.txt").write_text("pA\npB\n")
    plans = ((3, "IIv2v5_v4I_372", "input_fix", "train"), (3, "IIv2v5_v4I_372", "random", "test"),
             (8, "_8120", "input_fix", "train"), (1, "_192", "input_fix", "test"), (5, "_561", "input_fix", "train"))
    for ci, (V, mode, dmode, phase) in enumerate(plans):
        cfg = get_defaults()
        cfg.merge_from_file(resolve_config_path("config/nef_net.yml"))
        cfg.DATA.dataset, cfg.DATA.lead_num, cfg.DATA.super_mode, cfg.DATA.train_data_mode = "ptbv2", V, mode, dmode
        cfg.MODEL.jitter_factor = 2.5
        paths = dict(train_pkl_path=str(tmp_path / f"cache{ci}.pkl"), test_pkl_path=str(tmp_path / f"cache{ci}.pkl"),
                     train_label_path=str(tmp_path / "patients.txt"), test_label_path=str(tmp_path / "patients.txt"),
                     train_data_root=str(tmp_path))
        for attempt in range(2):                      # second pass reads the .npz beat cache written by the first
            random.seed(40 + ci)
            np.random.seed(40 + ci)
            ds = build_dataset(cfg, phase, ptb_paths=paths)
            assert len(ds) == int(z[f"cfg{ci}/n"])
            items = [ds[i] for i in range(len(ds))]
            for k in ("data", "rois", "input_theta", "target_view", "target_theta", "ori_data", "rest_view", "rest_theta",
                      "noise"):
                got, exp = np.stack([np.asarray(it[k]) for it in items]), z[f"cfg{ci}/{k}"]
                assert got.shape == exp.shape and got.dtype == exp.dtype, (ci, k, got.dtype, exp.dtype)
                assert np.array_equal(got, exp), (ci, k, attempt)
            assert items[0]["unsupervision_lead_name"] == z[f"cfg{ci}/unsup"].tolist()
        assert os.path.exists(tmp_path / f"cache{ci}.npz")
    # a pickle written by the reference (objects of ITS HeartBeat class) is readable
    import pickle
    import sys
    import types
    mod = types.ModuleType("dataset.ptbv2")

    class _RefHeartBeat:
        def __init__(self, data, rois_list):
            self.data, self.rois_list = data, rois_list
    _RefHeartBeat.__name__ = _RefHeartBeat.__qualname__ = "HeartBeat"
    _RefHeartBeat.__module__ = "dataset.ptbv2"
    mod.HeartBeat = _RefHeartBeat
    sys.modules["dataset"], sys.modules["dataset.ptbv2"] = types.ModuleType("dataset"), mod
    try:
        blob = pickle.dumps([_RefHeartBeat(np.ones((12, 5)), np.arange(14).reshape(7, 2))], pickle.HIGHEST_PROTOCOL)
    finally:
        del sys.modules["dataset"], sys.modules["dataset.ptbv2"]
    (tmp_path / "ref.pkl").write_bytes(blob)
    hbs = HeartBeatList("unused", "unused", str(tmp_path / "ref.pkl")).heart_beats
    assert isinstance(hbs[0], HeartBeat) and hbs[0].data.shape == (12, 5) and hbs[0].rois_list[6, 1] == 13


def test_checkpointer_load_order(tmp_path):
    """Reference checkpointer.py:40-60: an explicit path wins over `last_checkpoint`; best_valid only without a path."""
    import torch
    from electrocardio_panorama_amd.utils import CheckPointer
    m = torch.nn.Linear(2, 2)
    ck = CheckPointer(m, save_dir=str(tmp_path))
    assert ck.load() == {} and ck.load(best_valid=True) == {}
    for e in (0, 1):
        with torch.no_grad():
            m.weight.fill_(float(e))
        ck.save(f"epoch_{e}", epoch=e)
    with torch.no_grad():
        m.weight.fill_(7.0)
    ck.save("best_valid", epoch=0, best_test_psnr_gen=3.0)
    ck.save("epoch_2", epoch=2)
    assert ck.load(os.path.join(tmp_path, "epoch_0.pkl"))["epoch"] == 0 and float(m.weight[0, 0]) == 0.0
    assert ck.load("")["epoch"] == 2 and ck.load(None)["epoch"] == 2                     # the pointer
    assert ck.load(best_valid=True)["best_test_psnr_gen"] == 3.0 and float(m.weight[0, 0]) == 7.0
    assert ck.load(os.path.join(tmp_path, "epoch_1.pkl"), best_valid=True)["epoch"] == 1  # explicit path still wins
    with pytest.raises(FileNotFoundError):
        ck.load(os.path.join(tmp_path, "epoch_9.pkl"))
    sd = torch.load(os.path.join(tmp_path, "epoch_1.pkl"))
    torch.save({"model": {"module." + k: v for k, v in sd["model"].items()}, "epoch": 5}, os.path.join(tmp_path, "dp.pkl"))
    assert ck.load(os.path.join(tmp_path, "dp.pkl"))["epoch"] == 5 and float(m.weight[0, 0]) == 1.0
    # a Nef-Net checkpoint also carries the operand magnitudes of its split-fp16 call sites (empty here: no device); the extra key is
    # consumed by load() and never returned as an `extra`
    from electrocardio_panorama_amd.network.model_nefnet import Model_nefnet
    net = Model_nefnet(lead_num=1)
    ck2 = CheckPointer(net, save_dir=str(tmp_path / "nef"))
    ck2.save("epoch_0", epoch=0)
    blob = torch.load(os.path.join(tmp_path, "nef", "epoch_0.pkl"))
    assert blob["h2_state"]["version"] == 1 and blob["h2_state"]["keys"] == []
    scope = net._nef_scope
    assert ck2.load() == {"epoch": 0} and net._nef_scope != scope


def test_sharded_loader_and_missing_label_lists(tmp_path):
    from electrocardio_panorama_amd import parallel, synth
    from electrocardio_panorama_amd.config import get_defaults
    batches = [synth.make_batch(8, 2, 64, seed=s) for s in range(3)]
    seen = [list(parallel.ShardedLoader(batches, r, 4)) for r in range(4)]
    assert all(len(s_) == 3 for s_ in seen) and len(parallel.ShardedLoader(batches, 0, 4)) == 3
    for i, full in enumerate(batches):
        for k in full:
            assert np.array_equal(np.concatenate([seen[r][i][k] for r in range(4)]), full[k])
    assert list(parallel.ShardedLoader(batches, 0, 1))[1] is batches[1]
    with pytest.raises(ValueError):
        parallel.shard_batch(batches[0], 0, 3)
    # a mistyped label list is an error (as in the reference), not a silent switch to synthetic data
    from electrocardio_panorama_amd import train_net
    cfg = get_defaults()
    cfg.DATA.train_label_path = str(tmp_path / "nope.txt")
    with pytest.raises(FileNotFoundError):
        train_net.build_loaders(cfg)
    cfg.DATA.synthetic = True
    cfg.DATA.lead_num = 3
    tr, te = train_net.build_loaders(cfg, batch_size=4)
    b = next(iter(tr))
    assert b["data"].shape == (4, 3, 512) and b["rest_view"].shape[1] == 9 and next(iter(te))["data"].shape[0] == 4


def test_oracle_decision_replay_and_dp_step():
    """The test instruments of the oracle: replaying the oracle's OWN decisions changes nothing and reports no flips; a
    forced difference is reported with its distance from the switching point; dp_train_step with one replica equals
    train_step, with two replicas it averages per-shard gradients (per-shard BatchNorm)."""
    import random
    import torch
    from electrocardio_panorama_amd import parallel, synth
    from oracle import hashweights as hw
    from oracle import nefnet_oracle as orc
    V, B, L = 2, 2, 256
    b = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.make_batch(B, V, L, seed=9).items()}

    def run(dec):
        P = orc.require_grad(hw.hashed_params(V))
        o = orc.forward(P, hw.hashed_buffers(), b["data"], b["input_theta"], b["target_theta"], b["rois"], phase="train",
                        training=True, p=0.0, lead_choice=(0, 1), dec=dec)
        orc.loss_v1(o[0], o[1], o[2], b["target_view"].unsqueeze(1), dec=dec)[0].backward()
        return o, torch.cat([v.grad.reshape(-1) for k, v in P.items() if v.grad is not None])

    class Spy(orc.Decisions):                       # records the oracle's own decisions
        def relu(self, site, pre):
            self.masks_seen = getattr(self, "masks_seen", {})
            self.masks_seen[site] = pre.detach() > 0
            return super().relu(site, pre)

    spy = Spy()
    o0, g0 = run(spy)
    own = dict(spy.masks_seen)
    own.update(loss1=torch.sign(o0[0].detach() - o0[1].detach()), loss2=torch.sign(o0[0].detach() - o0[2].detach()),
               loss3=torch.sign(o0[0].detach() - b["target_view"].unsqueeze(1)))
    d1 = orc.Decisions(own)
    _, g1 = run(d1)
    assert d1.total_flips() == 0 and len(d1.report) == len(own) and float((g1 - g0).abs().max()) < 1e-7 * float(g0.abs().max())
    site = "pass0.decoder.3.double_conv.4"
    forced = dict(own)
    forced[site] = own[site].clone()
    forced[site][0, 0, 0] = ~forced[site][0, 0, 0]
    d2 = orc.Decisions(forced)
    run(d2)
    assert d2.total_flips() == 1 and d2.report[site]["flips"] == 1 and d2.report[site]["worst"] > 0

    def fresh():
        return orc.require_grad(hw.hashed_params(V)), orc.SGDState(0.1)

    P1, o1 = fresh()
    random.seed(1)
    v1 = orc.train_step(P1, hw.hashed_buffers(), o1, b, p=0.0, lead_choice=(1, 0))
    P2, o2 = fresh()
    vals, _ = orc.dp_train_step(P2, [hw.hashed_buffers()], o2, [b], p=0.0, lead_choice=(1, 0))
    assert vals[0] == v1 and all(torch.equal(P1[k], P2[k]) for k in P1)
    P3, o3 = fresh()
    shards = [{k: v[r:r + 1] for k, v in b.items()} for r in range(2)]
    Bfs = [hw.hashed_buffers(), hw.hashed_buffers()]
    vals3, avg = orc.dp_train_step(P3, Bfs, o3, shards, p=0.0, lead_choice=(1, 0))
    assert len(vals3) == 2 and not torch.equal(Bfs[0]["decoder.1.double_conv.1.running_mean"],
                                               Bfs[1]["decoder.1.double_conv.1.running_mean"])
    assert not all(torch.equal(P1[k], P3[k]) for k in P1)        # per-shard BatchNorm != full-batch BatchNorm


def test_dropin_shim_resolves_reference_import_lines():
    """With electrocardio_panorama_amd/dropin first on sys.path, the import statements of the reference's entry scripts
    (codes/main.py:1-9, train_net.py:1-8, val_net.py:1-6, solver/solver.py:10-13) resolve to this build's objects."""
    code = (
        "from train_net import main\n"
        "from config import cfg\n"
        "from dataset import build_dataset\n"
        "from dataset import *\n"
        "from solver import Solver\n"
        "from utils import seed_torch\n"
        "from network import build_model, build_loss\n"
        "from solver.optim_scheduler import get_optimizer, get_lr_scheduler\n"
        "from utils.mertic import SSIM, PSNR\n"
        "from utils import CheckPointer\n"
        "import val_net, electrocardio_panorama_amd.train_net as tn, electrocardio_panorama_amd.solver.solver as ss\n"
        "import electrocardio_panorama_amd.utils.metric as mm, electrocardio_panorama_amd.config as cc\n"
        "assert main is tn.main and Solver is ss.Solver and PSNR is mm.PSNR and cfg is cc.cfg\n"
        "assert callable(val_net.main) and callable(build_dataset) and cfg.SOLVER.optim == 'sgd'\n"
        "print('DROPIN_OK')\n")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "electrocardio_panorama_amd", "dropin"), ROOT]))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd="/tmp")
    assert r.returncode == 0 and "DROPIN_OK" in r.stdout, r.stderr[-2000:]


def test_env_switch_groups(monkeypatch, capsys):
    """Product switches are always honoured; diagnostics switches only under NEF_DIAG=1 (ignored with one warning otherwise);
    anything else is a programming error.  The C side gates its own diagnostics reads the same way (nef_common.h: nef_diag_env)."""
    from electrocardio_panorama_amd import _env
    monkeypatch.delenv("NEF_DIAG", raising=False)
    monkeypatch.setenv("NEF_H2", "0")
    assert _env.get("NEF_H2", "1") == "0"
    monkeypatch.setenv("NEF_FUSE_STATS", "0")
    _env._warned.discard("NEF_FUSE_STATS")
    assert _env.get("NEF_FUSE_STATS", "1") == "1" and _env.get("NEF_FUSE_STATS", "1") == "1"
    assert capsys.readouterr().err.count("NEF_FUSE_STATS=0 ignored") == 1
    monkeypatch.setenv("NEF_DIAG", "1")
    assert _env.get("NEF_FUSE_STATS", "1") == "0"
    monkeypatch.setenv("NEF_NOT_A_SWITCH", "1")
    with pytest.raises(KeyError):
        _env.get("NEF_NOT_A_SWITCH")
    # every switch the sources read is registered in one of the groups
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seen = set()
    for dirpath, _, files in os.walk(os.path.join(root, "electrocardio_panorama_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                seen |= set(re.findall(r'(?:_env\.get|nef_diag_env|getenv)\(["\'](NEF_[A-Z0-9_]+)', txt))
    hooks = {"NEF_SHARE_GPU", "NEF_DIST_BACKEND", "NEF_DIST_FORCE"}
    assert seen - set(_env.PRODUCT) - _env.DIAGNOSTICS - hooks == set(), seen - set(_env.PRODUCT) - _env.DIAGNOSTICS - hooks


def test_polyphase_form_of_conv_behind_upsampling_is_exact_algebra():
    """The identities csrc/elementwise.hip (poly_weights_kernel, poly_fwd_edge_kernel, poly_bwd_edge_kernel, poly_wgrad_fold_kernel)
    and DESIGN 3.0b rest on, in fp64 on the CPU against torch's own ops (codes/network/model_nefnet.py:102-105): forward = two K = 3
    convs of the half-resolution input with the phase weights (clamped ends) minus the two row-end columns; backward-data = conv of
    the phase-major gradient with the transposed / flipped phase weights plus its row-end terms; weight gradient = phase weight
    gradients folded back minus theirs."""
    import torch
    import torch.nn.functional as F
    torch.manual_seed(0)
    B, Ci, Co, Th = 3, 5, 4, 9
    x = torch.randn(B, Ci, Th, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Co, Ci, 3, dtype=torch.float64, requires_grad=True)
    gy = torch.randn(B, Co, 2 * Th, dtype=torch.float64)
    y = F.conv1d(F.interpolate(x, scale_factor=2, mode="linear", align_corners=False), w, padding=1)
    y.backward(gy)
    w0, w1, w2 = w.detach().unbind(2)
    W = [torch.stack([.75 * w0 + .25 * w1, .25 * w0 + .75 * w1 + .75 * w2, .25 * w2], 2),
         torch.stack([.25 * w0, .75 * w0 + .75 * w1 + .25 * w2, .25 * w1 + .75 * w2], 2)]
    xd = x.detach()
    xe = torch.cat([xd[:, :, :1], xd, xd[:, :, -1:]], 2)                       # nn.Upsample's clamped sources
    yp = torch.stack([F.conv1d(xe, W[p]) for p in (0, 1)], 3).reshape(B, Co, 2 * Th)
    yp[:, :, 0] -= torch.einsum("oc,bc->bo", w0, xd[:, :, 0])
    yp[:, :, -1] -= torch.einsum("oc,bc->bo", w2, xd[:, :, -1])
    assert torch.allclose(yp, y.detach(), atol=1e-12)
    # backward-data on the phase-major gradient [B, 2 Co, Th] (row 2 co + p), zero padding
    gpm = gy.view(B, Co, Th, 2).permute(0, 1, 3, 2).reshape(B, 2 * Co, Th)
    Wsyn = torch.stack(W, 1).reshape(2 * Co, Ci, 3)                             # row 2 co + p
    gx = F.conv1d(gpm, Wsyn.permute(1, 0, 2).flip(2), padding=1)
    gx[:, :, 0] += .25 * (torch.einsum("oc,bo->bc", w1 - w0, gy[:, :, 0]) + torch.einsum("oc,bo->bc", w0, gy[:, :, 1]))
    gx[:, :, -1] += .25 * (torch.einsum("oc,bo->bc", w1 - w2, gy[:, :, -1]) + torch.einsum("oc,bo->bc", w2, gy[:, :, -2]))
    assert torch.allclose(gx, x.grad, atol=1e-12)
    # weight gradient: phase weight gradients over the clamped-end input, folded back, minus the row-end terms
    gW = torch.stack([torch.einsum("bot,bct->oc", gpm, xe[:, :, j:j + Th]) for j in range(3)], 2).view(Co, 2, Ci, 3)
    A, Bq = gW[:, 0], gW[:, 1]
    gw = torch.stack([.75 * (A[..., 0] + Bq[..., 1]) + .25 * (A[..., 1] + Bq[..., 0]),
                      .75 * (A[..., 1] + Bq[..., 1]) + .25 * (A[..., 0] + Bq[..., 2]),
                      .75 * (A[..., 1] + Bq[..., 2]) + .25 * (A[..., 2] + Bq[..., 1])], 2)
    gw[:, :, 0] -= torch.einsum("bo,bc->oc", gy[:, :, 0], xd[:, :, 0])
    gw[:, :, 2] -= torch.einsum("bo,bc->oc", gy[:, :, -1], xd[:, :, -1])
    assert torch.allclose(gw, w.grad, atol=1e-12)
