"""Data parallelism with world_size 2 on the REAL train step (BASELINE configs[2] semantics).

The test box has one GPU, so two ranks share cuda:0 and exchange gradients over gloo (test hooks NEF_SHARE_GPU /
NEF_DIST_BACKEND of parallel.init_from_env, honoured only under NEF_TEST_HOOKS=1); everything else -- sharding, per-shard BatchNorm, the flat-gradient
all-reduce folded into the fused SGD kernel, rank-0 buffers, bench.py's multi-rank branch -- is the production path.
The oracle side restates the reference's nn.DataParallel semantics (oracle.dp_train_step)."""
import json
import os
import random
import subprocess
import sys

import numpy as np
import pytest
import torch

from util import rel

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(port):
    return dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", NEF_DIST_BACKEND="gloo",
                NEF_SHARE_GPU="1", NEF_TEST_HOOKS="1", PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")


@pytest.fixture(scope="module")
def dp_run(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("dp2"))
    script = os.path.join(ROOT, "tests", "dp2_worker.py")
    procs = [subprocess.Popen([sys.executable, script, out], env=dict(_env(29681), RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    logs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(lg[-3000:] for lg in logs)
    return out


def test_world2_step_8_leads_vs_per_shard_bn_oracle(dp_run):
    from oracle import hashweights as hw
    from oracle import nefnet_oracle as orc
    from electrocardio_panorama_amd import parallel, synth
    a, b = (np.load(os.path.join(dp_run, f"step_rank{r}.npz")) for r in range(2))
    # (i) both ranks hold bit-identical parameters after the step; the shards cover the batch exactly once
    assert np.array_equal(a["params"], b["params"]) and np.array_equal(a["avg_grad"], b["avg_grad"])
    assert sorted(np.concatenate([a["idx"], b["idx"]]).tolist()) == [0, 1, 2, 3]
    # (ii) against the oracle: two shard gradients with PER-SHARD BatchNorm statistics, averaged, one SGD step
    V, B, L, seed = 8, 4, 1000, 21
    full = synth.make_batch(B, V, L, seed=seed)
    masks = hw.hashed_masks(V, B, L // 4)
    shards, mranks = [], []
    for r in range(2):
        sh = parallel.shard_batch(full, r, 2)
        idx = parallel.shard_indices(B, r, 2)
        shards.append({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sh.items()})
        mranks.append({k: v[idx[0]:idx[-1] + 1] for k, v in masks.items()})
    P = orc.require_grad(hw.hashed_params(V))
    P0 = {k: v.detach().clone() for k, v in P.items()}
    Bfs = [hw.hashed_buffers(), hw.hashed_buffers()]
    random.seed(seed)
    vals, avg = orc.dp_train_step(P, Bfs, orc.SGDState(0.1), shards, masks_ranks=mranks, loss_factor=(0.5, 0.5, 1.0))
    for r, z in enumerate((a, b)):
        assert np.abs(z["losses"] - np.array(vals[r])).max() < 2e-6, (r, z["losses"], vals[r])
    names = [str(n) for n in a["names"]]
    assert set(names) == {k for k in P if k not in orc.DEAD_PARAMS}
    # each rank checked its own shard gradient against the oracle (decision replay, flat <= 1e-4 / tensor <= 1e-3, inside
    # the worker); the all-reduced, 1/world-scaled buffer must be the average of the two oracle shard gradients
    assert float(a["shard_flat"]) < 1e-4 and float(b["shard_flat"]) < 1e-4
    want = 0.5 * (a["oracle_grad"].astype(np.float64) + b["oracle_grad"].astype(np.float64))
    assert rel(a["avg_grad"], want) < 1e-4, rel(a["avg_grad"], want)
    if int(a["flips"]) == 0 and int(b["flips"]) == 0:       # no ReLU / L1 tie on either shard: the plain oracle, too
        plain = torch.cat([avg[n].reshape(-1) for n in names]).numpy()
        assert rel(a["avg_grad"], plain) < 2e-4, rel(a["avg_grad"], plain)
    # post-step parameters: compare the UPDATE (p_new - p_old) = -lr * averaged gradient (first step: buf = g)
    from test_model_gpu import make_cfg
    from electrocardio_panorama_amd.network import build_model
    order = [n for n, _ in build_model(make_cfg(V)).named_parameters()]
    old = torch.cat([P0[n].reshape(-1) for n in order]).numpy()
    upd = {n: np.zeros(P0[n].numel(), np.float64) for n in order}
    off = 0
    for n in names:
        k = P0[n].numel()
        upd[n] = -0.1 * want[off:off + k]
        off += k
    assert rel(a["params"].astype(np.float64) - old, np.concatenate([upd[n] for n in order])) < 2e-4
    # (iii) running statistics are per shard until rank 0's are broadcast
    keys = [k[7:] for k in a.files if k.startswith("before:")]
    assert keys and any(not np.array_equal(a["before:" + k], b["before:" + k]) for k in keys)
    for k in keys:
        assert np.array_equal(a["after:" + k], a["before:" + k])            # rank 0 is authoritative
        assert np.array_equal(b["after:" + k], a["before:" + k])
        assert rel(a["before:" + k], Bfs[0][k].numpy()) < 1e-5 and rel(b["before:" + k], Bfs[1][k].numpy()) < 1e-5


def test_world2_solver_epoch_sharded_loader_vs_oracle(dp_run):
    """The packaged driver's path -- Solver.run_one_epoch over parallel.ShardedLoader with FusedSGD -- on two ranks: one
    iteration checked like the 8-lead step above (each rank's shard against the decision-replaying oracle inside the
    worker; here the all-reduced buffer and the parameter update against the average of the two oracle shard gradients),
    then two more iterations with momentum: parameters stay bit-identical across ranks, losses are those of the shards."""
    from oracle import hashweights as hw
    from electrocardio_panorama_amd.network import build_model
    from test_model_gpu import make_cfg
    a, b = (np.load(os.path.join(dp_run, f"solver_rank{r}.npz")) for r in range(2))
    assert np.array_equal(a["params_1"], b["params_1"]) and np.array_equal(a["params_3"], b["params_3"])
    assert np.array_equal(a["avg_grad"], b["avg_grad"]) and not np.array_equal(a["params_1"], a["params_3"])
    for z in (a, b):
        assert np.abs(z["losses"][0] - z["oracle_losses"]).max() < 2e-6
        assert z["losses"].shape == (3, 4) and np.isfinite(z["losses"]).all()
    assert not np.array_equal(a["losses"][0], b["losses"][0])                 # different shards, different losses
    want = 0.5 * (a["oracle_grad"].astype(np.float64) + b["oracle_grad"].astype(np.float64))
    assert rel(a["avg_grad"], want) < 1e-4, rel(a["avg_grad"], want)
    V = 3
    P0 = hw.hashed_params(V)
    order = [n for n, _ in build_model(make_cfg(V)).named_parameters()]
    live = [str(n) for n in a["live"]]
    upd = {n: np.zeros(P0[n].numel(), np.float64) for n in order}
    off = 0
    for n in live:
        k = P0[n].numel()
        upd[n] = -0.1 * want[off:off + k]
        off += k
    old = torch.cat([P0[n].reshape(-1) for n in order]).numpy()
    assert rel(a["params_1"].astype(np.float64) - old, np.concatenate([upd[n] for n in order])) < 2e-4


def test_world2_rccl_on_two_devices(tmp_path):
    """The same worker over backend 'nccl' (= RCCL) with one rank per device -- the production transport -- whenever the box
    has two GPUs (the single-GPU test box skips): both ranks must end with bit-identical parameters, and the averaged
    gradient must equal the gloo run's contract (mean of the two oracle shard gradients)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two HIP devices (RCCL between two ranks)")
    out = str(tmp_path)
    env = _env(29685)
    env.pop("NEF_DIST_BACKEND")
    env.pop("NEF_SHARE_GPU")
    script = os.path.join(ROOT, "tests", "dp2_worker.py")
    procs = [subprocess.Popen([sys.executable, script, out], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    logs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(lg[-3000:] for lg in logs)
    a, b = (np.load(os.path.join(out, f"step_rank{r}.npz")) for r in range(2))
    assert np.array_equal(a["params"], b["params"]) and np.array_equal(a["avg_grad"], b["avg_grad"])
    want = 0.5 * (a["oracle_grad"].astype(np.float64) + b["oracle_grad"].astype(np.float64))
    assert rel(a["avg_grad"], want) < 1e-4, rel(a["avg_grad"], want)
    sa, sb = (np.load(os.path.join(out, f"solver_rank{r}.npz")) for r in range(2))
    assert np.array_equal(sa["params_3"], sb["params_3"])


def test_early_bucket_is_used_and_matches_single_bucket(dp_run):
    """FusedSGD with the early gradient bucket (everything behind the per-lead encoder is all-reduced while the encoder
    blocks are still being back-propagated) must give bit-for-bit the buffer a single all-reduce gives: the sums are the
    same pairs of numbers.  The worker records both."""
    a, b = (np.load(os.path.join(dp_run, f"step_rank{r}.npz")) for r in range(2))
    assert int(a["early_params"]) > 0 and int(a["early_params"]) == int(b["early_params"])
    assert np.array_equal(a["avg_grad"], a["avg_grad_single"]) and np.array_equal(b["avg_grad"], b["avg_grad_single"])


def test_early_bucket_is_dropped_after_two_backward_passes(dp_run):
    """Gradient accumulation (two backward passes, one step): the early bucket is a snapshot of ONE pass, so the optimiser must
    fall back to the single all-reduce of p.grad -- bit for bit the manual all-reduce of the accumulated gradients -- and leave
    no pending collective behind."""
    for r in range(2):
        z = np.load(os.path.join(dp_run, f"accum_rank{r}.npz"))
        assert bool(z["ok"]) and bool(z["pending"])


def test_world2_graphed_solver_is_bit_identical_to_eager(dp_run):
    """Solver.run_one_epoch with cfg.SOLVER.graph (hipGraph replay + one flat all-reduce) vs the eager two-bucket path on the
    same three sharded iterations: bit-identical parameters on both ranks, losses equal."""
    for r in range(2):
        e, g = np.load(os.path.join(dp_run, f"solver_rank{r}.npz")), np.load(os.path.join(dp_run, f"graph_rank{r}.npz"))
        assert np.array_equal(e["params_3"], g["params_3"])
        assert np.abs(e["losses"] - g["losses"]).max() < 1e-6
    a, b = (np.load(os.path.join(dp_run, f"graph_rank{r}.npz")) for r in range(2))
    assert np.array_equal(a["params_3"], b["params_3"])


def test_world2_clamp_on_one_rank_skips_the_step_on_both(dp_run):
    """A split-fp16 launch that clamps on ONE rank taints the step on EVERY rank (the taint word is all-reduced in front of the
    gradients): nobody applies the update, parameters stay bit-identical, the next step runs normally -- eager and graphed."""
    a, b = (np.load(os.path.join(dp_run, f"taint_rank{r}.npz")) for r in range(2))
    for z in (a, b):
        for label in ("eager", "graph"):
            assert bool(z[label + "_unchanged"]) and bool(z[label + "_warned"]) and bool(z[label + "_resumed"]), label
        assert int(z["eager_skipped"]) == 1 and int(z["graph_skipped"]) == 2
    assert np.array_equal(a["eager_params"], b["eager_params"]) and np.array_equal(a["graph_params"], b["graph_params"])


def test_bench_two_ranks_prints_one_json_line():
    """bench.py's torchrun branch (barrier, max-over-ranks timing, whole-job value) with world_size 2."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29683", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--batch", "8", "--len", "1000", "--leads", "8"]
    env = _env(29683)
    env.pop("WORLD_SIZE")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["global_batch"] == 16
    assert line["value"] > 0 and abs(line["value"] - 16 * 2 / (line["ms_per_step"] * 2e-3)) < 1e-3 * line["value"]
    assert line["cpu_baseline"] is None and np.isfinite(line["final_loss"])
    assert line["allreduce_ms_exposed"] is not None and line["allreduce_ms_exposed"] >= 0


def test_bench_self_launches_two_ranks():
    """Plain `python bench.py --gpus 2` (no torchrun environment -- how the round driver calls it): bench.py re-executes
    itself under torch.distributed.run and rank 0 still prints exactly one JSON line for the whole job."""
    env = _env(0)
    for k in ("WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8",
           "--len", "1000", "--leads", "3"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 16 and line["config"]["parallelism"] == "dp2"
    assert line["value"] > 0 and line["allreduce_ms_exposed"] is not None and np.isfinite(line["final_loss"])


def test_bench_self_launches_eight_ranks():
    """`python bench.py --gpus 8` end to end on a one-GPU box: eight ranks share cuda:0 and sum their gradients over gloo (the
    NEF_SHARE_GPU / NEF_DIST_BACKEND test hooks) -- the launcher, the rendezvous, the 8-way shard arithmetic, the two-graph split
    capture with the early bucket between the replays, max-over-ranks timing and the single JSON line are the production path;
    only the transport differs from RCCL over xGMI.  Tiny shape (batch 2 per rank, L = 512)."""
    env = _env(0)
    for k in ("WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--batch", "2",
           "--len", "512", "--leads", "3", "--no-secondary"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["config"]["global_batch"] == 16 and line["config"]["parallelism"] == "dp8"
    assert line["scaling"] == "weak" and line["hip_graph"] is True
    assert line["value"] > 0 and line["allreduce_ms_exposed"] is not None and np.isfinite(line["final_loss"])
