"""One data-parallel rank of tests/test_dp_gpu.py.  Two of these share cuda:0 (NEF_SHARE_GPU=1) and talk over gloo
(NEF_DIST_BACKEND=gloo), so the REAL train step -- Model_nefnet forward, losswrapper, backward, FusedSGD with its
flat-gradient all-reduce -- runs with world_size 2 on a one-GPU box.  Launched with RANK / WORLD_SIZE / MASTER_* set.

part 1 (BASELINE configs[2] shape: 8 leads): one step on this rank's shard of a (4, 8, L) batch, dropout masks replayed.
part 2: two iterations of Solver.run_one_epoch over parallel.ShardedLoader (the packaged driver's sharding)."""
import os
import random
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from electrocardio_panorama_amd import parallel, synth                       # noqa: E402
from electrocardio_panorama_amd.network import build_loss, build_model       # noqa: E402
from electrocardio_panorama_amd.solver import Solver                         # noqa: E402
from electrocardio_panorama_amd.solver.optim_scheduler import get_optimizer  # noqa: E402
from oracle import hashweights as hw                                         # noqa: E402
from test_model_gpu import make_cfg, oracle_replaying                        # noqa: E402

out_dir = sys.argv[1]
rank, world, local = parallel.init_from_env()
assert world == 2 and dist.is_initialized() and local == (0 if os.environ.get("NEF_SHARE_GPU") == "1" else rank)
dev = torch.device("cuda", local)


def flat_params(model):
    return torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu().numpy()


def buffers(model):
    return {k: v.detach().cpu().numpy().copy() for k, v in model.named_buffers() if "running" in k}


# ---------------------------------------------------------------- part 1: one 8-lead step
V, B, L, seed = 8, 4, 1000, 21
cfg = make_cfg(V)
model = build_model(cfg).float()
model.load_state_dict({**hw.hashed_params(V), **hw.hashed_buffers()})
model.to(dev).train()
model.keep_saved = True
full = synth.make_batch(B, V, L, seed=seed)
shard = parallel.shard_batch(full, rank, world)
idx = parallel.shard_indices(B, rank, world)
masks = hw.hashed_masks(V, B, L // 4)
shard_masks = {k: v[idx[0]:idx[-1] + 1].contiguous() for k, v in masks.items()}
model.dropout_masks = {k: v.to(dev) for k, v in shard_masks.items()}
b = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in shard.items()}
optim = get_optimizer(cfg, model.parameters())
random.seed(seed)
outs = model(b["data"], b["input_theta"], b["target_theta"], b["rois"], phase="train")
losses = build_loss(cfg)(outs[0], outs[1], outs[2], b["target_view"].unsqueeze(1), cfg)
losses[0].backward()
names = [n for n, p in model.named_parameters() if p.grad is not None]
# this rank's shard gradient against the oracle on the same shard (tie-free bars, decisions replayed); the oracle's
# shard gradient is handed to the parent, which checks the all-reduced average against the average of the two
bc = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in shard.items()}
_, _, dec, shard_flat = oracle_replaying(model, outs, bc, V, seed, masks=shard_masks)
oracle_grad = torch.cat([dec.oracle_params[n].grad.reshape(-1) for n in names]).numpy()
model.last_saved = None
# reference result: ONE all-reduce of the whole flat gradient (what FusedSGD did before the early bucket existed)
single = torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.grad is not None])
dist.all_reduce(single)
avg_grad_single = (single / world).cpu().numpy()
pend = parallel._EARLY["pending"]
early_params = 0 if pend is None else len(pend["names"])
optim.step()
avg_grad = (optim._flat[0]["g"] / world).cpu().numpy()          # the all-reduced sum / world, in `names` order
optim.zero_grad()
buf_before = buffers(model)
parallel.broadcast_buffers(model)
buf_after = buffers(model)
np.savez(os.path.join(out_dir, f"step_rank{rank}.npz"), params=flat_params(model), avg_grad=avg_grad,
         names=np.array(names), losses=np.array([float(v) for v in losses]), idx=np.array(idx), oracle_grad=oracle_grad,
         flips=np.array(dec.total_flips()), shard_flat=np.array(shard_flat), avg_grad_single=avg_grad_single,
         early_params=np.array(early_params),
         **{"before:" + k: v for k, v in buf_before.items()}, **{"after:" + k: v for k, v in buf_after.items()})
# gradient accumulation: TWO backward passes before one optimiser step.  The early bucket (a snapshot of the first pass) must be
# dropped and the step must reduce what is in p.grad -- the sum of both passes
lossf = build_loss(cfg)
for _ in range(2):
    random.seed(seed)
    o2 = model(b["data"], b["input_theta"], b["target_theta"], b["rois"], phase="train")
    lossf(o2[0], o2[1], o2[2], b["target_view"].unsqueeze(1), cfg)[0].backward()
expect = torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.grad is not None])
dist.all_reduce(expect)
optim.step()
np.savez(os.path.join(out_dir, f"accum_rank{rank}.npz"), ok=np.array(bool(torch.equal(optim._flat[0]["g"], expect))),
         pending=np.array(parallel._EARLY["pending"] is None and parallel._EARLY["backwards"] == 0))
optim.zero_grad()
del model, optim, outs, losses, o2
dist.barrier()

# ---------------------------------------------------------------- part 2: Solver.run_one_epoch over sharded loaders
V, B, L, seed = 3, 4, 512, 5
cfg = make_cfg(V)
sol = Solver(cfg, use_tensorboardx=False)
assert sol.device == dev
sol.model.load_state_dict({**hw.hashed_params(V), **hw.hashed_buffers()})
sol.model.dropout_p = 0.0
sol.model.keep_saved = True
full = synth.make_batch(B, V, L, seed=seed, Q=2)
optim = get_optimizer(cfg, sol.model.parameters())
random.seed(seed)
res = sol.run_one_epoch(parallel.ShardedLoader([full]), "train", optim)      # ONE iteration of the packaged driver
assert len(res[2]) == B // world and res[2][0].shape == (L,)                  # predicted views of this rank's shard
names = [n for n, p in sol.model.named_parameters() if n not in ("w_feature_extractor.0.weight", "w_feature_extractor.0.bias")
         and not n.startswith("w_conv.0.residual_conv") and not n.startswith("z2_conv2.0.residual_conv")]
avg_grad = (optim._flat[0]["g"] / world).cpu().numpy()
# this rank's shard against the decision-replaying oracle: gradient = what this rank contributed to the all-reduce.  The
# step has been applied already, so the shard gradient is not in p.grad any more; the oracle's shard gradients are handed
# to the parent, which checks their average against the all-reduced buffer (and with it the parameter update).
sv = sol.model.last_saved
out3 = sv["dec"][2]
Bs = B // world
outs = (out3[0:Bs], out3[Bs:2 * Bs], out3[2 * Bs:3 * Bs])
shard = parallel.shard_batch(full, rank, world)
bc = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in shard.items()}
from decisions import assert_flips_are_ties, gpu_decisions                   # noqa: E402
from oracle import nefnet_oracle as orc                                      # noqa: E402
dec = orc.Decisions(gpu_decisions(sol.model, outs, bc["target_view"].unsqueeze(1)))
P = orc.require_grad(hw.hashed_params(V))
random.seed(seed)
ref = orc.forward(P, hw.hashed_buffers(), bc["data"], bc["input_theta"], bc["target_theta"], bc["rois"], phase="train",
                  training=True, p=0.0, dec=dec)
rl = orc.loss_v1(ref[0], ref[1], ref[2], bc["target_view"].unsqueeze(1), dec=dec)
rl[0].backward()
assert_flips_are_ties(dec)
live = [n for n, p in sol.model.named_parameters() if P[n].grad is not None]
oracle_grad = torch.cat([P[n].grad.reshape(-1) for n in live]).numpy()
params_1 = flat_params(sol.model)
# two more iterations (momentum carried) for the trajectory-level checks
more = [synth.make_batch(B, V, L, seed=seed + 1 + s, Q=2) for s in range(2)]
sol.model.keep_saved = False
sol.model.last_saved = None
res2 = sol.run_one_epoch(parallel.ShardedLoader(more), "train", optim)
np.savez(os.path.join(out_dir, f"solver_rank{rank}.npz"), params_1=params_1, params_3=flat_params(sol.model),
         losses=np.array(res[0] + res2[0]), oracle_losses=np.array([float(v) for v in rl]), avg_grad=avg_grad,
         oracle_grad=oracle_grad, live=np.array(live), flips=np.array(dec.total_flips()))
dist.barrier()

# ---------------------------------------------------------------- part 3: the same three iterations through the GRAPHED Solver
# (cfg.SOLVER.graph: forward + loss + backward captured once per shape, the flat gradient as ONE all-reduce behind the replay,
# FusedSGD's own flat buffers stepped): bit-identical parameters to the eager, two-bucket run above
cfg_g = make_cfg(V)
cfg_g.SOLVER["graph"] = True
sol_g = Solver(cfg_g, use_tensorboardx=False)
sol_g.model.load_state_dict({**hw.hashed_params(V), **hw.hashed_buffers()})
sol_g.model.dropout_p = 0.0
optim_g = get_optimizer(cfg_g, sol_g.model.parameters())
random.seed(seed)
res_g = sol_g.run_one_epoch(parallel.ShardedLoader([full]), "train", optim_g, collect_views=False)
res_g2 = sol_g.run_one_epoch(parallel.ShardedLoader(more), "train", optim_g, collect_views=False)
assert sol_g._graph_stepper is not None and sol_g._graph_stepper.calls == 3
np.savez(os.path.join(out_dir, f"graph_rank{rank}.npz"), params_3=flat_params(sol_g.model),
         losses=np.array(res_g[0] + res_g2[0]))
dist.barrier()

# ---------------------------------------------------------------- part 4: a split-fp16 launch clamps on ONE rank
# (simulated: rank 1 bumps its device clamp counter -- exactly what conv_h2_kernel's x_clamped does).  The taint word travels in
# front of the flat gradient buffer through the same all-reduce, so BOTH ranks skip the update: parameters and momentum stay as
# they were, bit-identical across ranks, and every rank's h2_skipped() says 1 -- eager step, then a replayed (graphed) step
from electrocardio_panorama_amd import ops                                   # noqa: E402
ops.h2_clamped(), ops.h2_skipped()
taint = {}
for label, solver, opt in (("eager", sol, optim), ("graph", sol_g, optim_g)):
    before = flat_params(solver.model)
    mom_before = opt._flat[0]["buf"].clone()
    if rank == 1:
        ops._amax_state(dev)["clamped"] += 3
    try:
        solver.run_one_epoch(parallel.ShardedLoader([full]), "train", opt, collect_views=False)
        warned = True                       # FusedSGD skipped the step: Solver only warns
    except RuntimeError:
        warned = False
    taint[label + "_unchanged"] = bool(np.array_equal(before, flat_params(solver.model)) and torch.equal(mom_before, opt._flat[0]["buf"]))
    taint[label + "_warned"] = warned
    taint[label + "_skipped"] = int(ops._amax_state(dev)["skipped"].item())
    # the step after it is a normal step again
    solver.run_one_epoch(parallel.ShardedLoader([full]), "train", opt, collect_views=False)
    taint[label + "_resumed"] = bool(not np.array_equal(before, flat_params(solver.model)))
    taint[label + "_params"] = flat_params(solver.model)
np.savez(os.path.join(out_dir, f"taint_rank{rank}.npz"), **{k: np.array(v) for k, v in taint.items()})
dist.barrier()
dist.destroy_process_group()
print("DP2_OK", rank)
