#!/usr/bin/env python
"""Nef-Net train-step benchmark (BASELINE.json metric: ECG-samples/sec of one train step).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus 8 --steps 10 --warmup 3        (re-executes itself under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One process per GPU; each rank trains on its own shard of the synthetic batch (weak scaling: per-GPU batch
fixed), gradients are summed with ONE RCCL all-reduce of the flat gradient buffer per step.  A step is
forward + losswrapper + backward + fused momentum-SGD on inputs already resident in HBM.  Rank 0 prints one
JSON line.  At N=1 the line also carries the CPU baseline (the torch-CPU oracle timed on this box's host cores on
a bounded sample of the same workload), the roofline of the dominant kernel (the k=7 grouped-conv MFMA kernel: EXECUTED
matrix-core flops over its HIP-event time on the launch stream, against the fp32 MFMA peak; `roofline.whole_step`: the
executed matrix work of the WHOLE step over ms_per_step against the same peak) and `secondary`: the reference's own training
shape (batch 32, 3 leads, len 512) eager and graph-replayed, and the two inference configs of BASELINE.json (configs[3] panorama
sweep, configs[4] per-GPU share of gen_ecg), all timed after the train step.
"""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np          # noqa: E402
import torch                # noqa: E402
import torch.distributed as dist  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
FP16_MFMA_PEAK_TFLOPS = 2500.0  # same guide: v_mfma_f32_32x32x16_f16, dense
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (BASELINE config 2: 256)")
    ap.add_argument("--leads", type=int, default=3)
    ap.add_argument("--len", type=int, default=5000, dest="length")
    ap.add_argument("--cpu-batch", type=int, default=32, help="CPU-baseline batch (SURVEY 8d: config-2 shape at B=32)")
    ap.add_argument("--cpu-steps", type=int, default=4)
    ap.add_argument("--cpu-threads", type=str, default="16,1",
                    help="thread pools to time the CPU baseline with ('all' = os.cpu_count(): 5 MINUTES PER STEP on the "
                         "256-core GPU box, profiles/r02_cpu_thread_sweep.md); the best one is headlined")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dropout", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="do not bracket launches with HIP events (roofline / hbm_bound are then null): at launch-bound "
                         "shapes the ~250 event pairs per step cost more host time than the launches themselves")
    ap.add_argument("--graph", action="store_true", help="(default) the timed steps replay the captured hipGraph of the step")
    ap.add_argument("--no-graph", action="store_true",
                    help="issue every launch of the timed steps from Python instead (what the per-kernel breakdown steps always do)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the configs[3] / configs[4] inference timings")
    ap.add_argument("--dry-collective", type=int, default=0, metavar="N",
                    help="ONE-GPU run of the N-rank data-parallel graphed step with the gradient all-reduces replaced by device-side "
                         "stand-ins of their modelled xGMI duration (parallel.DryCollective): reports allreduce_ms_exposed for the "
                         "two-graph and the one-graph schedule next to the collective-free step; prints its own JSON line")
    return ap.parse_args()


def make_cfg(V):
    from electrocardio_panorama_amd.config import get_defaults, resolve_config_path
    cfg = get_defaults()
    cfg.merge_from_file(resolve_config_path("config/nef_net.yml"))
    cfg.DATA.lead_num = V
    return cfg


def cpu_baseline(V, L, B, steps, pools):
    """The oracle (a torch-CPU port of the reference step) on this box's host cores, SURVEY 8d: config-2 shape at batch
    32, 1 warm-up + `steps` timed steps per thread pool, plus a 1-thread figure (batch 4, fewer steps -- a bounded
    sample).  SURVEY 8d asks for torch.set_num_threads(os.cpu_count()); on the 256-core GPU box that pool takes 307 s
    per step (0.10 samples/s, measured: profiles/r02_cpu_thread_sweep.md) -- torch-CPU oversubscribes -- so the default
    pools are the measured optimum (16 threads) and 1 thread; `--cpu-threads all,16,1` reproduces the full-pool
    figure.  `value` is the BEST pool's throughput, i.e. the strongest CPU baseline measured."""
    from electrocardio_panorama_amd import synth
    from oracle import nefnet_oracle as orc
    ncpu = os.cpu_count() or 1
    runs = []
    for pool in pools:
        n = ncpu if pool == "all" else max(1, min(int(pool), ncpu))
        if any(r["threads"] == n for r in runs):
            continue
        b_, k_ = (B, steps) if n > 1 else (min(B, 4), max(1, steps // 2))
        torch.set_num_threads(n)
        P = orc.require_grad(orc.reference_style_init(V, seed=123))
        Bf = orc.fresh_buffers()
        opt = orc.SGDState(0.1)
        batch = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.make_batch(b_, V, L, seed=123).items()}
        random.seed(123)
        orc.train_step(P, Bf, opt, batch)
        t0 = time.perf_counter()
        for _ in range(k_):
            orc.train_step(P, Bf, opt, batch)
        dt = time.perf_counter() - t0
        runs.append({"threads": n, "batch": b_, "steps": k_, "samples_per_s": round(b_ * k_ / dt, 3),
                     "ms_per_step": round(1e3 * dt / k_, 1)})
    best = max(runs, key=lambda r: r["samples_per_s"])
    one = [r for r in runs if r["threads"] == 1]
    return {"value": best["samples_per_s"], "unit": "ECG-samples/sec", "cores": best["threads"], "host_cores": ncpu,
            "kind": "port",
            "sample": f"{best['steps']} train steps of the torch-CPU oracle at batch {best['batch']}, V={V}, L={L} "
                      f"(1 warm-up step untimed), best of the thread pools listed in `pools`",
            "ms_per_step": best["ms_per_step"], "pools": runs,
            "one_thread_samples_per_s": one[0]["samples_per_s"] if one else None}


def secondary(dev, headline=None):
    """`headline` = (V, B, L, dropout) of the timed workload: its strict-fp32 leg (strict_fp32) runs first.  BASELINE configs[3] and configs[4] on one GPU, through the fp16 panorama decoder (Model_nefnet.panorama_dtype =
    'fp16'; encoder, ROI path and angular encoding stay fp32).  configs[3]: eval-mode sweep, 1 view in -> 360 queried
    angles, batch 1024, L=512 (reference model_nefnet.py:181-192); configs[4]: gen_ecg on one GPU's share of the global
    batch 4096 = 512 samples x 3 leads x 12 angles, L=5000 (model_nefnet.py:196-218).  Inputs resident in HBM, random-init
    weights, synthetic data.  Rates: views = (sample, angle) pairs; out_GBps = fp32 output bytes / time; hbm_frac /
    mfma_frac price the four wide decoder convs' ALGORITHMIC bytes (450 KB per view at L=512: one fp16 write + one fp16
    read of each conv output) and flops (113.5 MFLOP per view at L=512) against 8 TB/s and the dense fp16 MFMA peak."""
    from electrocardio_panorama_amd import synth
    from electrocardio_panorama_amd.network import build_model

    def rates(n_views, L, Q, dt, B):
        flops, byts = n_views * 113.5e6 * (L / 512), n_views * 450e3 * (L / 512)
        return {"ms": round(dt * 1e3, 3), "views_per_s": round(n_views / dt, 1), "samples_per_s": round(B / dt, 1),
                "out_GBps": round(4.0 * n_views * L / dt / 1e9, 2),
                "hbm_frac": round(byts / dt / 1e9 / HBM_PEAK_GBS, 4), "mfma_frac": round(flops / dt / 1e12 / FP16_MFMA_PEAK_TFLOPS, 4)}

    out = {}
    if headline is not None:
        try:
            out["strict_fp32"] = strict_fp32(dev, headline[0], headline[1], headline[2], dropout=headline[3])
        except Exception as exc:
            out["strict_fp32"] = {"error": f"{type(exc).__name__}: {exc}"}
    out["reference-native 32x3x512"] = native_shape(dev)
    # configs[3]
    B, V, L, Q = 1024, 1, 512, 360
    torch.manual_seed(123)
    m = build_model(make_cfg(V)).float().to(dev).eval()
    m.panorama_dtype = "fp16"
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in synth.make_batch(B, V, L, seed=123, Q=Q).items()
         if k in ("data", "input_theta", "target_theta", "rois", "rest_theta")}

    def sweep():
        random.seed(0)
        return m(t["data"], t["input_theta"], t["target_theta"], t["rois"], rest_theta=t["rest_theta"], phase="test")
    sweep()
    torch.cuda.synchronize(dev)
    n, t0 = 3, time.perf_counter()
    for _ in range(n):
        res = sweep()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / n
    out["configs[3]"] = dict(rates(B * Q, L, Q, dt, B), workload=f"panorama sweep, batch={B}, {V} lead in -> {Q} angles, "
                             f"len={L}, fp16 decoder, whole forward (encoder + ROI path + {Q}-angle decoder)", sweeps=n,
                             finite=bool(torch.isfinite(res[1] if isinstance(res, (tuple, list)) else res).all()))
    del m, t, res
    torch.cuda.empty_cache()
    # configs[4], one GPU's share
    B, V, L, Q = 512, 3, 5000, 12
    torch.manual_seed(123)
    m = build_model(make_cfg(V)).float().to(dev).eval()
    m.panorama_dtype = "fp16"
    meta = synth.make_batch(B, V, L, seed=123, Q=Q)
    rois = torch.from_numpy(np.ascontiguousarray(meta["rois"])).to(dev)
    theta = torch.from_numpy(np.ascontiguousarray(meta["rest_theta"])).to(dev)
    g = torch.Generator(device=dev).manual_seed(5)
    z1 = torch.randn(B, 128 * V, L // 4, device=dev, generator=g) * 0.1
    z2 = torch.randn(B, 128 * V, 7, 32, device=dev, generator=g) * 0.1
    res = m.gen_ecg(z1, z2, theta, rois)
    torch.cuda.synchronize(dev)
    n, t0 = 5, time.perf_counter()
    for _ in range(n):
        res = m.gen_ecg(z1, z2, theta, rois)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / n
    out["configs[4] share"] = dict(rates(B * Q, L, Q, dt, B), workload=f"gen_ecg from latents, one GPU's share of the global "
                                   f"batch 4096: {B} samples x {V} leads x {Q} angles, len={L}, fp16 decoder", calls=n,
                                   finite=bool(torch.isfinite(res).all()))
    del m, res, z1, z2
    torch.cuda.empty_cache()
    return out


def strict_fp32(dev, V, B, L, steps=5, warmup=2, dropout=True):
    """The headline workload once more with EVERY conv on the fp32 matrix-core kernels (what NEF_H2=0 selects: Winograd / direct
    forms of conv_mfma.hip, IEEE fp32 products and accumulation -- the reference's own arithmetic, model_nefnet.py:18-21), same
    process, after the headline: the strict-fp32 figure is driver-timed every round next to the fp32-class one."""
    from electrocardio_panorama_amd import ops, synth
    from electrocardio_panorama_amd.graph import GraphedTrainStep
    from electrocardio_panorama_amd.network import build_model
    from electrocardio_panorama_amd.solver.optim_scheduler import get_optimizer
    from electrocardio_panorama_amd.utils import seed_torch
    cfg = make_cfg(V)
    saved, ops.H2 = ops.H2, False
    try:
        seed_torch(cfg.seed)
        model = build_model(cfg).float().to(dev).train()
        if not dropout:
            model.dropout_p = 0.0
        optim = get_optimizer(cfg, model.parameters())
        meta = synth.make_batch(B, V, L, seed=123)
        data, rois, in_theta, tgt_view, tgt_theta = (torch.from_numpy(np.ascontiguousarray(meta[k])).to(dev) for k in
                                                     ("data", "rois", "input_theta", "target_view", "target_theta"))
        tgt_view = tgt_view.unsqueeze(1)
        g = GraphedTrainStep(model, cfg, optimizer=optim)
        for _ in range(warmup):
            loss = g(data, in_theta, tgt_theta, rois, tgt_view)[0]
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = g(data, in_theta, tgt_theta, rois, tgt_view)[0]
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / steps
        res = {"workload": f"{_config_name(V, B, L)} with every conv on the fp32 matrix-core kernels (NEF_H2=0), graph-replayed",
               "dtype": "f32 (IEEE fp32 products and accumulation, v_mfma_f32_32x32x2_f32; Winograd F(2,.)/F(4,.) forms where the "
                        "shape allows)", "steps": steps, "warmup": warmup, "ms_per_step": round(dt * 1e3, 3),
               "samples_per_s": round(B / dt, 1), "final_loss": float(loss)}
        del model, optim, g
    finally:
        ops.H2 = saved
    torch.cuda.empty_cache()
    return res


def native_shape(dev, steps=40, warmup=10):
    """The only shape the reference itself trains (codes/config/nef_net.yml:8-13, codes/train_net.py:27-28: batch 32, 3 leads,
    beats of 512 samples): the train step eager and replayed as one captured hipGraph (graph.GraphedTrainStep -- what
    Solver.run_one_epoch uses at this size).  Launch-bound: ~250 launches for ~0.8 ms of matrix work."""
    from electrocardio_panorama_amd import synth
    from electrocardio_panorama_amd.graph import GraphedTrainStep
    from electrocardio_panorama_amd.network import build_loss, build_model
    from electrocardio_panorama_amd.solver.optim_scheduler import get_optimizer
    from electrocardio_panorama_amd.utils import seed_torch
    B, V, L = 32, 3, 512
    cfg = make_cfg(V)
    res = {"workload": f"Nef-Net train step at the reference's own training shape: batch={B}, {V}-lead, len={L}, Standin losses "
                       f"on, dropout on", "steps": steps}
    meta = synth.make_batch(B, V, L, seed=123)
    data, rois, in_theta, tgt_view, tgt_theta = (torch.from_numpy(np.ascontiguousarray(meta[k])).to(dev) for k in
                                                 ("data", "rois", "input_theta", "target_view", "target_theta"))
    tgt_view = tgt_view.unsqueeze(1)
    for mode in ("eager", "graph"):
        seed_torch(cfg.seed)
        model = build_model(cfg).float().to(dev).train()
        lossf = build_loss(cfg)
        if mode == "graph":
            g = GraphedTrainStep(model, cfg)
            step = lambda: g(data, in_theta, tgt_theta, rois, tgt_view)[0]
        else:
            optim = get_optimizer(cfg, model.parameters())

            def step():
                o, sp, sl = model(data, in_theta, tgt_theta, rois, phase="train")
                ls = lossf(o, sp, sl, tgt_view, cfg)
                ls[0].backward()
                optim.step()
                optim.zero_grad()
                return ls[0]
        for _ in range(warmup):
            loss = step()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step()
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / steps
        res[mode] = {"ms_per_step": round(dt * 1e3, 3), "samples_per_s": round(B / dt, 1), "final_loss": float(loss)}
        del model
    torch.cuda.empty_cache()
    return res


def dry_collective(args, dev):
    """bench.py --dry-collective N [--leads 8]: what the overlap logic of the data-parallel graphed step does at the real step
    length, on one GPU.  Three schedules of the same shard, each captured and timed on its own: no collective at all; the
    two-graph schedule (graph A | early bucket's all-reduce under graph B | encoder bucket exposed); the one-graph schedule (one
    fully exposed all-reduce behind the replay).  The collectives are parallel.DryCollective stand-ins: a few workgroups kept busy
    on the communication stream for 2 (N-1)/N S / 153 GB/s + 20 us -- the MODELLED ring all-reduce over xGMI;
    no multi-GPU node was available to the build, so this is the overlap exercised, not a scaling measurement."""
    from electrocardio_panorama_amd import parallel, synth
    from electrocardio_panorama_amd.graph import GraphedTrainStep
    from electrocardio_panorama_amd.network import build_model
    from electrocardio_panorama_amd.solver.optim_scheduler import get_optimizer
    from electrocardio_panorama_amd.utils import seed_torch
    V, L, B, N = args.leads, args.length, args.batch, args.dry_collective
    cfg = make_cfg(V)
    meta = synth.make_batch(B, V, L, seed=123)
    data, rois, in_theta, tgt_view, tgt_theta = (torch.from_numpy(np.ascontiguousarray(meta[k])).to(dev) for k in
                                                 ("data", "rois", "input_theta", "target_view", "target_theta"))
    tgt_view = tgt_view.unsqueeze(1)
    res = {}
    for name in ("no_collective", "two_graph", "one_graph"):
        seed_torch(cfg.seed)
        model = build_model(cfg).float().to(dev).train()
        optim = get_optimizer(cfg, model.parameters())
        g = GraphedTrainStep(model, cfg, optimizer=optim)
        dry = None
        if name != "no_collective":
            dry = g.dry = parallel.DryCollective(N)
            g.dp, g.split_capture = True, name == "two_graph"
        for _ in range(args.warmup):
            g(data, in_theta, tgt_theta, rois, tgt_view)
        torch.cuda.synchronize(dev)
        parallel.TIMING = [] if dry is not None else None
        if dry is not None:
            dry.calls.clear()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = g(data, in_theta, tgt_theta, rois, tgt_view)[0]
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / args.steps
        ev, parallel.TIMING = parallel.TIMING or [], None
        res[name] = {"ms_per_step": round(dt * 1e3, 3), "samples_per_s_per_gpu": round(B / dt, 1), "final_loss": float(loss)}
        if dry is not None:
            per_step = dry.calls[:len(dry.calls) // max(args.steps, 1)]
            res[name].update(allreduce_ms_exposed=round(sum(a.elapsed_time(b) for a, b in ev) / max(args.steps, 1), 4),
                             collectives_per_step=[{"bytes": b_, "modelled_ms": round(m_, 4)} for b_, m_ in per_step])
        del model, optim, g
        torch.cuda.empty_cache()
    base = res["no_collective"]["ms_per_step"]
    for name in ("two_graph", "one_graph"):
        res[name]["step_overhead_ms_vs_no_collective"] = round(res[name]["ms_per_step"] - base, 3)
    print(json.dumps({
        "mode": "dry-collective (ONE GPU; modelled all-reduce durations, not a multi-GPU measurement)", "modelled_ranks": N,
        "model": f"ring all-reduce over xGMI, per-link bound: 2 (N-1)/N S / {parallel.DryCollective.LINK_GBPS} GB/s + "
                 f"{parallel.DryCollective.LATENCY_US} us per collective (SURVEY.md section 5)",
        "config": {"workload": f"{_config_name(V, B, L)}: Nef-Net train step, {V}-lead len={L}, batch={B}/GPU, graph-replayed"},
        "steps": args.steps, "warmup": args.warmup, "schedules": res,
        "exposed_basis": "HIP events on the launching stream around everything it waits for behind the last graph replay (the "
                         "encoder bucket's collective + what is left of the early bucket's)"}), flush=True)


def _config_name(V, B, L):
    """Which BASELINE.json config the per-GPU shape is."""
    return {(3, 256, 5000): "configs[1]", (8, 256, 5000): "configs[2] (per-GPU shard)", (1, 4, 2048): "configs[0]"}.get(
        (V, B, L), "custom shape")


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment (how the round driver may call it): re-execute under
    torch.distributed.run, one rank per GPU on 127.0.0.1 with a free port, and hand its exit code on.  Rank 0 of that job
    prints the JSON line.  The explicit `python -m torch.distributed.run ... bench.py --gpus N` form keeps working."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        raise SystemExit(self_launch(args))
    from electrocardio_panorama_amd import ops, parallel, synth
    from electrocardio_panorama_amd.network import build_loss, build_model
    from electrocardio_panorama_amd.solver.optim_scheduler import get_optimizer
    from electrocardio_panorama_amd.utils import seed_torch

    rank, world, local = parallel.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the two must agree (plain `python bench.py --gpus N` "
                         f"launches its own ranks)")
    if world > 1 and not parallel._hook("NEF_SHARE_GPU") and torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} but only {torch.cuda.device_count()} HIP device(s) are visible")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if args.dry_collective > 1:
        if world != 1:
            raise SystemExit("--dry-collective models the collectives of an N-rank job on ONE GPU: run it with --gpus 1")
        return dry_collective(args, dev)
    V, L, B = args.leads, args.length, args.batch
    cfg = make_cfg(V)
    seed_torch(cfg.seed)                       # identical init and Standin lead choices on every rank
    model = build_model(cfg).float().to(dev).train()
    if args.no_dropout:
        model.dropout_p = 0.0
    lossf = build_loss(cfg)
    optim = get_optimizer(cfg, model.parameters())
    meta = synth.make_batch(B, V, L, seed=123 + rank)
    data, rois, in_theta, tgt_view, tgt_theta = (torch.from_numpy(np.ascontiguousarray(meta[k])).to(dev) for k in
                                                 ("data", "rois", "input_theta", "target_view", "target_theta"))
    tgt_view = tgt_view.unsqueeze(1)

    # The timed steps replay the captured hipGraph of the step (graph.GraphedTrainStep over the SAME FusedSGD buffers -- what
    # Solver.run_one_epoch does): since the convs run on the fp16 matrix cores a step is ~290 launches of 36 ms in all, and
    # issuing them from Python leaves 2.5-4.5 ms of gaps per step on a busy host (measured: 37.2-39.2 ms eager, box and load
    # dependent, against 34.4-34.9 replayed).  Same kernels, same arithmetic; --no-graph times the eager issue.
    args.graph = not args.no_graph
    graphed = None
    if args.graph:
        from electrocardio_panorama_amd.graph import GraphedTrainStep
        graphed = GraphedTrainStep(model, cfg, optimizer=optim)

    def step(eager=False):
        if graphed is not None and not eager:
            return graphed(data, in_theta, tgt_theta, rois, tgt_view)[0]
        out, sp, sl = model(data, in_theta, tgt_theta, rois, phase="train")
        losses = lossf(out, sp, sl, tgt_view, cfg)
        losses[0].backward()
        optim.step()
        optim.zero_grad()
        return losses[0]

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    # Timed region: only the launches of the dominant kernel (the K=7 grouped conv, forward and backward-data) are
    # bracketed with HIP events -- 12 pairs per step.  Bracketing every launch (~250 pairs) costs 0.7 ms of GPU idle per
    # step (measured: 57.8 vs 57.1 ms), so the per-kernel breakdown and the HBM-bound set are taken from two extra,
    # untimed steps right after the timed region.
    T_lat = L // 4
    dom = {("conv_fwd", 7, V, 128, 128, B, T_lat), ("conv_bwd_data", 7, V, 128, 128, B, T_lat),
           ("conv_bwd_weight", 7, V, 128, 128, B, T_lat)}
    timing = rank == 0 and not args.no_kernel_events and not args.graph
    ops.PROFILE, ops.PROFILE_ONLY = ([] if timing else None), (lambda tag: tag in dom)
    parallel.TIMING = [] if (world > 1 and rank == 0) else None      # events around the exposed part of the gradient all-reduce
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    prof, ops.PROFILE, ops.PROFILE_ONLY = ops.PROFILE or [], None, None
    ar_events, parallel.TIMING = parallel.TIMING or [], None
    final_loss_t = loss.detach().clone()      # (the graphed step returns a view of its static loss buffer)
    prof_all, extra_steps = [], 2
    if not args.no_kernel_events:       # every rank takes the extra steps (they contain the all-reduce); always issued eagerly
        # Per-kernel breakdown: untimed extra steps with EVERY launch bracketed, on ONE stream (NEF_SIDE_STREAM=0 is read per
        # step by engine._side), so each kernel has the chip to itself and its event time is its own duration -- in the
        # default two-stream schedule a chain kernel that shares the chip with a side-stream weight-gradient kernel reads
        # up to 2x long.  The timed region above ran the real (two-stream) schedule.
        side_env = os.environ.get("NEF_SIDE_STREAM")
        os.environ["NEF_SIDE_STREAM"] = "0"
        ops.PROFILE = [] if rank == 0 else None
        # one more bracketed step first, whose events are dropped: the first event pair recorded behind a cross-stream
        # wait can come back with the wait inside it (seen once: 52 ms on a 0.23 ms launch of the first bracketed step)
        step(eager=True)
        fence()
        if rank == 0:
            ops.PROFILE = []
        for _ in range(extra_steps):
            step(eager=True)
        fence()
        prof_all, ops.PROFILE = ops.PROFILE or [], None
        if side_env is None:
            os.environ.pop("NEF_SIDE_STREAM")
        else:
            os.environ["NEF_SIDE_STREAM"] = side_env
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(final_loss_t.item())

    # the arithmetic of the step: fp32 data, parameters, accumulators and elementwise work; the conv products run on the fp16 matrix
    # cores on two-term fp16 splits of the fp32 operands (ops.H2; fp32-class: 22-23 bits per operand, block-scaled) -- or on the fp32
    # matrix cores throughout with NEF_H2=0
    DTYPE = ("f32 (conv products: fp32 operands split into fp16 hi+lo = 22-23 bits, block-scaled, 3 fp16 matrix products each, fp32 accumulate; fp32-class, not IEEE fp32)"
             if ops.H2 else "f32")
    if rank == 0:
        # Every tagged conv launch is priced against BOTH roofs and bound by the larger floor:
        #   matrix floor = executed fp32 flops / 157.3 TF + executed fp16 flops / 2500 TF  (ops.EXEC_FRAC / ops.EXEC_FP16: what the
        #                  matrix cores really do per algorithmic multiply -- 3 fp16 instructions-worth in the split-fp16 kernels,
        #                  9/14 .. 1/2 fp32 in the Winograd forms, 1 in the direct fp32 kernels),
        #   HBM floor    = algorithmic bytes (input + output rows once, + residual / gate operands where the block has them) / 8 TB/s.
        # `roofline` carries the kernel with the largest share of the step (by its launches in the single-stream breakdown
        # steps), timed over the launches of the TIMED region; `k7_kernels_one_at_a_time` the three K = 7 kernels;
        # `whole_step` the sum of the binding floors of all tagged conv launches + the algorithmic bytes of the HBM-bound
        # passes over the timed step.
        T = L // 4
        KNAME = {3: "conv_h2_kernel (direct conv on two-term fp16 splits of both fp32 operands, fp32 accumulate)",
                 2: "conv_wino4_kernel (Winograd F(4,3) / F(4,4)+F(4,3), fp32)", 1: "conv_wino_kernel (Winograd F(2,3) / F(2,4)+F(2,3), fp32)",
                 0: "conv_fwd_kernel (direct, fp32)"}

        def price(tag, ms):
            role, k_, g_, cig_, cog_, b_, t_ = tag[:7]
            extra = tag[7] if len(tag) > 7 else ""
            f_ = 2.0 * b_ * g_ * cog_ * t_ * cig_ * k_
            ex32, ex16 = f_ * ops.EXEC_FRAC.get(tag, 1.0), f_ * ops.EXEC_FP16.get(tag, 0.0)
            # external operands at the resolution they are actually read at: behind the x2-upsampling prologue the input
            # is the half-resolution tensor (round 4 priced it at full resolution); the BatchNorm-backward epilogue also reads the
            # layer's BatchNorm input (half resolution where the upsampling sits in between)
            t_in = t_ / 2 if extra.startswith("up") else t_
            byts = 4.0 * b_ * g_ * (cig_ * t_in + cog_ * t_) + 4.0 * g_ * cig_ * cog_ * k_
            if "bnb" in extra:
                byts += 4.0 * b_ * g_ * cog_ * (t_ / 2 if extra.endswith("bnbup") else t_)
            if role != "conv_bwd_weight" and cig_ == cog_ and k_ == 7:
                byts += 0.5 * 4.0 * b_ * g_ * cog_ * t_        # every second launch of an encoder block reads a residual / gate row
            fl_m = (ex32 / FP32_MFMA_PEAK_TFLOPS + ex16 / FP16_MFMA_PEAK_TFLOPS) / 1e9      # ms
            fl_h = byts / HBM_PEAK_GBS / 1e6                                              # ms
            bound = "mfma" if fl_m >= fl_h else "hbm"
            d = {"avg_ms": round(ms, 4), "bound": bound, "floor_ms": round(max(fl_m, fl_h), 4),
                 "frac": round(max(fl_m, fl_h) / ms, 4), "mfma_floor_ms": round(fl_m, 4), "hbm_floor_ms": round(fl_h, 4),
                 "algorithmic_flops_per_launch": f_, "executed_fp32_mfma_flops": ex32, "executed_fp16_mfma_flops": ex16,
                 "algorithmic_bytes_per_launch": byts}
            if bound == "mfma":
                peak = FP16_MFMA_PEAK_TFLOPS if ex16 > 0 else FP32_MFMA_PEAK_TFLOPS
                d.update(achieved=round((ex16 if ex16 > 0 else ex32) / ms / 1e9, 2), peak=peak, unit="TFLOP/s",
                         matrix_dtype="f16 (two-term splits of f32 operands, 22-23 bits), f32 accumulate" if ex16 > 0 else "f32")
            else:
                d.update(achieved=round(byts / ms / 1e6, 1), peak=HBM_PEAK_GBS, unit="GB/s")
            return d

        serial = {}
        for tag, s_, e_ in prof_all:
            if tag[0] in ("conv_fwd", "conv_bwd_data", "conv_bwd_weight"):
                serial.setdefault(tag, []).append(s_.elapsed_time(e_))
        timed = {}
        for tag, s_, e_ in prof:
            timed.setdefault(tag, []).append(s_.elapsed_time(e_))
        roof = None
        if serial:
            dom_tag = max(serial, key=lambda t_: sum(serial[t_]))
            # weight gradients run on the side stream, next to chain kernels: their event time in the timed region contains that
            # sharing, so they are priced by their launches of the single-stream breakdown steps
            use = (timed.get(dom_tag) if dom_tag[0] != "conv_bwd_weight" else None) or serial[dom_tag]
            roof = price(dom_tag, sum(use) / len(use))
            traffic = traffic_source = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath) and (V, B, L) == (3, 256, 5000):     # PMC pass was taken at configs[1] only
                tj = json.load(open(tpath))
                # kernel names of the trace: conv_h2w2_kernel<K, PRO, ...> (producer / consumer weight gradient; conv_h2w_kernel = its
                # first form) and conv_h2_kernel<K, PRO, TM>
                pre = (("conv_h2w2_kernel<%d, 0" % dom_tag[1], "conv_h2w_kernel<%d, 0" % dom_tag[1]) if dom_tag[0] == "conv_bwd_weight"
                       else ("conv_h2_kernel<%d, 0, 2" % dom_tag[1],))
                hits = [v for pfx in pre for k_, v in tj.get("by_kernel", {}).items() if k_.startswith(pfx)]
                traffic = hits[0] if (hits and ops.H2) else None
                # NOT measured by this run: replayed from the committed rocprofv3 --pmc pass (separate run, as the
                # counters cannot be collected together with timing)
                traffic_source = ("replayed from " + tj.get("source", "profiles/traffic.json")) if traffic else None
                # counter bytes / algorithmic bytes: the dominant kernel, and the most-launched K = 3 form (conv_h2_kernel<3, 0, 2>:
                # its launches differ in shape, so the ratio uses the launch-weighted mean of their algorithmic bytes)
                if traffic:
                    roof["traffic_ratio"] = round(traffic / roof["algorithmic_bytes_per_launch"], 3)
                k3 = [v for k_, v in tj.get("by_kernel", {}).items() if k_.startswith("conv_h2_kernel<3, 0, 2")]
                # launches of that kernel form: K = 3, 128-row tile, no input prologue (plain, phase-major gradient, BatchNorm-backward
                # epilogue, and the affine-less polyphase backward over the phase-stacked input is <3, 4, 2>: not this form)
                tags3 = [(tg, len(v)) for tg, v in serial.items() if tg[0] in ("conv_fwd", "conv_bwd_data") and tg[1] == 3 and
                         tg[4] % 128 == 0 and tg[6] >= 128 and (tg[7] if len(tg) > 7 else "") in ("", "pm", "bnb", "pmbnb")]
                if k3 and tags3 and ops.H2:
                    alg3 = sum(price(tg, 1.0)["algorithmic_bytes_per_launch"] * n_ for tg, n_ in tags3) / sum(n_ for _, n_ in tags3)
                    roof["traffic_ratio_k3"] = {"kernel": "conv_h2_kernel<3, 0, 2>", "counter_bytes_per_launch": k3[0],
                                                "algorithmic_bytes_per_launch_mean": round(alg3), "ratio": round(k3[0] / alg3, 3),
                                                "source": traffic_source or ("replayed from " + tj.get("source", "profiles/traffic.json"))}
            # effective clock and matrix-pipe occupancy of the dominant kernel: from the committed SQ / GRBM counter pass (a separate
            # rocprofv3 --pmc run, profiles/sq_k7.json) -- tells a pipe that idles from a chip that clocks down under the load:
            # `peak` assumes 2.4 GHz and a pipe that never waits
            sq = None
            spath = os.path.join(ROOT, "profiles", "sq_k7.json")
            if os.path.exists(spath) and ops.H2 and (V, B, L) == (3, 256, 5000):
                sj = json.load(open(spath))
                want = "conv_h2w2_kernel<%d" % dom_tag[1] if dom_tag[0] == "conv_bwd_weight" else "conv_h2_kernel<%d, 0, 2" % dom_tag[1]
                hit = [v for k_, v in sj.get("by_kernel", {}).items() if k_.startswith(want)]
                if hit:
                    sq = dict(hit[0], source="replayed from " + sj.get("source", "profiles/sq_k7.json"))
            roof.update(effective_clock_GHz=sq["effective_clock_GHz"] if sq else None, pipe_busy=sq["pipe_busy"] if sq else None,
                        clock_source=sq["source"] if sq else None)
            roof.update(traffic=traffic, traffic_source=traffic_source, kernel_tag="/".join(str(x) for x in dom_tag),
                        kernel=("conv_h2w2_kernel (weight gradient on two-term fp16 splits of both operands, producer / consumer waves)" if dom_tag[0] == "conv_bwd_weight" and roof["executed_fp16_mfma_flops"] > 0
                                else KNAME[3] if roof["executed_fp16_mfma_flops"] > 0 else "fp32 conv kernel") +
                               (", launches of the timed region" if use is not serial[dom_tag] else ", launches of the single-stream breakdown steps"),
                        launches=len(use), ms_per_step_serialized=round(sum(serial[dom_tag]) / max(extra_steps, 1), 3),
                        side_stream=os.environ.get("NEF_SIDE_STREAM", "auto") != "0")
            k7 = {}
            for role in ("conv_fwd", "conv_bwd_data", "conv_bwd_weight"):
                tg = (role, 7, V, 128, 128, B, T)
                if tg in serial:
                    k7[role] = price(tg, sum(serial[tg]) / len(serial[tg]))
            roof["k7_kernels_one_at_a_time"] = k7
            # whole step: binding floors of the conv launches + the HBM-bound passes at the HBM peak, over the timed step
            fl_conv = sum(price(tg, 1.0)["floor_ms"] * len(v) for tg, v in serial.items()) / max(extra_steps, 1)
            fl_hbm = sum(tag[2] for tag, s_, e_ in prof_all if tag[0] == "hbm") / max(extra_steps, 1) / HBM_PEAK_GBS / 1e6
            ms_step = 1e3 * dt / args.steps
            ex32 = sum(price(tg, 1.0)["executed_fp32_mfma_flops"] * len(v) for tg, v in serial.items()) / max(extra_steps, 1)
            ex16 = sum(price(tg, 1.0)["executed_fp16_mfma_flops"] * len(v) for tg, v in serial.items()) / max(extra_steps, 1)
            alg = sum(price(tg, 1.0)["algorithmic_flops_per_launch"] * len(v) for tg, v in serial.items()) / max(extra_steps, 1)
            roof["whole_step"] = {
                "floor_ms": round(fl_conv + fl_hbm, 3), "conv_floor_ms": round(fl_conv, 3), "hbm_pass_floor_ms": round(fl_hbm, 3),
                "frac": round((fl_conv + fl_hbm) / ms_step, 4),
                "algorithmic_TFLOP_per_step": round(alg / 1e12, 4), "executed_fp32_mfma_TFLOP_per_step": round(ex32 / 1e12, 4),
                "executed_fp16_mfma_TFLOP_per_step": round(ex16 / 1e12, 4),
                "basis": "sum over the tagged conv launches of one step of max(matrix floor, HBM floor) + algorithmic bytes of the "
                         "HBM-bound passes / 8 TB/s, over the timed ms_per_step"}
        by_kernel = {}
        hbm = {}
        for tag, s, e in prof_all:
            if tag[0] == "hbm":            # HBM-bound passes: ("hbm", name, algorithmic bytes)
                h = hbm.setdefault(tag[1], [0.0, 0.0, 0])
                h[0] += tag[2]
                h[1] += s.elapsed_time(e)
                h[2] += 1
                continue
            by_kernel.setdefault(tag, []).append(s.elapsed_time(e))
        # the set BASELINE.json's ">= 40 % of the HBM roofline" applies to; times are HIP events around each
        # launch of the single-stream breakdown steps: every pass alone on the chip
        if os.environ.get("NEF_BENCH_DUMP"):        # per-launch event times of the untimed breakdown steps
            with open(os.environ["NEF_BENCH_DUMP"], "w") as f:
                json.dump([["/".join(str(x) for x in tag), round(s.elapsed_time(e), 4)] for tag, s, e in prof_all], f)
        hbm_bound = {k: {"GBps": round(v[0] / (v[1] * 1e-3) / 1e9, 1), "frac": round(v[0] / (v[1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 3),
                         "ms_per_step": round(v[1] / extra_steps, 3), "launches_per_step": v[2] // extra_steps}
                     for k, v in sorted(hbm.items(), key=lambda kv: -kv[1][1]) if v[1] > 0}
        breakdown = {"/".join(str(x) for x in k): round(sum(v) / extra_steps, 3) for k, v in sorted(
            by_kernel.items(), key=lambda kv: -sum(kv[1]))[:12]}
        # every tagged conv launch of a step (single-stream breakdown steps): launches, time per launch, binding floor and fraction
        conv_table = {}
        for tg, v in sorted(serial.items(), key=lambda kv: -sum(kv[1])):
            pr = price(tg, sum(v) / len(v))
            conv_table["/".join(str(x) for x in tg)] = {
                "launches": len(v) // max(extra_steps, 1), "ms_per_launch": round(sum(v) / len(v), 4),
                "ms_per_step": round(sum(v) / max(extra_steps, 1), 3), "floor_ms": pr["floor_ms"], "bound": pr["bound"],
                "mfma_floor_ms": pr["mfma_floor_ms"], "hbm_floor_ms": pr["hbm_floor_ms"], "frac": pr["frac"]}
        sec = None
        if world == 1 and not args.no_secondary:
            del model, optim, data, loss, final_loss_t
            torch.cuda.empty_cache()
            try:
                sec = secondary(dev, (V, B, L, not args.no_dropout) if ops.H2 else None)
            except Exception as exc:  # the headline stays valid; the failure is visible in the line
                sec = {"error": f"{type(exc).__name__}: {exc}"}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(V, L, args.cpu_batch, args.cpu_steps, args.cpu_threads.split(","))
        line = {
            "metric": "ECG-samples/sec (train step)", "value": round(world * B * args.steps / dt, 2),
            "unit": "ECG-samples/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
            "config": {"workload": f"{_config_name(V, B, L)}: Nef-Net train step, {V}-lead len={L}, batch={B}/GPU, "
                                   f"3-view-in -> 1-view-out, Standin losses on, dropout "
                                   f"{'off' if args.no_dropout else 'on'}",
                       "global_batch": world * B, "seq_len": L, "leads": V, "parallelism": f"dp{world}"},
            "roofline": roof, "cpu_baseline": cpu, "final_loss": final_loss, "conv_ms_per_step": breakdown, "conv_launches": conv_table,
            "conv_launches_note": "every tagged conv launch of one step, each timed alone (single-stream breakdown steps): launches per "
                                  "step, ms per launch, max(matrix floor, HBM floor) and floor / time; tag = role/K/groups/Cin_g/Cout_g/"
                                  "batch/T[/prologue]",
            "hbm_bound": hbm_bound, "hbm_bound_schedule": "single stream, every launch alone (untimed breakdown steps)",
            "hbm_bound_notes": {"stem_bwd_weight": "listed for its traffic only: 2 x 28.8 MFLOP over 1.98 MB per sample = 29 FLOP/B, "
                                                   "above the 19.7 FLOP/B ridge -- fp32-compute-bound, runs its two GEMMs on the "
                                                   "matrix cores (DESIGN 3.7)",
                                "roi_unpool_bwd": "adjoint of a variable-length linear resampling: gather arithmetic, latency-bound"},
            "secondary": sec,
            "hip_graph": bool(args.graph),
            # split-fp16 launches (waves) of the whole run whose scaled operand left fp16 range and was clamped: 0 unless an operand
            # grew more than ops.H2_HEADROOM (64) x between two consecutive steps (ops.h2_clamped); a step that contains such a launch
            # is skipped on the device (ops.h2_taint -> sgd_momentum) and counted in h2_skipped_steps; Solver checks both per epoch
            "h2_clamped_waves": int(ops.h2_clamped(reset=False)) if ops.H2 else None,
            "h2_skipped_steps": int(ops.h2_skipped(reset=False)) if ops.H2 else None,
            # call sites whose operand was heavy-tailed when the site measured it (more than ops.H2_TAIL_FRAC = 90 % of its nonzero
            # elements below 2^-11 of its largest: outside the format's full-precision window) -- a model that counts here wants NEF_H2=0
            "h2_tail_sites": int(ops.h2_tail_sites(reset=False)) if ops.H2 else None,
            "h2_tail_worst": ([round(v, 12) for v in ops.h2_tail_stats()] if ops.H2 else None),      # largest (count, energy) fraction below the window at any site
            "h2_headroom": f"{ops.H2_HEADROOM}x growth of an operand between two consecutive steps",
            # N > 1: time the launching stream spends waiting for gradient collectives per step (the encoder bucket's
            # all-reduce + whatever is left of the early bucket's, which runs under the encoder's backward pass)
            "allreduce_ms_exposed": (round(sum(a.elapsed_time(b) for a, b in ar_events) / max(args.steps, 1), 4)
                                     if ar_events else None),
        }
        if cpu:
            line["gpu_over_cpu"] = round(line["value"] / cpu["value"], 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
