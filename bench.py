#!/usr/bin/env python
"""Nef-Net train-step benchmark (BASELINE.json metric: ECG-samples/sec of one train step).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One process per GPU; each rank trains on its own shard of the synthetic batch (weak scaling: per-GPU batch
fixed), gradients are summed with ONE RCCL all-reduce of the flat gradient buffer per step.  A step is
forward + losswrapper + backward + fused momentum-SGD on inputs already resident in HBM.  Rank 0 prints one
JSON line.  At N=1 the line also carries the CPU baseline (the torch-CPU oracle timed on this box's host cores on
a bounded sample of the same workload) and the roofline of the dominant kernel (the k=7 grouped-conv MFMA
kernel), timed live with HIP events on the launch stream.
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
    ap.add_argument("--graph", action="store_true", help="replay the step as one captured hipGraph (launch-bound shapes)")
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


def _config_name(V, B, L):
    """Which BASELINE.json config the per-GPU shape is."""
    return {(3, 256, 5000): "configs[1]", (8, 256, 5000): "configs[2] (per-GPU shard)", (1, 4, 2048): "configs[0]"}.get(
        (V, B, L), "custom shape")


def main():
    args = parse()
    from electrocardio_panorama_amd import ops, parallel, synth
    from electrocardio_panorama_amd.network import build_loss, build_model
    from electrocardio_panorama_amd.solver.optim_scheduler import get_optimizer
    from electrocardio_panorama_amd.utils import seed_torch

    rank, world, local = parallel.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
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

    graphed = None
    if args.graph:
        from electrocardio_panorama_amd.graph import GraphedTrainStep
        graphed = GraphedTrainStep(model, cfg)

    def step():
        if graphed is not None:
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
    dom = {("conv_fwd", 7, V, 128, 128, B, T_lat), ("conv_bwd_data", 7, V, 128, 128, B, T_lat)}
    timing = rank == 0 and not args.no_kernel_events and not args.graph
    ops.PROFILE, ops.PROFILE_ONLY = ([] if timing else None), (lambda tag: tag in dom)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    prof, ops.PROFILE, ops.PROFILE_ONLY = ops.PROFILE or [], None, None
    final_loss_t = loss
    prof_all, extra_steps = [], 2
    if not args.no_kernel_events and not args.graph:       # every rank takes the extra steps (they contain the all-reduce)
        ops.PROFILE = [] if rank == 0 else None
        # one more bracketed step first, whose events are dropped: the first event pair recorded behind a cross-stream
        # wait can come back with the wait inside it (seen once: 52 ms on a 0.23 ms launch of the first bracketed step)
        step()
        fence()
        if rank == 0:
            ops.PROFILE = []
        for _ in range(extra_steps):
            step()
        fence()
        prof_all, ops.PROFILE = ops.PROFILE or [], None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(final_loss_t.item())

    if rank == 0:
        # dominant kernel: the K=7 grouped conv over the encoder's [B,128V,T] activations (conv_wino_kernel<7,..>: Winograd
        # F(2,3) on the taps split 3+3+1; conv_fwd_kernel<7,..> with NEF_WINOGRAD=0): the 6 forward launches per step.  The 6
        # backward-data launches of the same shape run on conv_wino4_kernel<7,..> (F(4,3)) and share the matrix pipes with
        # the bwd-weight kernels of the side stream, so the roofline is read from the forward launches of the timed region;
        # the backward-data average is reported next to it.
        # `achieved` counts ALGORITHMIC flops (2*B*Cout*T*Cin_g*K, the direct-convolution count every conv is priced
        # with); the Winograd form executes 10/14 of them on the matrix cores, so `frac` can exceed 1 --
        # `mfma_pipe_frac` is the executed matrix-core work against the same peak.
        T = L // 4
        key = ("conv_fwd", 7, V, 128, 128, B, T)
        times = [s.elapsed_time(e) for tag, s, e in prof if tag == key]
        times_bd = [s.elapsed_time(e) for tag, s, e in prof if tag == ("conv_bwd_data",) + key[1:]]
        flops = 2.0 * B * (128 * V) * T * 128 * 7
        # multiplies executed on the matrix cores per algorithmic multiply: direct 1, F(2,3) on the taps split 3+3+1 10/14
        # (the encoder convs always take F(2,3); F(4,3) is for the decoder convs only, see ops.WINO_FWD)
        wino_exec = 10.0 / 14.0 if ops.WINOGRAD else 1.0
        roof = None
        if times:
            avg_ms = sum(times) / len(times)
            ach = flops / (avg_ms * 1e-3) / 1e12
            traffic = traffic_source = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath) and (V, B, L) == (3, 256, 5000):     # PMC pass was taken at configs[1] only
                tj = json.load(open(tpath))
                traffic = tj.get("conv_fwd_k7_bytes_per_launch")
                # NOT measured by this run: replayed from the committed rocprofv3 --pmc pass (separate run, as the
                # counters cannot be collected together with timing)
                traffic_source = "replayed from " + tj.get("source", "profiles/traffic.json")
            # forward launches: conv1 of a block reads x and writes h; conv2 reads h AND the residual x, writes y
            act = 4.0 * B * 128 * V * T
            alg_bytes = ((2 * act) + (3 * act)) / 2 + 4.0 * 128 * V * 128 * 7
            roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                    "kernel": ("conv_wino_kernel<7,2,0> (k7 grouped conv, Winograd F(2,3) on taps 3+3+1)" if ops.WINOGRAD
                               else "conv_fwd_kernel<7,2,0> (k7 grouped conv, direct)") + ", forward launches",
                    "launches": len(times),
                    "executed_mfma_flops_per_launch": flops * wino_exec,
                    "mfma_pipe_frac": round(ach * wino_exec / FP32_MFMA_PEAK_TFLOPS, 4),
                    "avg_ms": round(avg_ms, 4), "flops_per_launch": flops, "algorithmic_bytes_per_launch": alg_bytes,
                    "hbm_GBps": round(alg_bytes / (avg_ms * 1e-3) / 1e9, 1),
                    "avg_ms_bwd_data_launches_overlapped": round(sum(times_bd) / max(len(times_bd), 1), 4),
                    "bwd_data_kernel": "conv_wino4_kernel<7,4,0> (F(4,3) on taps 3+3+1: no decision is taken on a gradient)",
                    "side_stream": os.environ.get("NEF_SIDE_STREAM", "auto") != "0"}
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
        # launch in the live (two-stream) schedule, so a pass that shares the chip with a side-stream MFMA kernel reads low;
        # profiles/r02_hbm_kernels.md has the same table with every launch alone
        if os.environ.get("NEF_BENCH_DUMP"):        # per-launch event times of the untimed breakdown steps
            with open(os.environ["NEF_BENCH_DUMP"], "w") as f:
                json.dump([["/".join(str(x) for x in tag), round(s.elapsed_time(e), 4)] for tag, s, e in prof_all], f)
        hbm_bound = {k: {"GBps": round(v[0] / (v[1] * 1e-3) / 1e9, 1), "frac": round(v[0] / (v[1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 3),
                         "ms_per_step": round(v[1] / extra_steps, 3), "launches_per_step": v[2] // extra_steps}
                     for k, v in sorted(hbm.items(), key=lambda kv: -kv[1][1]) if v[1] > 0}
        breakdown = {"/".join(str(x) for x in k): round(sum(v) / extra_steps, 3) for k, v in sorted(
            by_kernel.items(), key=lambda kv: -sum(kv[1]))[:12]}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(V, L, args.cpu_batch, args.cpu_steps, args.cpu_threads.split(","))
        line = {
            "metric": "ECG-samples/sec (train step)", "value": round(world * B * args.steps / dt, 2),
            "unit": "ECG-samples/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{_config_name(V, B, L)}: Nef-Net train step, {V}-lead len={L}, batch={B}/GPU, "
                                   f"3-view-in -> 1-view-out, Standin losses on, dropout "
                                   f"{'off' if args.no_dropout else 'on'}",
                       "global_batch": world * B, "seq_len": L, "leads": V, "parallelism": f"dp{world}"},
            "roofline": roof, "cpu_baseline": cpu, "final_loss": final_loss, "conv_ms_per_step": breakdown,
            "hbm_bound": hbm_bound,
            "hip_graph": bool(args.graph),
        }
        if cpu:
            line["gpu_over_cpu"] = round(line["value"] / cpu["value"], 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
