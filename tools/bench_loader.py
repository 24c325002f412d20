"""Throughput of the host `meta` pipeline (reference codes/train_net.py:27-28 + dataset/tianchi.py:84-225).

Builds a synthetic Tianchi-format tree (N recordings of 8 x 5000 samples with P/R/T on/off index lists, the on-disk layout of
reference codes/README.md:13), then drives it exactly as the packaged trainer does -- train_net.build_loaders (batch 32,
16 workers, pinned memory, the per-item restatement of EcgTianChiInterval.__getitem__) behind prefetch.DevicePrefetcher when
a HIP device is present -- and prints samples/s next to the rate one GPU consumes at that shape.

    python tools/bench_loader.py [--n 2048] [--epochs 3] [--workers 16] [--batch 32]
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np   # noqa: E402
import torch         # noqa: E402


def make_tree(root, n, seed=0):
    """n recordings: smooth 8-lead signals with a beat every 600..900 samples and plausible wave boundaries."""
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(root, "npy"), exist_ok=True)
    os.makedirs(os.path.join(root, "json"), exist_ok=True)
    names = []
    t = np.arange(5000)
    for i in range(n):
        period = int(rng.integers(600, 900))
        p_on = np.arange(int(rng.integers(20, 100)), 5000 - period, period)
        lab = {"P on": p_on, "P off": p_on + 60, "R on": p_on + 110, "R off": p_on + 170, "T on": p_on + 260, "T off": p_on + 400}
        sig = np.zeros((8, 5000))
        for c, w in ((85, 18), (140, 9), (330, 40)):
            for p in p_on:
                sig += rng.normal(1.0, 0.3, size=(8, 1)) * np.exp(-0.5 * ((t - p - c) / w) ** 2)
        sig += rng.normal(0, 0.01, size=sig.shape)
        np.save(os.path.join(root, "npy", f"{i}.npy"), sig.astype(np.float32))
        with open(os.path.join(root, "json", f"{i}.json"), "w") as f:
            json.dump({k: v.tolist() for k, v in lab.items()}, f)
        names.append(f"{i}.json")
    with open(os.path.join(root, "list.txt"), "w") as f:
        f.write("\n".join(names))
    return os.path.join(root, "list.txt")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2048)
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--workers", type=int, default=16)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--leads", type=int, default=3)
    a = ap.parse_args()
    from electrocardio_panorama_amd import train_net
    from electrocardio_panorama_amd.config import get_defaults, resolve_config_path
    from electrocardio_panorama_amd.prefetch import DevicePrefetcher
    root = tempfile.mkdtemp(prefix="nef_loader_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        lst = make_tree(root, a.n)
        cfg = get_defaults()
        cfg.merge_from_file(resolve_config_path("config/nef_net.yml"))
        cfg.DATA.lead_num = a.leads
        cfg.DATA.dataset = "tianchi"
        cfg.DATA.train_label_path = cfg.DATA.test_label_path = lst
        cfg.DATA.train_data_root = os.path.join(root, "npy")
        cfg.DATA.train_label_root = os.path.join(root, "json")
        # the packaged loader, with the worker count under test
        import torch.utils.data as tud
        orig = tud.DataLoader

        def patched(*args, **kw):
            if kw.get("num_workers", 0) == 16:
                kw["num_workers"] = a.workers
            return orig(*args, **kw)
        tud.DataLoader = patched
        try:
            dl = train_net.build_loaders(cfg, batch_size=a.batch, phases=("train",))[0]
        finally:
            tud.DataLoader = orig
        dev = torch.device("cuda", 0) if torch.cuda.is_available() else None
        res = {}
        for mode in (["host"] + (["device"] if dev is not None else [])):
            it_src = dl if mode == "host" else DevicePrefetcher(dl, dev)
            n, t0 = 0, None
            for ep in range(a.epochs + 1):            # epoch 0: worker start-up, untimed
                if ep == 1:
                    t0, n = time.perf_counter(), 0
                for meta in it_src:
                    n += int(meta["data"].shape[0])
            if dev is not None:
                torch.cuda.synchronize()
            res[mode] = n / (time.perf_counter() - t0)
        out = {"recordings": a.n, "batch": a.batch, "workers": a.workers, "host_cores": os.cpu_count(), "leads": a.leads,
               "samples_per_s_host_batches": round(res["host"], 1),
               "samples_per_s_on_device": round(res.get("device", float("nan")), 1)}
        print(json.dumps(out))
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
