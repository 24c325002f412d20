"""Soak: N train steps at config 2, report loss trajectory and allocator high-water marks (leak check)."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from electrocardio_panorama_amd import synth
from electrocardio_panorama_amd.network import build_loss, build_model
from electrocardio_panorama_amd.solver.optim_scheduler import get_optimizer
from electrocardio_panorama_amd.utils import seed_torch

B, V, L, N = int(os.environ.get("B", 256)), 3, 5000, int(os.environ.get("N", 40))
cfg = bench.make_cfg(V)
seed_torch(123)
model = build_model(cfg).float().cuda().train()
lossf, optim = build_loss(cfg), get_optimizer(cfg, model.parameters())
meta = [synth.make_batch(B, V, L, seed=s) for s in range(4)]
dev = [{k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in m.items()} for m in meta]
hist = []
for i in range(N):
    d = dev[i % 4]
    out, sp, sl = model(d["data"], d["input_theta"], d["target_theta"], d["rois"], phase="train")
    ls = lossf(out, sp, sl, d["target_view"].unsqueeze(1), cfg)
    ls[0].backward(); optim.step(); optim.zero_grad()
    hist.append(ls[0].detach())
    if i in (4, N // 2, N - 1):
        torch.cuda.synchronize()
        print(f"step {i}: loss {float(hist[-1]):.5f}  allocated {torch.cuda.memory_allocated()/2**30:.2f} GiB  "
              f"reserved {torch.cuda.memory_reserved()/2**30:.2f} GiB  peak {torch.cuda.max_memory_allocated()/2**30:.2f} GiB", flush=True)
h = torch.stack(hist).cpu().numpy()
print("loss first/last:", h[0], h[-1], "finite:", bool(np.isfinite(h).all()), "status:", model.segment_status())
from electrocardio_panorama_amd import ops
print("split-fp16: clamped waves", ops.h2_clamped(reset=False), "skipped steps", ops.h2_skipped(reset=False))
st = next(iter(ops._AMAX.values())) if ops._AMAX else None
if st is not None:
    cur = st["cur"][:st["n"]].cpu().numpy()
    if (cur > 0).any():
        print(f"sites {st['n']}, reference magnitudes min {cur[cur > 0].min():.3e} max {cur.max():.3e}")
