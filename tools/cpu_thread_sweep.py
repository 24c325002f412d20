"""Host-thread sweep of the CPU baseline (the torch-CPU oracle's train step, config-2 shape at batch 32): which
torch.set_num_threads value is the strongest CPU baseline on this box.  Writes a markdown table.
usage: python tools/cpu_thread_sweep.py [out.md]"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np      # noqa: E402
import torch            # noqa: E402

from electrocardio_panorama_amd import synth          # noqa: E402
from oracle import nefnet_oracle as orc               # noqa: E402

V, L, B, STEPS = 3, 5000, 32, 2
ncpu = os.cpu_count() or 1
rows = []
for n in [t for t in (8, 16, 32, 64, 128, 256) if t <= ncpu] + ([ncpu] if ncpu not in (8, 16, 32, 64, 128, 256) else []):
    torch.set_num_threads(n)
    P = orc.require_grad(orc.reference_style_init(V, seed=123))
    Bf, opt = orc.fresh_buffers(), orc.SGDState(0.1)
    batch = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.make_batch(B, V, L, seed=123).items()}
    random.seed(123)
    orc.train_step(P, Bf, opt, batch)
    t0 = time.perf_counter()
    for _ in range(STEPS):
        orc.train_step(P, Bf, opt, batch)
    dt = (time.perf_counter() - t0) / STEPS
    rows.append((n, dt, B / dt))
    print(n, round(dt, 3), round(B / dt, 2), flush=True)
out = [f"# CPU-baseline thread sweep (torch-CPU oracle train step, batch {B}, V={V}, L={L}; {STEPS} timed steps after 1 warm-up)",
       "", f"host: os.cpu_count() = {ncpu}, torch {torch.__version__}", "",
       "| torch threads | s / step | ECG-samples/s |", "|---:|---:|---:|"]
out += [f"| {n} | {dt:.3f} | {v:.2f} |" for n, dt, v in rows]
text = "\n".join(out) + "\n"
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(text)
print(text)
