"""Inference panorama sweep (BASELINE configs[3]): 1 view in -> Q queried angles, eval mode. Prints latency + rates."""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from electrocardio_panorama_amd import synth
from electrocardio_panorama_amd.network import build_model
import bench

B = int(os.environ.get("B", 1024)); V = int(os.environ.get("V", 1)); L = int(os.environ.get("L", 512)); Q = int(os.environ.get("Q", 360))
cfg = bench.make_cfg(V)
torch.manual_seed(123)
m = build_model(cfg).float().cuda().eval()
m.panorama_dtype = os.environ.get("PANO", "fp32")
BUDGET = os.environ.get("PAIR_BUDGET")
if BUDGET:
    from electrocardio_panorama_amd import engine
    engine.sweep_eval_h.__defaults__ = (int(BUDGET),)
meta = synth.make_batch(B, V, L, seed=123, Q=Q)
t = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in meta.items()}
def run():
    random.seed(0)
    return m(t["data"], t["input_theta"], t["target_theta"], t["rois"], rest_theta=t["rest_theta"], phase="test")
run(); torch.cuda.synchronize()
t0 = time.perf_counter(); n = 3
for _ in range(n):
    out = run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
flops = B * Q * 113.5e6 * (L / 512)
print(f"sweep[{m.panorama_dtype}] B={B} V={V} L={L} Q={Q}: {dt*1e3:.1f} ms  {B/dt:.1f} samples/s  {B*Q/dt:.0f} views/s  "
      f"{flops/dt/1e12:.1f} TFLOP/s (decoder convs)  out {4*B*Q*L/dt/1e9:.1f} GB/s  peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
