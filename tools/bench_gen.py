"""Synthesis from latents (BASELINE configs[4]): gen_ecg(z1, z2, query_theta, rois) on one GPU's share (512 samples,
3 leads, 12 angles, L=5000).  PANO=fp16|fp32 selects the decoder arithmetic.  Prints latency and rates."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from electrocardio_panorama_amd import synth
from electrocardio_panorama_amd.network import build_model
import bench

B = int(os.environ.get("B", 512)); V = int(os.environ.get("V", 3)); L = int(os.environ.get("L", 5000)); Q = int(os.environ.get("Q", 12))
torch.manual_seed(123)
m = build_model(bench.make_cfg(V)).float().cuda().eval()
m.panorama_dtype = os.environ.get("PANO", "fp16")
meta = synth.make_batch(B, V, L, seed=123, Q=Q)
rois = torch.from_numpy(np.ascontiguousarray(meta["rois"])).cuda()
theta = torch.from_numpy(np.ascontiguousarray(meta["rest_theta"])).cuda()
g = torch.Generator(device="cuda").manual_seed(5)
z1 = torch.randn(B, 128 * V, L // 4, device="cuda", generator=g) * 0.1
z2 = torch.randn(B, 128 * V, 7, 32, device="cuda", generator=g) * 0.1
out = m.gen_ecg(z1, z2, theta, rois); torch.cuda.synchronize()
n = int(os.environ.get("N", 5)); t0 = time.perf_counter()
for _ in range(n):
    out = m.gen_ecg(z1, z2, theta, rois)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
flops = B * Q * 113.5e6 * (L / 512)
print(f"gen_ecg[{m.panorama_dtype}] B={B} V={V} L={L} Q={Q}: {dt*1e3:.2f} ms  {B/dt:.0f} samples/s  {B*Q/dt:.0f} views/s  "
      f"{flops/dt/1e12:.1f} TFLOP/s (decoder convs)  finite={bool(torch.isfinite(out).all())}")
