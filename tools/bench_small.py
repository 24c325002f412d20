"""The slow HBM-bound passes in isolation at the config-2 shape: roi_unpool fwd/bwd, stem fwd/bwd (for PMC runs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from electrocardio_panorama_amd import ops, synth

B, V, L = 256, 3, 5000
T, C = L // 4, 128 * V
rois = torch.from_numpy(synth.make_rois(np.random.default_rng(1), B, L)).cuda()
zs = torch.randn(B, C, 7, 32, device="cuda")
gy = torch.randn(B, C, T, device="cuda")
x = torch.rand(B, V, L, device="cuda")
w = torch.randn(C, 1, 15, device="cuda") * 0.1
n = int(os.environ.get("ITERS", 5))
for _ in range(n):
    ops.roi_unpool_fwd(zs, rois, T)
    ops.roi_unpool_bwd(gy, rois)
    ops.stem_fwd(x, w)
    ops.stem_bwd_weight(x, w, gy)
torch.cuda.synchronize()
