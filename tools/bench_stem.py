"""Stem forward / weight-gradient at the config-2 shape (one line each)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from electrocardio_panorama_amd import ops

B, V, L = 256, 3, 5000
x = torch.rand(B, V, L, device="cuda")
w = torch.randn(128 * V, 1, 15, device="cuda") * 0.1
gy = torch.randn(B, 128 * V, L // 4, device="cuda")


def t(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


print("stem_fwd        %.1f us" % t(lambda: ops.stem_fwd(x, w)))
print("stem_bwd_weight %.1f us" % t(lambda: ops.stem_bwd_weight(x, w, gy)))
