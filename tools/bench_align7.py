"""Row-alignment sensitivity of the K=7 encoder kernels: T = 1250 (reference) vs 1280 / 1216 (line-aligned rows)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from electrocardio_panorama_amd import ops
from electrocardio_panorama_amd.ops import GV
B, G, C, K = 256, 3, 128, 7
for T in (1250, 1280, 1216, 1250):
    x = torch.randn(B, G * C, T, device="cuda")
    gy = torch.randn(B, G * C, T, device="cuda")
    w = torch.randn(G * C, C, K, device="cuda") * 0.05
    wf = ops.pack_weight(w, G, T=T)
    wb = ops.pack_weight(w, G, flip=True, T=T, f4=True)
    for name, fn in [("fwd F(2,4)+F(2,3)", lambda: ops.conv(GV.dense(x, G), wf, C, K, relu=True)),
                     ("bwd-data F(4,4)+F(4,3)", lambda: ops.conv(GV.dense(gy, G), wb, C, K, gate=GV.dense(x, G), gate_scale=1.25)),
                     ("bwd-weight", lambda: ops.conv_bwd_weight(GV.dense(x, G), GV.dense(gy, G), K, wino=4))]:
        for _ in range(60):        # the first kernel timed in a process runs ~10 % slow for its first few dozen launches
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            fn()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 20
        print(f"T={T} {name:24s} {ms:.3f} ms  {ms * 1e6 / (B * G * C * T):.4f} ns/elem", flush=True)
