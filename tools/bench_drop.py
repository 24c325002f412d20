"""Cost of the in-kernel dropout decision in the conv epilogues: K=7 / K=3 encoder shapes with and without drop_p."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from electrocardio_panorama_amd import ops
from electrocardio_panorama_amd.ops import GV
for name, K, G, C, B, T in [("enc k7", 7, 3, 128, 256, 1250), ("w_conv k3", 3, 3, 128, 256, 1250)]:
    x = torch.randn(B, G * C, T, device="cuda")
    w = torch.randn(G * C, C, K, device="cuda") * 0.05
    wp = ops.pack_weight(w, G, T=T)
    for label, kw in [("relu", dict(relu=True)), ("relu+dropout", dict(relu=True, drop_p=0.2, drop_scale=1.25, seed=3))]:
        fn = lambda: ops.conv(GV.dense(x, G), wp, C, K, **kw)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            fn()
        e.record()
        torch.cuda.synchronize()
        print(f"{name:10s} {label:14s} {s.elapsed_time(e) / 20:.3f} ms")
