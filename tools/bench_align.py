"""Does row alignment of the [B, C, T] tensors matter to the F(4,3) conv kernels?  Same conv at T = 2500 (rows start 16 bytes
off a 128-byte line, every tile edge is a partial line) and T = 2560 (every row and tile edge line-aligned); ns per output element."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from electrocardio_panorama_amd import ops
from electrocardio_panorama_amd.ops import GV
for name, Cig, Cog, B in [("128->128", 128, 128, 768), ("64->128", 64, 128, 768), ("64->64", 64, 64, 768)]:
    for T in (2500, 2560, 2432):
        x = torch.randn(B, Cig, T, device="cuda")
        w = torch.randn(Cog, Cig, 3, device="cuda") * 0.05
        wp = ops.pack_weight(w, 1, T=T, f4=True)
        fn = lambda: ops.conv(GV.dense(x, 1), wp, Cog, 3, relu=True)
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
        print(f"{name:9s} T={T}: {ms:.3f} ms  {ms * 1e6 / (B * Cog * T):.4f} ns per output element", flush=True)
