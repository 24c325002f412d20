"""Micro-benchmark of the MFMA conv kernels at the bench shapes (one line per shape: ms, TFLOP/s)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from electrocardio_panorama_amd import ops
from electrocardio_panorama_amd.ops import GV

SHAPES = [  # (name, K, G, Cig, Cog, B, T)
    ("enc k7 128->128 g3", 7, 3, 128, 128, 256, 1250),
    ("w_conv k3 128->128 g3", 3, 3, 128, 128, 256, 1250),
    ("z1 k3 64->128 g3", 3, 3, 64, 128, 256, 1250),
    ("dec k3 256->128", 3, 1, 256, 128, 768, 2500),
    ("dec k3 128->128", 3, 1, 128, 128, 768, 2500),
    ("dec k3 128->64", 3, 1, 128, 64, 768, 5000),
    ("dec k3 64->64", 3, 1, 64, 64, 768, 5000),
    ("dec bwd k3 64->128", 3, 1, 64, 128, 768, 5000),
    ("roi k3 128->128 g21 T16", 3, 21, 128, 128, 256, 16),
]
only = sys.argv[1] if len(sys.argv) > 1 else None
iters = int(os.environ.get("ITERS", 10))
whats = os.environ.get("ONLY_WHAT", "fwd,wino,bwd_w,bwd_ww,bwd_h2").split(",")
for name, K, G, Cig, Cog, B, T in SHAPES:
    if only and only not in name:
        continue
    x = torch.randn(B, G * Cig, T, device="cuda")
    w = torch.randn(G * Cog, Cig, K, device="cuda") * 0.05
    gy = torch.randn(B, G * Cog, T, device="cuda")
    wp = ops.pack_weight(w, G)
    wpw = ops.pack_weight(w, G, T=T, f4=os.environ.get("F4", "1") == "1")      # F4=0: the F(2,3) form
    flops = 2.0 * B * G * Cog * T * Cig * K
    for what in ("fwd", "wino", "bwd_w", "bwd_ww", "bwd_h2"):
        if what not in whats:
            continue
        if what == "wino" and not getattr(wpw, "nef_wino", False):
            continue
        fn = (lambda: ops.conv(GV.dense(x, G), wp, Cog, K, relu=True)) if what == "fwd" else \
             (lambda: ops.conv(GV.dense(x, G), wpw, Cog, K, relu=True, x_scale=(1.0 if getattr(wpw, "nef_wino", 0) == 3 else 0.0))) if what == "wino" else \
             (lambda: ops.conv_bwd_weight(GV.dense(x, G), GV.dense(gy, G), K, wino=False)) if what == "bwd_w" else \
             (lambda: ops.conv_bwd_weight(GV.dense(x, G), GV.dense(gy, G), K, wino=4)) if what == "bwd_ww" else \
             (lambda: ops.conv_bwd_weight(GV.dense(x, G), GV.dense(gy, G), K, h2=True, x_scale=64.0, gy_scale=64.0))
        if what == "bwd_h2" and not ops.h2w_ok(K, Cig, Cog, T):
            continue
        if what == "bwd_ww" and (K not in (3, 7) or T < 64 or T % 2):
            continue
        for _ in range(int(os.environ.get("WARM", 30))):      # the first kernel timed in a process runs ~10 % slow for a few dozen launches
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / iters
        print(f"{name:26s} {what:6s} {ms:8.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s", flush=True)
