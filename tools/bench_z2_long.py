"""Experiment (round 6): the ROI-segment convs (21 groups x 128 ch, T = 16 / 32 per sample) as they run today -- the packed short-row
form of conv_h2_kernel / the fp32 direct weight gradient -- against the SAME work laid out as long rows [1][21*128][B*(T+gap)]
(samples end to end with zero gaps) on the regular split-fp16 kernels.  Timing only (gap columns are garbage here)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from electrocardio_panorama_amd import ops
from electrocardio_panorama_amd.ops import GV

B = int(os.environ.get("B", 256))
iters = int(os.environ.get("ITERS", 20))


def timeit(fn):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for K, G, Cig, Cog, T, gap in ((3, 21, 128, 128, 16, 4), (3, 21, 64, 128, 32, 8), (3, 21, 128, 128, 32, 8), (1, 21, 64, 128, 32, 8)):
    w = torch.randn(G * Cog, Cig, K, device="cuda") * 0.05
    # today's form
    x = torch.randn(B, G * Cig, T, device="cuda")
    gy = torch.randn(B, G * Cog, T, device="cuda")
    wp = ops.pack_weight(w, G, T=T)
    assert getattr(wp, "nef_wino", 0) == 3
    t_f = timeit(lambda: ops.conv(GV.dense(x, G), wp, Cog, K, relu=True, x_scale=1.0))
    t_w = timeit(lambda: ops.conv_bwd_weight(GV.dense(x, G), GV.dense(gy, G), K))
    # long rows
    TL = B * (T + gap)
    xl = torch.randn(1, G * Cig, TL, device="cuda")
    gl = torch.randn(1, G * Cog, TL, device="cuda")
    wpl = ops.pack_weight(w, G, T=TL)
    assert getattr(wpl, "nef_wino", 0) == 3
    t_fl = timeit(lambda: ops.conv(GV.dense(xl, G), wpl, Cog, K, relu=True, x_scale=1.0))
    ok = ops.h2w_ok(K, Cig, Cog, TL)
    t_wl = timeit(lambda: ops.conv_bwd_weight(GV.dense(xl, G), GV.dense(gl, G), K, h2=True, x_scale=64.0, gy_scale=64.0)) if ok else float("nan")
    print(f"K={K} {Cig}->{Cog} g{G} T={T}: fwd packed {t_f*1e3:7.1f} us -> long rows {t_fl*1e3:7.1f} us;  weight grad today {t_w*1e3:7.1f} us -> long rows (split-fp16) {t_wl*1e3:7.1f} us", flush=True)
